"""Profiling build of conv_winox3 (WX_DBG bit 128, tools/kernel_ablation.sh): prints where consumer wave 0 of block 0 spends
its clocks.  GPU box:  DBGS="128 135" bash tools/kernel_ablation.sh conv_winox3 WX_DBG "python tools/wino_prof.py" """
import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from pb_sed_amd import ops
dev = 'cuda'
cin, cout, f, b, t = 128, 128, 16, 32, 500
x = torch.randn(b, cin, f, t, device=dev)
w = torch.randn(cout, cin, 3, 3, device=dev) / (cin * 9) ** .5
pc = ops.PackedConv(w)
sc, sh = torch.rand(cin, device=dev) + .5, torch.randn(cin, device=dev) * .3
seq = torch.full((b,), t, dtype=torch.int32, device=dev)
wp = pc.fwd('winox3')
for _ in range(3):
    y, _, _ = ops.conv_fwd(x, pc, wp, scale=sc, shift=sh, seq_len=seq, want_stats=True, precision='winox3')
torch.cuda.synchronize()
v = y.flatten()[:6].tolist()
print(f'128->128: k-clocks mfma phases {v[0]:.0f}  barriers {v[1]:.0f}  epilogues {v[2]:.0f}  total {v[3]:.0f}  tiles {v[4]:.0f} chunks/tile {v[5]:.0f}')
