"""Bitwise run-to-run comparison of single forward ops (same inputs, 20 launches each)."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from pb_sed_amd import ops
dev = 'cuda'
torch.manual_seed(0)
N = 20

def check(name, f):
    ref = [o.clone() for o in f()]
    bad = [0] * len(ref)
    worst = [0.] * len(ref)
    for _ in range(N):
        out = f()
        torch.cuda.synchronize()
        for i, (a, b) in enumerate(zip(out, ref)):
            if not torch.equal(a, b):
                bad[i] += 1
                worst[i] = max(worst[i], ((a.double() - b.double()).abs().max() / b.double().abs().max()).item())
    print(f'{name}: differing launches per output {bad} of {N}, worst relative deviation {["%.1e" % w for w in worst]}')

b, t = 32, 500
seq = torch.full((b,), t, dtype=torch.int32, device=dev); seq[5:] -= torch.randint(0, 200, (b - 5,), device=dev, dtype=torch.int32)
for cin, cout, f, prec, pool in [(16, 16, 128, 'f32', True), (32, 64, 32, 'wino', False), (128, 128, 16, 'wino', True),
                                 (1, 16, 128, 'f32', False)]:
    x = torch.randn(b, cin, f, t, device=dev)
    w = torch.randn(cout, cin, 3, 3, device=dev) / (cin * 9) ** .5
    pc = ops.PackedConv(w)
    wp = pc.fwd(prec)
    sc, sh = (torch.rand(cin, device=dev) + .5, torch.randn(cin, device=dev) * .3) if cin > 1 else (None, None)
    def f_():
        y, idx, st = ops.conv_fwd(x, pc, wp, scale=sc, shift=sh, seq_len=seq, pool=pool, want_stats=True, precision=prec)
        return [y, st.sum(0).float()] + ([idx] if pool else [])
    check(f'conv_fwd {cin}->{cout} {prec} pool={pool} (y, stats as f32, idx)', f_)
    g = torch.randn(b, cout, f // 2 if pool else f, t, device=dev)
    idx = torch.randint(0, 2, g.shape, device=dev, dtype=torch.uint8) if pool else None
    if cin > 1:
        wd = pc.dgrad(prec)
        mean, invstd = torch.randn(cin, device=dev) * .1, torch.rand(cin, device=dev) + .5
        def g_():
            dz, st = ops.conv_bwd_data(g, pc, wd, x.shape, idx, seq, bn=(x, mean, invstd, sc, sh), precision=prec)
            return [dz, st.sum(0).float()]
        check(f'conv_bwd_data {cin}->{cout} {prec} (dz, stats as f32)', g_)

h = 256
gi = [torch.randn(t, b, 3 * h, device=dev) for _ in range(2)]
w_ih = [None, torch.randn(3 * h, h, device=dev) / 16, None, torch.randn(3 * h, h, device=dev) / 16]
b_ih = [None, torch.zeros(3 * h, device=dev), None, torch.zeros(3 * h, device=dev)]
w_hh = [torch.randn(3 * h, h, device=dev) / 16 for _ in range(4)]
b_hh = [torch.zeros(3 * h, device=dev) for _ in range(4)]
sl = torch.sort(seq, descending=True).values.contiguous()
def r_():
    hs, sv = ops.gru_stack_fwd(gi, w_ih, b_ih, w_hh, b_hh, [0, 1], sl, 2, save=True)
    return hs + sv
check('gru_stack_fwd granule (hs x4, save x4)', r_)
ops.check_gru_sync()
