"""Conv1d layers of the headline step: pipelined bf16x3 kernel (csrc/conv_bf16.hip) vs the producer / consumer one
(csrc/conv1d_pc.hip), forward (BN-ReLU prologue, statistics) and data gradient (BN-ReLU backward epilogue)."""
import sys, torch
import os; sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from pb_sed_amd import ops
dev = 'cuda'
def tm(f, n=20):
    for _ in range(3): f()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n): f()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n
LAYERS = [(2048, 256, 1), (256, 256, 3), (256, 256, 1), (4096, 512, 1), (512, 512, 3)]
if os.environ.get('ONLY'):
    LAYERS = [l for l in LAYERS if f'{l[0]}x{l[1]}k{l[2]}' in os.environ['ONLY'].split(',')]
PRECS = os.environ.get('PRECS', 'bf16x3,c1x3').split(',')
for cin, cout, kw in LAYERS:
    b, t = 32, 500
    x = torch.randn(b, cin, t, device=dev)
    w = torch.randn(cout, cin, kw, device=dev) / (cin * kw) ** .5
    pc = ops.PackedConv(w)
    sc, sh = torch.rand(cin, device=dev) + .5, torch.randn(cin, device=dev) * .3
    mean, invstd = torch.randn(cin, device=dev) * .1, torch.rand(cin, device=dev) + .5
    seq = torch.full((b,), t, dtype=torch.int32, device=dev)
    g = torch.randn(b, cout, t, device=dev)
    res = []
    for prec in PRECS:
        wp = pc.fwd(prec); wd = pc.dgrad(prec)
        ms = tm(lambda: ops.conv_fwd(x, pc, wp, scale=sc, shift=sh, seq_len=seq, want_stats=True, precision=prec))
        msd = tm(lambda: ops.conv_bwd_data(g, pc, wd, x.shape, None, seq, bn=(x, mean, invstd, sc, sh), precision=prec))
        res.append((ms, msd))
    fl = 2 * b * cout * cin * kw * t / 1e9
    print(f'{cin}->{cout} k{kw}: ' + ' | '.join(f'{p} fwd {r[0]:.3f} ({fl / r[0]:.0f} TF) dgrad {r[1]:.3f} ({fl / r[1]:.0f} TF)' for p, r in zip(PRECS, res)), flush=True)
