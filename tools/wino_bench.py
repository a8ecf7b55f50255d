import sys, torch
import os; sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from pb_sed_amd import ops
dev = 'cuda'
def tm(f, n=10):
    for _ in range(3): f()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n): f()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n
LAYERS = [(128, 128, 16), (128, 256, 8), (64, 64, 32), (64, 128, 16), (32, 64, 32), (32, 32, 64), (16, 32, 64)]
if os.environ.get('ONLY'):
    LAYERS = [l for l in LAYERS if f'{l[0]}x{l[1]}' in os.environ['ONLY'].split(',')]
PRECS = os.environ.get('PRECS', 'f32,wino,bf16x3,winox3').split(',')
for cin, cout, f in LAYERS:
    b, t = 32, 500
    x = torch.randn(b, cin, f, t, device=dev)
    w = torch.randn(cout, cin, 3, 3, device=dev) / (cin * 9) ** .5
    pc = ops.PackedConv(w)
    sc, sh = torch.rand(cin, device=dev) + .5, torch.randn(cin, device=dev) * .3
    seq = torch.full((b,), t, dtype=torch.int32, device=dev)
    res = []
    for prec in PRECS:
        wp = pc.fwd(prec); wd = pc.dgrad(prec)
        ms = tm(lambda: ops.conv_fwd(x, pc, wp, scale=sc, shift=sh, seq_len=seq, want_stats=True, precision=prec))
        g = torch.randn(b, cout, f, t, device=dev)
        msd = tm(lambda: ops.conv_bwd_data(g, pc, wd, x.shape, None, seq, precision=prec))
        res.append((ms, msd))
    fl = 2 * b * cout * cin * 9 * f * t / 1e9
    print(f'{cin}->{cout} F{f}: ' + ' | '.join(f'{p} fwd {r[0]:.3f} ({fl / r[0]:.0f} TF) dgrad {r[1]:.3f}' for p, r in zip(PRECS, res)))
if os.environ.get('NOERR'):
    sys.exit(0)

# rounding error of both kernels against an fp64 convolution (small batch)
import torch.nn.functional as F
torch.manual_seed(0)
for cin, cout, f in [(128, 128, 16), (64, 64, 32)]:
    b, t = 2, 500
    x = torch.randn(b, cin, f, t)
    w = torch.randn(cout, cin, 3, 3) / (cin * 9) ** .5
    ref = F.conv2d(F.pad(x.double(), (1, 1, 1, 1)), w.double())
    pc = ops.PackedConv(w.to(dev))
    for prec in ('f32', 'wino', 'bf16x3', 'winox3'):
        y, _, _ = ops.conv_fwd(x.to(dev), pc, pc.fwd(prec), precision=prec)
        e = (y.cpu().double() - ref).abs()
        print(f'{cin}->{cout} {prec}: max abs err {e.max():.2e}  rms err {e.pow(2).mean().sqrt():.2e}  (output rms {ref.pow(2).mean().sqrt():.2f})')
