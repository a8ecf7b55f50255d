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
for cin, cout, f in [(128, 128, 16), (128, 256, 8), (64, 64, 32), (64, 128, 16), (32, 64, 32)]:
    b, t = 32, 500
    x = torch.randn(b, cin, f, t, device=dev)
    w = torch.randn(cout, cin, 3, 3, device=dev) / (cin * 9) ** .5
    pc = ops.PackedConv(w)
    sc, sh = torch.rand(cin, device=dev) + .5, torch.randn(cin, device=dev) * .3
    seq = torch.full((b,), t, dtype=torch.int32, device=dev)
    res = []
    for prec in ('f32', 'wino', 'bf16x3'):
        wp = pc.fwd(prec); wd = pc.dgrad(prec)
        ms = tm(lambda: ops.conv_fwd(x, pc, wp, scale=sc, shift=sh, seq_len=seq, want_stats=True, precision=prec))
        g = torch.randn(b, cout, f, t, device=dev)
        msd = tm(lambda: ops.conv_bwd_data(g, pc, wd, x.shape, None, seq, precision=prec))
        res.append((ms, msd))
    fl = 2 * b * cout * cin * 9 * f * t / 1e9
    print(f'{cin}->{cout} F{f}: fwd direct {res[0][0]:.3f} ms ({fl/res[0][0]:.0f} TF)  wino {res[1][0]:.3f} ms ({fl/res[1][0]:.0f} TF eff) x{res[0][0]/res[1][0]:.2f} | '
          f'dgrad direct {res[0][1]:.3f}  wino {res[1][1]:.3f} x{res[0][1]/res[1][1]:.2f} | bf16x3 fwd {res[2][0]:.3f} dgrad {res[2][1]:.3f}')

# rounding error of both kernels against an fp64 convolution (small batch)
import torch.nn.functional as F
torch.manual_seed(0)
for cin, cout, f in [(128, 128, 16), (64, 64, 32)]:
    b, t = 2, 500
    x = torch.randn(b, cin, f, t)
    w = torch.randn(cout, cin, 3, 3) / (cin * 9) ** .5
    ref = F.conv2d(F.pad(x.double(), (1, 1, 1, 1)), w.double())
    pc = ops.PackedConv(w.to(dev))
    for prec in ('f32', 'wino', 'bf16x3'):
        y, _, _ = ops.conv_fwd(x.to(dev), pc, pc.fwd(prec), precision=prec)
        e = (y.cpu().double() - ref).abs()
        print(f'{cin}->{cout} {prec}: max abs err {e.max():.2e}  rms err {e.pow(2).mean().sqrt():.2e}  (output rms {ref.pow(2).mean().sqrt():.2f})')
