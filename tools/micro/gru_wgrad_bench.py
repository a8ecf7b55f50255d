"""Time the batched GRU weight-gradient launch of the headline step (8 GEMMs [768 x 256] over 16000 rows) with the bf16x3
kernel (fp32 path of pbsed_gru_wgrad) and plain bf16 operands.  gpurun: python tools/micro/gru_wgrad_bench.py"""
import os, sys, subprocess
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
if len(sys.argv) == 1:
    for prec in ('f32', 'bf16'):
        subprocess.run([sys.executable, __file__, prec])
    sys.exit(0)
import torch
from pb_sed_amd import ops
prec = sys.argv[1]
t, b, g = 500, 32, 768
for k, n in ((256, 8), (512, 4), (256, 2), (512, 2)):
    dg = [torch.randn(t, b, g, device='cuda') for _ in range(n)]
    x = [torch.randn(t, b, k, device='cuda') for _ in range(n)]
    dw = [torch.zeros(g, k, device='cuda') for _ in range(n)]
    db = [torch.zeros(g, device='cuda') for _ in range(n)]
    fn = lambda: ops.gru_wgrad(dg, x, [0, -1] * (n // 2), dw, db, precision=prec)
    for _ in range(3):
        fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(20):
        fn()
    e1.record()
    torch.cuda.synchronize()
    ms = e0.elapsed_time(e1) / 20
    print(f'{prec:5s} {n} x [{g} x {k}] x {t * b}: {ms:.3f} ms  {2 * n * t * b * g * k / ms / 1e9:.1f} TFLOP/s')
