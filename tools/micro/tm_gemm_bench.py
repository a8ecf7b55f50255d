"""Time the time-major projections of the headline step against the conv + transpose launches they replace.
gpurun: python tools/micro/tm_gemm_bench.py"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
from pb_sed_amd import ops

def timed(fn, n=30):
    for _ in range(3):
        fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n):
        fn()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n * 1e3

t, b = 500, 32
for ks, n in (([256], 768), ([768, 768], 256), ([512], 768), ([768, 768], 512)):
    xs = [torch.randn(t, b, k, device='cuda') for k in ks]
    ws = [torch.randn(n, k, device='cuda') * .05 for k in ks]
    bias = torch.randn(n, device='cuda')
    gf = 2 * t * b * n * sum(ks) / 1e9
    for prec in ('f32', 'bf16'):
        us = timed(lambda: ops.tm_gemm(xs, ws, bias, precision=prec))
        print(f'tm_gemm {ks}->{n} {prec}: {us:.1f} us {gf / us * 1e3:.1f} TFLOP/s')
# what it replaces (fp32): conv k=1 on [B,C,T] + transpose of the result
x = torch.randn(b, 256, t, device='cuda')
w = torch.randn(768, 256, 1, device='cuda') * .05
pc = ops.PackedConv(w)
wp = pc._pack(0)
us = timed(lambda: ops.bct_to_tbc(ops.conv_fwd(x, pc, wp, torch.zeros(768, device='cuda'))[0]))
print(f'conv_fwd 256->768 + bct_to_tbc: {us:.1f} us')
