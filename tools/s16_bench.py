"""Time the few-channel 3x3 layers (csrc/conv_s16.hip, bf16x3) against the direct fp32-MFMA kernel at batch 32, T = 500:
forward and data gradient of 16->16 (+ pool), 16->32, 11->16."""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from pb_sed_amd import ops  # noqa: E402

DEV, B, T = 'cuda:0', 32, 500


def timeit(fn, n=10):
    fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n):
        fn()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n * 1e3


for cin, cout, f, pool in ((16, 16, 128, 1), (16, 32, 64, 0), (11, 16, 128, 0)):
    torch.manual_seed(0)
    x = torch.randn(B, cin, f, T, device=DEV)
    w = torch.randn(cout, cin, 3, 3, device=DEV) * .05
    bias = torch.zeros(cout, device=DEV)
    seq = torch.full((B,), T, dtype=torch.int32, device=DEV)
    scale, shift, mean = torch.ones(cin, device=DEV), torch.zeros(cin, device=DEV), torch.zeros(cin, device=DEV)
    pc = ops.PackedConv(w)
    y, idx, _ = ops.conv_fwd(x, pc, pc.fwd(), bias, scale, shift, True, seq, bool(pool), True)
    g = torch.randn_like(y)
    mb = (x.numel() + y.numel()) * 4 / 1e6
    row = f'{cin:2d}->{cout:2d} F{f:3d} pool{pool} ({mb:.0f} MB in + out):'
    for prec in ('f32', 's16x3'):
        wp = pc.fwd(prec)
        us = timeit(lambda: ops.conv_fwd(x, pc, wp, bias, scale, shift, True, seq, bool(pool), True, precision=prec))
        row += f'  fwd {prec} {us:6.1f} us ({mb / us * 1e3 / 1e3:.2f} TB/s)'
    if cout <= 16:
        for prec in ('f32', 's16x3'):
            wd = pc.dgrad(prec)
            us = timeit(lambda: ops.conv_bwd_data(g, pc, wd, x.shape, idx, seq, bn=(x, mean, scale, scale, shift), precision=prec))
            row += f'  dgrad {prec} {us:6.1f} us'
    print(row, flush=True)
