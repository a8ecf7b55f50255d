"""Time the one-layer bf16 scans (2 directions, B 32, H 256, T 500: one BiGRU layer of BASELINE configs[2]) - forward and BPTT.
PBSED_LIB selects a build variant, PBSED_GRU_LOCAL=0 the exchanging (granule) kernels.  gpurun: python tools/micro/gru_local_bench.py"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np
import torch
from pb_sed_amd import ops
DEV = 'cuda:0'
t, b, h, nch = 500, int(os.environ.get('B', 32)), 256, 2
torch.manual_seed(0)
seq = torch.full((b,), t, dtype=torch.int32, device=DEV)
gi0 = [torch.randn(t, b, 3 * h, device=DEV) * .5 for _ in range(nch)]
mk = lambda *s: torch.randn(*s, device=DEV) * h ** -.5
w_hh = [mk(3 * h, h) for _ in range(nch)]
b_hh = [mk(3 * h) for _ in range(nch)]
dy = [torch.randn(t, b, h, device=DEV) for _ in range(nch)]
w_hh_t = [ops.transpose2d(w) for w in w_hh]


def run(n):
    for _ in range(n):
        hs, save = ops.gru_stack_fwd(gi0, [None] * nch, [None] * nch, w_hh, b_hh, [False, True], seq, 1, save=True, precision='bf16')
    return hs, save


hs, save = run(3)
torch.cuda.synchronize()
e0, e1, e2 = (torch.cuda.Event(enable_timing=True) for _ in range(3))
e0.record()
run(10)
e1.record()
for _ in range(10):
    ops.gru_stack_bwd(w_hh_t, [None] * nch, hs, save, dy, [False, True], seq, 1, precision='bf16')
e2.record()
torch.cuda.synchronize()
ops.check_gru_sync()
f, g = e0.elapsed_time(e1) / 10, e1.elapsed_time(e2) / 10
print(f'{os.environ.get("PBSED_LIB", "default").split("/")[-1]} LOCAL={os.environ.get("PBSED_GRU_LOCAL", "1")}: forward {f:.3f} ms ({f * 1e3 / t:.2f} us/step)  BPTT {g:.3f} ms ({g * 1e3 / t:.2f} us/step)')
