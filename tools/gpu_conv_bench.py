"""Micro-benchmark (GPU box): time conv fwd / dgrad / wgrad of every FBCRNN layer shape at batch 32.
The library reads its switches from the environment once per process (README.md, "Environment switches"), so A/B
runs are separate processes (tools/kernel_ablation.sh builds variants of one kernel: PBSED_LIB=... python tools/gpu_conv_bench.py)"""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from pb_sed_amd import ops  # noqa: E402

DEV = 'cuda:0'
B, T = 32, 500
LAYERS = [  # cin, cout, F, kh, kw, pool, prologue
    (1, 16, 128, 3, 3, 0, 0), (16, 16, 128, 3, 3, 1, 1), (16, 32, 64, 3, 3, 0, 1), (32, 32, 64, 3, 3, 1, 1),
    (32, 64, 32, 3, 3, 0, 1), (64, 64, 32, 3, 3, 1, 1), (64, 128, 16, 3, 3, 0, 1), (128, 128, 16, 3, 3, 1, 1),
    (128, 256, 8, 3, 3, 0, 1), (2048, 256, 1, 1, 1, 0, 1), (256, 256, 1, 1, 3, 0, 1), (256, 256, 1, 1, 1, 0, 1),
    (256, 768, 1, 1, 1, 0, 0), (256, 10, 1, 1, 1, 0, 1),
]


def timeit(fn, n=5):
    fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n):
        fn()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n


def main():
    only = os.environ.get('ONLY')
    tot = [0., 0., 0.]
    print('knobs:', {k: v for k, v in os.environ.items() if k.startswith('PBSED_')})
    for cin, cout, f, kh, kw, pool, pro in LAYERS:
        if only and f'{cin}x{cout}' not in only.split(','):
            continue
        torch.manual_seed(0)
        x = torch.randn(B, cin, f, T, device=DEV)
        w = torch.randn(cout, cin, kh, kw, device=DEV) * .05
        bias = torch.zeros(cout, device=DEV)
        seq = torch.full((B,), T, dtype=torch.int32, device=DEV)
        scale = torch.ones(cin, device=DEV) if pro else None
        shift = torch.zeros(cin, device=DEV) if pro else None
        pc = ops.PackedConv(w)
        wp, wd = pc.fwd(), pc.dgrad()
        y, idx, _ = ops.conv_fwd(x, pc, wp, bias, scale, shift, True, seq, bool(pool), True)
        g = torch.randn_like(y)
        dw, db = torch.zeros_like(w), torch.zeros_like(bias)
        mean = torch.zeros(cin, device=DEV)
        fl = 2 * B * cout * cin * kh * kw * f * T / 1e9
        t_f = timeit(lambda: ops.conv_fwd(x, pc, wp, bias, scale, shift, True, seq, bool(pool), True))
        t_d = timeit(lambda: ops.conv_bwd_data(g, pc, wd, x.shape, idx, seq,
                                               bn=(x, mean, scale, scale, shift) if pro else None)) if cin > 1 else 0.
        t_w = timeit(lambda: ops.conv_bwd_weight(x, g, pc, dw, db, scale, shift, True, seq, idx))
        tot[0] += t_f; tot[1] += t_d; tot[2] += t_w
        print(f'{cin:5d}->{cout:4d} k{kh}x{kw} F{f:3d} pool{pool}: fwd {t_f:6.3f} ms {fl / t_f:6.1f} TF | '
              f'dgrad {t_d:6.3f} ms {fl / t_d if t_d else 0:6.1f} TF | wgrad {t_w:6.3f} ms {fl / t_w:6.1f} TF')
    print(f'total fwd {tot[0]:.3f}  dgrad {tot[1]:.3f}  wgrad {tot[2]:.3f} ms')


if __name__ == '__main__':
    main()
