"""Time pbsed_gru_wgrad_multi on the shapes of the bench configs (historic: round 3 compared kernel forms through PBSED_GRU_WGRAD_PC; the switch is read once).
c2: 8 x [768 x 256] over 16 000 rows (FBCRNN 2 x 2 stack), c3: 4 x [768 x 256] + ... per BiGRU layer, deep: 8 x [1536 x 512]."""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from pb_sed_amd import ops  # noqa: E402

DEV = torch.device('cuda:0')
SHAPES = {'c2': (500, 32, 768, [256] * 8), 'c3': (500, 32, 768, [256] * 4), 'deep': (500, 32, 1536, [512] * 8),
          'b48': (500, 48, 768, [256] * 8)}


def main():
    names = sys.argv[1:] or ['c2', 'c3', 'deep']
    for name in names:
        t, b, g, ks = SHAPES[name]
        for precision in ('f32', 'bf16'):
            torch.manual_seed(0)
            dg = [torch.randn(t, b, g, device=DEV) for _ in ks]
            x = [torch.randn(t, b, k, device=DEV) for k in ks]
            dw = [torch.zeros(g, k, device=DEV) for k in ks]
            db = [torch.zeros(g, device=DEV) for _ in ks]
            shift = [(-1, 0, 1)[i % 3] for i in range(len(ks))]
            for _ in range(3):
                ops.gru_wgrad(dg, x, shift, dw, db, precision=precision)
            torch.cuda.synchronize()
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            n = 20
            torch.cuda._sleep(int(60e6))          # ~25 ms of spinning: the launches below queue up behind it (device time, not enqueue time)
            e0.record()
            for _ in range(n):
                ops.gru_wgrad(dg, x, shift, dw, db, precision=precision)
            e1.record()
            torch.cuda.synchronize()
            ms = e0.elapsed_time(e1) / n
            fl = 2. * t * b * g * sum(ks)
            print(f'{name} {precision}: {ms * 1e3:.1f} us  {fl / ms / 1e9:.0f} TFLOP/s fp32-equivalent'
                  '', flush=True)


if __name__ == '__main__':
    main()
