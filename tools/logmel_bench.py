"""Device time of the fused log-mel front-end on a bench batch (32 x 10 s @ 16 kHz); PBSED_LOGMEL=1 selects the first form."""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from pb_sed_amd import ops  # noqa: E402
from pb_sed_amd.modules import get_fbanks, num_frames  # noqa: E402

DEV = torch.device('cuda:0')


def main():
    b, n = 32, 160000
    t = num_frames(n)
    wav = torch.randn(b, n, device=DEV)
    tables = ops.LogMelTables(get_fbanks(16000, 1024, 128), DEV)
    mean, inv_std = torch.zeros(128, device=DEV), torch.ones(128, device=DEV)
    stats = torch.zeros(32 * 128 * 2, dtype=torch.float64, device=DEV)
    for name, kw in (('plain', {}), ('stats', {'stats': stats})):
        for _ in range(3):
            ops.logmel_fwd(wav, tables, mean, inv_std, t, **kw)
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        torch.cuda._sleep(int(60e6))
        e0.record()
        for _ in range(20):
            ops.logmel_fwd(wav, tables, mean, inv_std, t, **kw)
        e1.record()
        torch.cuda.synchronize()
        us = e0.elapsed_time(e1) / 20 * 1e3
        nbytes = b * (n * 4 + 128 * t * 4)
        print(f'wav->logmel {name}: {us:.1f} us  {nbytes / us / 1e3:.0f} GB/s of algorithmic traffic (form {os.environ.get("PBSED_LOGMEL", "2")})', flush=True)


if __name__ == '__main__':
    main()
