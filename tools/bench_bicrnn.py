"""Config-3 measurement (BASELINE.json configs[2]): tag-conditioned strong_label BiCRNN train step, batch 32, 10 s clips,
1x MI355X.  python tools/bench_bicrnn.py [--conv-precision f32|bf16]   -> one JSON line (clips/s of the full step)."""
import argparse
import json
import os
import sys
import time

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument('--batch', type=int, default=32)
    ap.add_argument('--steps', type=int, default=20)
    ap.add_argument('--warmup', type=int, default=5)
    ap.add_argument('--conv-precision', default='f32', choices=['f32', 'bf16', 'bf16x3'])
    args = ap.parse_args()
    dev = 'cuda:0'
    torch.cuda.set_device(0)
    from pb_sed_amd import ops
    from pb_sed_amd.models import strong_label
    from pb_sed_amd.trainer import Trainer
    torch.manual_seed(0)
    model = strong_label.CRNN.build(tag_conditioning=True).to(dev)
    model.conv_precision = args.conv_precision
    trainer = Trainer(model, lr=5e-4, gradient_clipping=1e10)
    g = torch.Generator().manual_seed(2)
    b, t, k = args.batch, 500, 10
    wav = torch.randn(b, 160000, generator=g)
    wav = wav / wav.abs().max(-1, keepdim=True)[0]
    weak = (torch.rand(b, k, generator=g) < .25).float()
    weak[:, 0] = 1
    strong = torch.zeros(b, k, t)
    rng = np.random.RandomState(3)
    for i in range(b):
        for c in range(k):
            if weak[i, c]:
                on = rng.randint(0, 400)
                strong[i, c, on:on + rng.randint(10, 100)] = 1
    batch = {'audio_data': wav.to(dev), 'seq_len': [t] * b, 'weak_targets': weak.to(dev), 'strong_targets': strong.to(dev),
             'tag_condition': weak.to(dev)}
    for _ in range(args.warmup):
        trainer.step(batch)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(args.steps):
        rev = trainer.step(batch)
    torch.cuda.synchronize()
    dt = (time.perf_counter() - t0) / args.steps
    ops.check_gru_sync()
    print(json.dumps({'metric': 'tag-conditioned BiCRNN train step clips/s (batch 32, 10 s @ 16 kHz)', 'value': round(b / dt, 2),
                      'unit': 'clips/s', 'n_gpus': 1, 'ms_per_step': round(dt * 1e3, 3), 'dtype': args.conv_precision,
                      'loss': float(rev['loss'].item()), 'n_params': sum(p.numel() for p in model.parameters())}))


if __name__ == '__main__':
    main()
