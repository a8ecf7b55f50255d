"""Config-5 measurement (BASELINE.json configs[4]): strong_label_crnn_inference-style ensemble - FBCRNN taggers ->
tags -> tag-conditioned BiCRNN detectors, 5 models, batch 64, with the GPU post-processing chain (mean, mask,
per-class median filters for 3 hyper-parameter variants, tag masking) and event extraction.

    python tools/bench_inference.py [--gpus N]        (torchrun for N > 1: clips are sharded over ranks, no collective)

Prints one JSON line: clips/s of the whole pipeline (waveform resident in HBM -> event lists on the host)."""
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
    ap.add_argument('--gpus', type=int, default=1)
    ap.add_argument('--batch', type=int, default=64)
    ap.add_argument('--iters', type=int, default=5)
    args = ap.parse_args()
    world = int(os.environ.get('WORLD_SIZE', '1'))
    rank = int(os.environ.get('RANK', '0'))
    local = int(os.environ.get('LOCAL_RANK', '0'))
    torch.cuda.set_device(local)
    dev = f'cuda:{local}'
    if world > 1:
        import torch.distributed as dist
        dist.init_process_group('nccl', device_id=torch.device(dev))
    from pb_sed_amd import inference as inf
    from pb_sed_amd.models import strong_label, weak_label
    torch.manual_seed(0)
    taggers = [weak_label.CRNN.build() for _ in range(2)]
    detectors = [strong_label.CRNN.build(tag_conditioning=True) for _ in range(3)]
    classes = [f'class{i}' for i in range(10)]
    g = torch.Generator().manual_seed(1)
    wav = torch.randn(args.batch, 160000, generator=g)
    wav = (wav / wav.abs().max(-1, keepdim=True)[0]).to(dev)
    ids = [f'clip{i}' for i in range(args.batch)]
    batch = {'audio_data': wav, 'seq_len': [500] * args.batch, 'example_id': ids}
    medfilt = np.array([[1, 3, 5, 7, 9, 11, 21, 31, 41, 51], [11] * 10, [51] * 10])
    ts = np.round(np.arange(0, 100000) * .02, 6)

    def run():
        tag_scores = inf.tagging(taggers, [dict(batch)], dev, rank=rank, world_size=world)
        tags = {a: (s[0] > .5).astype(np.float32) for a, s in tag_scores.items()}
        my_ids = sorted(tags)
        sub = {'audio_data': wav[rank * len(my_ids):(rank + 1) * len(my_ids)], 'seq_len': [500] * len(my_ids),
               'example_id': ids[rank * len(my_ids):(rank + 1) * len(my_ids)]}
        cond = torch.tensor(np.stack([tags[a] for a in sub['example_id']])).to(dev)
        sed = inf.sound_event_detection(detectors, [dict(sub, tag_condition=cond)], dev, medfilt_length=medfilt,
                                        apply_mask=True, masks=tags)
        events = inf.scores_to_event_list({a: s[0] for a, s in sed.items()}, .5, classes, ts, device=dev)
        return len(events)

    run()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(args.iters):
        n = run()
    torch.cuda.synchronize()
    dt = (time.perf_counter() - t0) / args.iters
    if world > 1:
        tt = torch.tensor([dt], device=dev, dtype=torch.float64)
        dist.all_reduce(tt, op=dist.ReduceOp.MAX)
        dt = tt.item()
    if rank == 0:
        print(json.dumps({'metric': 'ensemble inference clips/s (2 FBCRNN taggers + 3 tag-conditioned BiCRNN detectors, '
                                    '3 median-filter variants, event extraction)', 'value': round(args.batch / dt, 1),
                          'unit': 'clips/s', 'n_gpus': world, 'batch': args.batch, 'ms_per_batch': round(dt * 1e3, 2),
                          'dtype': 'f32', 'clips_per_rank': n}))
    if world > 1:
        dist.destroy_process_group()


if __name__ == '__main__':
    main()
