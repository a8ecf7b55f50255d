"""Headline benchmark: 10 s @ 16 kHz clips/sec for one full FBCRNN train step (fused log-mel front-end
-> CNN2d/CNN1d -> fwd/bwd GRUs -> heads -> loss -> backward -> grad-norm clip -> Adam [+ RCCL gradient
all-reduce]) at batch 32 per GPU, fp32, synthetic waveforms / random-init weights.

    python bench.py --gpus 1 --steps K --warmup W
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N ... bench.py --gpus N --steps K --warmup W

Prints ONE JSON line on rank 0 (contract in the task statement) with two extra objects:
  roofline     - the dominant MFMA kernel launch (largest total time among conv launches), timed live with
                 HIP events on the launch stream inside the timed region; algorithmic FLOPs = 2*MACs of that
                 launch (SURVEY.md 8(d) / DESIGN.md "Rooflines"); peak = 157.3 TFLOP/s dense fp32 MFMA.
  cpu_baseline - the oracle (CPU restatement, kind "port") timed on this host on a bounded sample.
"""
import argparse
import json
import os
import sys
import time

import numpy as np
import torch

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

PEAK_FP32_MFMA_TFLOPS = 157.3      # /opt/skills/guides/MI355X_MICROARCH.md (dense f32-in MFMA)
PEAK_HBM_GBS = 8000.0
FWD_GFLOP_PER_CLIP = 11.82         # BASELINE.md section 2 (conv+GRU+heads+mel, forward)
TRAIN_GFLOP_PER_CLIP = 35.3


def synth_batch(b, device, n_samples=160000, k=10, t=500, seed=0):
    """SURVEY.md 8(d) synthetic inputs: randn waveforms (max-abs normalised), Bernoulli(.25) weak targets
    with >=1 positive, 1/8 of clips unlabeled (0.5), one segment per positive class."""
    g = torch.Generator().manual_seed(1234 + seed)
    wav = torch.randn(b, n_samples, generator=g)
    wav = wav / wav.abs().max(-1, keepdim=True)[0]
    rng = np.random.RandomState(1236 + seed)
    weak = (rng.rand(b, k) < .25).astype(np.float32)
    for i in range(b):
        if weak[i].sum() == 0:
            weak[i, rng.randint(k)] = 1
    bnd = np.zeros((b, k, t), np.float32)
    for i in range(b):
        for c in range(k):
            if weak[i, c]:
                on = rng.randint(0, 401)
                bnd[i, c, on:on + rng.randint(10, 101)] = 1
    for i in range(0, b, 8):
        weak[i] += (1 - weak[i]) * .5
        bnd[i] += (1 - bnd[i]) * .5
    return {'audio_data': wav.to(device), 'seq_len': [t] * b, 'weak_targets': torch.tensor(weak).to(device),
            'boundary_targets': torch.tensor(bnd).to(device)}


def pmc_traffic(kernel_tag):
    """HBM bytes per launch of the roofline kernel from the PMC passes committed under profiles/ (counters need their
    own rocprofv3 runs, tools/run_profiles.sh + tools/prof_summary.py); None if that layer was not profiled."""
    path = os.path.join(os.path.dirname(os.path.abspath(__file__)), 'profiles', 'roofline_traffic.json')
    try:
        row = json.load(open(path)).get(kernel_tag)
    except (OSError, ValueError):
        return None, None
    return (None, None) if row is None else (row['hbm_bytes_per_launch'], row['source'])


def cpu_baseline(batch=8, steps=2):
    """Oracle FBCRNN train step on the host cores (the checker, timed as the CPU baseline)."""
    from oracle import frontend as ofe, models as om
    torch.manual_seed(0)
    n = len(os.sched_getaffinity(0)) if hasattr(os, 'sched_getaffinity') else (os.cpu_count() or 1)
    n = min(n, 32)       # beyond ~32 threads the small GRU/conv ops of this model only get slower on CPU
    torch.set_num_threads(n)
    model = om.FBCRNN.build().train()
    opt = torch.optim.Adam(model.parameters(), lr=5e-4)
    b = synth_batch(batch, 'cpu')
    times = []
    for i in range(steps + 1):
        t0 = time.perf_counter()
        inp = {'stft': ofe.stft(b['audio_data']), 'seq_len': b['seq_len'], 'weak_targets': b['weak_targets'],
               'boundary_targets': b['boundary_targets']}
        opt.zero_grad()
        out = model(inp)
        loss = model.review(inp, out)['loss']
        loss.backward()
        torch.nn.utils.clip_grad_norm_(model.parameters(), 1e10)
        opt.step()
        times.append(time.perf_counter() - t0)
        print(f'[bench] cpu_baseline step {i}: {times[-1]:.2f} s on {n} threads', file=sys.stderr, flush=True)
        if sum(times) > 45.:                     # bounded sample: stop early on a slow host
            break
    dt = float(np.median(times[1:])) if len(times) > 1 else times[0]
    return {'value': round(batch / dt, 3), 'unit': 'clips/s', 'cores': n, 'kind': 'port',
            'sample': f'oracle (stock-PyTorch CPU restatement) FBCRNN train step incl. STFT, batch {batch} x 10 s '
                      f'clips, 1 warm-up + {steps} timed steps, median'}


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument('--gpus', type=int, default=1)
    ap.add_argument('--steps', type=int, default=20)
    ap.add_argument('--warmup', type=int, default=5)
    ap.add_argument('--batch', type=int, default=32, help='clips per GPU')
    ap.add_argument('--no-cpu-baseline', action='store_true')
    ap.add_argument('--conv-precision', default='f32', choices=['f32', 'bf16', 'bf16x3'],
                    help="operand format of the conv MFMAs; only 'f32' is the BASELINE configs[1] number")
    args = ap.parse_args()

    import torch.distributed as dist
    world = int(os.environ.get('WORLD_SIZE', '1'))
    rank = int(os.environ.get('RANK', '0'))
    local_rank = int(os.environ.get('LOCAL_RANK', '0'))
    torch.cuda.set_device(local_rank)
    device = f'cuda:{local_rank}'
    if world > 1:
        dist.init_process_group('nccl', device_id=torch.device(device))
    assert world == args.gpus, f'--gpus {args.gpus} but WORLD_SIZE={world}'

    from pb_sed_amd import _lib
    from pb_sed_amd.models import weak_label
    from pb_sed_amd.trainer import Trainer
    torch.manual_seed(0)
    model = weak_label.CRNN.build().to(device)
    model.conv_precision = args.conv_precision
    n_params = sum(p.numel() for p in model.parameters())
    trainer = Trainer(model, lr=5e-4, gradient_clipping=1e10)
    batch = synth_batch(args.batch, device, seed=rank)       # weak scaling: 32 clips per GPU

    for i in range(args.warmup):
        t_w = time.perf_counter()
        trainer.step(batch)
        torch.cuda.synchronize()
        if rank == 0:
            print(f'[bench] warm-up step {i}: {(time.perf_counter() - t_w) * 1e3:.1f} ms', file=sys.stderr, flush=True)
    torch.cuda.synchronize()
    if world > 1:
        dist.barrier()
    torch.cuda.synchronize()
    _lib.timing = []                                          # per-call HIP events on the launch stream
    ev_mode = os.environ.get('PBSED_BENCH_EVENTS', 'all' if os.environ.get('PBSED_BENCH_TABLE') else 'conv')
    # HIP events bracket only the conv / front-end launches by default (what `roofline` needs): an event pair
    # around each of the ~330 calls of a step costs ~1.3 ms/step of device idle time (PBSED_BENCH_EVENTS=all)
    _lib.timing_filter = {'all': None, 'conv': (lambda n: n.startswith(('pbsed_conv', 'pbsed_gru_stack_fwd')) or
                                                          n == 'pbsed_logmel_fwd')}[ev_mode]
    t0 = time.perf_counter()
    for _ in range(args.steps):
        review = trainer.step(batch)
    t_enq = time.perf_counter() - t0
    torch.cuda.synchronize()
    if world > 1:
        dist.barrier()
    torch.cuda.synchronize()
    dt = time.perf_counter() - t0
    events, _lib.timing = _lib.timing, None
    if world > 1:
        tt = torch.tensor([dt], device=device, dtype=torch.float64)
        dist.all_reduce(tt, op=dist.ReduceOp.MAX)
        dt = tt.item()
    loss = float(review['loss'].item())
    # host -> device hand-over of one batch (pinned, PCIe), outside the timed region: `value` is HBM-resident
    host_batch = {k: v.cpu().pin_memory() for k, v in batch.items() if isinstance(v, torch.Tensor)}
    torch.cuda.synchronize()
    t_h = time.perf_counter()
    for _ in range(5):
        for k, v in host_batch.items():
            batch[k].copy_(v, non_blocking=True)
    torch.cuda.synchronize()
    h2d_ms = (time.perf_counter() - t_h) / 5 * 1e3
    from pb_sed_amd import ops as _ops
    _ops.check_gru_sync()

    if rank == 0:
        # ---- per-call timing -> dominant MFMA launch
        agg = {}
        for name, tag, flops, byts, e0, e1 in events:
            ms = e0.elapsed_time(e1)
            a = agg.setdefault((name, tag), [0.0, 0, flops, byts])
            a[0] += ms
            a[1] += 1
        conv = {k: v for k, v in agg.items() if k[0].startswith('pbsed_conv')}
        (dname, dtag), (tot_ms, cnt, flops, _) = max(conv.items(), key=lambda kv: kv[1][0])
        avg_ms = tot_ms / cnt
        achieved = flops / (avg_ms * 1e-3) / 1e12
        fe = [v for k, v in agg.items() if k[0] == 'pbsed_logmel_fwd']
        by_family = {}
        for (name, tag), (ms, c, fl, by) in agg.items():
            by_family[name] = by_family.get(name, 0.0) + ms / args.steps
        ms_step = dt / args.steps * 1e3
        clips = args.batch * world
        out = {
            'metric': '10s@16kHz clips/sec (train step) FBCRNN batch32',
            'value': round(clips / (dt / args.steps), 2), 'unit': 'clips/s', 'n_gpus': world,
            'steps': args.steps, 'warmup': args.warmup, 'ms_per_step': round(ms_step, 3),
            'higher_is_better': True, 'scaling': 'weak', 'vs_baseline': None,
            'dtype': {'f32': 'f32', 'bf16': 'bf16 (conv MFMAs; fp32 accumulate/BN/GRU)',
                      'bf16x3': 'bf16x3 (fp32 operands split into 3 bf16 terms, 6 MFMAs per product, fp32 accumulate)'}[args.conv_precision],
            'data': 'synthetic (randn waveforms, random-init weights)',
            'config': {'workload': 'FBCRNN weak_label_crnn.training batch 32/GPU fp32, 10 s 16 kHz clips '
                                   '(BASELINE.json configs[1]); full train step incl. fused log-mel front-end, '
                                   'loss, backward, grad-norm clip, Adam' + (', RCCL grad all-reduce' if world > 1 else ''),
                       'global_batch': clips, 'n_params': n_params, 'parallelism': f'dp{world}'},
            'roofline': {'bound': 'mfma', 'achieved': round(achieved, 2), 'peak': PEAK_FP32_MFMA_TFLOPS,
                         'unit': 'TFLOP/s', 'frac': round(achieved / PEAK_FP32_MFMA_TFLOPS, 4),
                         'traffic': pmc_traffic(f'{dname} {dtag}')[0], 'traffic_unit': 'HBM bytes per launch (PMC)',
                         'traffic_source': pmc_traffic(f'{dname} {dtag}')[1],
                         'kernel': f'{dname} {dtag}', 'avg_ms': round(avg_ms, 4), 'launches': cnt,
                         'flops_per_launch': flops,
                         # MFMA instructions actually issued: Winograd F(4,3) executes half the direct convolution's MACs
                         'achieved_executed': round(achieved / (2 if dtag.endswith('wino') else 1), 2),
                         'frac_executed': round(achieved / (2 if dtag.endswith('wino') else 1) / PEAK_FP32_MFMA_TFLOPS, 4),
                         'note': ('algorithmic FLOPs = 2*MACs of the direct 3x3 convolution; this launch runs the '
                                  'Winograd-F(4,3) kernel, which executes half of those multiplications on the MFMA '
                                  f'pipe ({achieved / 2:.1f} TFLOP/s executed): frac = algorithmic / peak can exceed 1, frac_executed is the '
                                  'MFMA-pipe utilisation') if dtag.endswith('wino') else
                                 'algorithmic FLOPs = 2*MACs, all executed on the MFMA pipe'},
            'step_mfma': {'algorithmic_tflop_per_step': round(TRAIN_GFLOP_PER_CLIP * args.batch / 1e3, 4),
                          'achieved_tflops_per_gpu': round(TRAIN_GFLOP_PER_CLIP * args.batch / 1e3 / (ms_step * 1e-3), 2),
                          'frac_of_fp32_mfma_peak': round(TRAIN_GFLOP_PER_CLIP * args.batch / 1e3 / (ms_step * 1e-3) / PEAK_FP32_MFMA_TFLOPS, 4)},
            # BASELINE north_star target: MFMA roofline of the conv + GRU forward pass (algorithmic forward FLOPs over the
            # summed event time of the forward conv / GRU-scan / front-end launches)
            'forward_conv_gru': (lambda ms: {'ms_per_step': round(ms, 3),
                                             'algorithmic_tflop': round(FWD_GFLOP_PER_CLIP * args.batch / 1e3, 4),
                                             'achieved_tflops': round(FWD_GFLOP_PER_CLIP * args.batch / 1e3 / (ms * 1e-3), 2),
                                             'frac_of_fp32_mfma_peak': round(FWD_GFLOP_PER_CLIP * args.batch / 1e3 / (ms * 1e-3)
                                                                             / PEAK_FP32_MFMA_TFLOPS, 4)})(
                sum(v for k, v in by_family.items() if k.startswith(('pbsed_conv_fwd', 'pbsed_gru_stack_fwd', 'pbsed_logmel')))),
            'ms_per_step_by_entry_point': {k: round(v, 3) for k, v in sorted(by_family.items(), key=lambda kv: -kv[1])},
            'host_enqueue_ms_per_step': round(t_enq / args.steps * 1e3, 3),
            'h2d': {'ms_per_batch': round(h2d_ms, 3), 'bytes': int(sum(v.numel() * v.element_size() for v in host_batch.values())),
                    'clips_per_s_if_serialised': round(clips / (dt / args.steps + h2d_ms * 1e-3), 2)},
            'loss': loss,
        }
        if fe:
            fe_ms = fe[0][0] / fe[0][1]
            gbs = 896000.0 * args.batch / (fe_ms * 1e-3) / 1e9
            out['frontend_hbm'] = {'bound': 'hbm', 'achieved': round(gbs, 1), 'peak': PEAK_HBM_GBS, 'unit': 'GB/s',
                                   'frac': round(gbs / PEAK_HBM_GBS, 4), 'avg_ms': round(fe_ms, 4)}
        print('[bench] gpu: ' + json.dumps({k: out[k] for k in ('value', 'ms_per_step', 'roofline')}),
              file=sys.stderr, flush=True)
        if os.environ.get('PBSED_BENCH_TABLE'):
            print('[bench] per-call table (ms/step, calls/step, TFLOP/s):', file=sys.stderr)
            for (name, tag), (ms, c, fl, _) in sorted(agg.items(), key=lambda kv: -kv[1][0])[:45]:
                tf = fl * c / (ms * 1e-3) / 1e12 if fl else 0.
                print(f'   {ms / args.steps:8.3f} {c / args.steps:5.1f} {tf:7.1f}  {name[6:]} {tag}', file=sys.stderr)
        if world == 1 and not args.no_cpu_baseline:
            out['cpu_baseline'] = cpu_baseline()
        print(json.dumps(out))
    if world > 1:
        dist.destroy_process_group()


if __name__ == '__main__':
    main()
