"""Benchmarks of the pb_sed hot path on MI355X.  Default = the headline metric (BASELINE.json configs[1]):

    10 s @ 16 kHz clips/sec for one full FBCRNN train step (fused log-mel front-end -> CNN2d/CNN1d -> fwd/bwd GRUs ->
    heads -> loss -> backward -> grad-norm clip -> Adam [+ RCCL gradient all-reduce]) at batch 32 per GPU, fp32.

    python bench.py --gpus 1 --steps K --warmup W                  [--config c2|c3|c5]
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N ... bench.py --gpus N --steps K --warmup W

  --config c2  FBCRNN train step, batch 32/GPU, fp32                     (BASELINE.json configs[1], configs[3] at N > 1)
  --config c3  tag-conditioned BiCRNN train step, batch 32/GPU, bf16     (BASELINE.json configs[2])
  --config c5  5-model ensemble inference (2 FBCRNN taggers -> tags -> 3 tag-conditioned BiCRNN detectors -> GPU
               post-processing -> event lists), batch 64, clips sharded over ranks   (BASELINE.json configs[4])

Rank 0 prints ONE JSON line (contract in the task statement).  The timed region holds NO event pairs: K steps over >= 2
distinct HBM-resident batches between barrier + synchronize pairs, max over ranks.  Per-kernel figures come from a second,
separate pass with HIP events on the launch stream:
  roofline      the DOMINANT launch of the step (largest time per launch): a persistent GRU scan in every config since round 3
                (latency-bound: its flops over its duration against the ceiling of its operand type); else the conv launch.
  roofline_conv the conv launch with the largest total time: `achieved` / `frac` = EXECUTED flops on the MFMA pipe (a
                Winograd-F(4,3) launch executes half the multiplications of the direct convolution) over the dense peak of
                the operand type, always <= 1; `achieved_algorithmic` = 2*MACs of the direct convolution over the same time.
                (Named `roofline` when no scan is ahead of it.)
  roofline_gru  forward / BPTT persistent scans: recurrent + layer-boundary projection flops over the scan's duration.
  frontend_hbm  896 000 B per clip over the front-end kernel's duration against 8 TB/s.
  cpu_baseline  the oracle (CPU restatement, kind "port") on this host: same workload at batch 32 and at batch 16
                (the stand-in for configs[0]), 1 warm-up + 3 timed steps, per-stage breakdown, CPU model string.
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

PEAK_TFLOPS = {'f32': 157.3, 'bf16': 2500.0}      # /opt/skills/guides/MI355X_MICROARCH.md: dense MFMA, f32 / bf16 operands
PEAK_HBM_GBS = 8000.0
XGMI_LINK_GBS, XGMI_LINKS = 153.0, 7
FWD_GFLOP = {'c2': 11.82, 'c3': 12.34}
# what 'dtype' means in detail (the value itself stays the plain type name)
ARITHMETIC_NOTE = {
    'f32': 'fp32 storage, accumulation, state and results throughout.  Every product with >= 32 channels on either side - the '
           '3x3 conv layers (Winograd F(4,3) along t) and their weight gradients, the Conv1d layers, GRU scans, GRU weight '
           'gradients, the projections around the scans - is formed on the bf16 MFMA from EXACT three-way bf16 splits of both '
           'fp32 operands (x = hi + mid + lo by truncation, 8 + 8 + 8 significant bits; the six part products above 2^-24 '
           'accumulated in fp32): fp32-class results - rms error vs an fp64 convolution 9.4e-7 (Winograd bf16x3) / 5.0e-7 (direct '
           'bf16x3) against 1.1e-6 / 5.9e-7 of the fp32-MFMA kernels on the same layers - held to the same 1e-4 logit / gradient '
           'parity tests.  The 1- and 16-channel layers run on the fp32 MFMA',
    'bf16': 'bf16 MFMA operands (rounded to nearest even while staged) in every conv / projection / scan / weight-gradient '
            'product with >= 32 channels; fp32 accumulation, BN, GRU state, losses, master weights and optimiser',
    'bf16x3': 'every conv product from exact three-way bf16 operand splits on the bf16 MFMA (fp32-class); the rest as f32',
}            # forward per clip (train step = 3x)


# ------------------------------------------------------------------------------------------------ synthetic data
def synth_batch(b, device, n_samples=160000, k=10, t=500, seed=0, kind='c2'):
    """SURVEY.md 8(d) synthetic inputs: randn waveforms (max-abs normalised), Bernoulli(.25) weak targets with >= 1
    positive, 1/8 of the clips unlabeled (0.5), one segment per positive class; c3: strong targets + tag condition."""
    g = torch.Generator().manual_seed(1234 + seed)
    wav = torch.randn(b, n_samples, generator=g)
    wav = wav / wav.abs().max(-1, keepdim=True)[0]
    rng = np.random.RandomState(1236 + seed)
    weak = (rng.rand(b, k) < .25).astype(np.float32)
    for i in range(b):
        if weak[i].sum() == 0:
            weak[i, rng.randint(k)] = 1
    tgt = np.zeros((b, k, t), np.float32)
    for i in range(b):
        for c in range(k):
            if weak[i, c]:
                on = rng.randint(0, 401)
                tgt[i, c, on:on + rng.randint(10, 101)] = 1
    batch = {'audio_data': wav.to(device), 'seq_len': [t] * b}
    if kind == 'c3':
        batch.update(weak_targets=torch.tensor(weak).to(device), strong_targets=torch.tensor(tgt).to(device),
                     tag_condition=torch.tensor(weak).to(device))
        return batch
    for i in range(0, b, 8):
        weak[i] += (1 - weak[i]) * .5
        tgt[i] += (1 - tgt[i]) * .5
    batch.update(weak_targets=torch.tensor(weak).to(device), boundary_targets=torch.tensor(tgt).to(device))
    return batch


def pmc_traffic(kernel_tag):
    """HBM bytes per launch of the roofline kernel from the PMC passes committed under profiles/ (counters need their
    own rocprofv3 runs, tools/run_profiles.sh + tools/prof_summary.py); None if that layer was not profiled.  The persistent
    scans (keys 'gru_granule_fwd' / 'gru_granule_bwd'): L2 request bytes of the TCC pass - they exchange through L2 / the
    fabric, HBM is not what they move."""
    try:
        row = json.load(open(os.path.join(ROOT, 'profiles', 'roofline_traffic.json'))).get(kernel_tag)
    except (OSError, ValueError):
        return None, None
    if row is None:
        return None, None
    return row.get('hbm_bytes_per_launch', row.get('l2_request_bytes_per_launch')), row['source']


def scan_polled_bytes(kind, nch, nl, t, b, h):
    """Bytes the polling loads of ONE persistent scan launch request when every first look succeeds (tools/gru_scan_prof.py: no
    missed polls at the tuned delays): every ring / projection workgroup of 16 hidden units x 16 batch rows polls the whole
    16 x H state tile of its source per step (tagged 4-byte words), the gate threads of the upper layers their projected
    inputs (forward: 3 words, BPTT: 1 word per (row, unit)).  New state per step is 1 / (H / 16) of that."""
    tiles = (b + 15) // 16
    groups = nch * (2 * nl - 1)                        # rings + layer-boundary projection groups
    state = groups * (h // 16) * tiles * 16 * h * 4
    own = nch * (nl - 1) * tiles * 16 * h * 4 * (3 if kind == 'forward_scan' else 1)
    return int(t * (state + own))


def _pmc_mfma_busy(kernel_prefixes):
    """MFMA-pipe busy share of kernels of the headline step from the committed counter pass (profiles/r05_pmc_mfma.csv:
    SQ_VALU_MFMA_BUSY_CYCLES per SIMD over the kernel's clocks, tools/run_profiles.sh step 5); {} if the file is not there."""
    import csv
    path = os.path.join(ROOT, 'profiles', 'r05_pmc_mfma.csv')
    out = {}
    try:
        for r in csv.DictReader(open(path)):
            for pre in kernel_prefixes:
                if pre in r['kernel']:
                    key = [c for c in r if c.startswith('mfma_pipe_busy_frac')][0]
                    out[pre] = float(r[key])
    except (OSError, ValueError, IndexError, KeyError):
        return {}
    if out:
        out['source'] = 'profiles/r05_pmc_mfma.csv (rocprofv3 --pmc SQ_VALU_MFMA_BUSY_CYCLES GRBM_GUI_ACTIVE ..., own pass)'
    return out


def host_cpu():
    model = 'unknown'
    try:
        for line in open('/proc/cpuinfo'):
            if line.startswith('model name'):
                model = line.split(':', 1)[1].strip()
                break
    except OSError:
        pass
    aff = len(os.sched_getaffinity(0)) if hasattr(os, 'sched_getaffinity') else (os.cpu_count() or 1)
    return model, os.cpu_count() or 1, aff


# ------------------------------------------------------------------------------------------------ CPU baselines
def _cpu_threads():
    _, _, aff = host_cpu()
    n = min(aff, 32)        # measured: beyond ~32 threads the small GRU / conv ops of this model only get slower on CPU
    torch.set_num_threads(n)
    return n


_ALLOCATOR = None


def _tune_host_allocator():
    """glibc malloc as a CPU training run would set it up (MALLOC_MMAP_MAX_=0, MALLOC_TRIM_THRESHOLD_ / MALLOC_TOP_PAD_ large; here through
    mallopt, because the process is already running): big tensors are recycled from the heap instead of being mmap'ed, page-faulted
    and unmapped on every use.  Why it matters for THIS baseline (measured): torch's CPU GRU over a packed
    sequence runs its recurrence as 500 autograd slices per (layer, direction) and every SliceBackward materialises a zero buffer of
    the whole [T*B, 3H] gradient - 49 MB at batch 32, 2 000 of them per train step.  glibc's dynamic mmap threshold stops at 32 MB:
    the 24.5 MB buffers of batch 16 are recycled from the heap, the 49 MB ones of batch 32 are mapped anew each time - 19 M page
    faults and 70 s of kernel time per step on 8 cores (24.8 s per step against 8.0 s with this setting; batch 16: 5.4 -> 4.5 s).
    That cliff, not thread oversubscription, made batch 32 cost 4.4x batch 16 per clip in round 5 and the figure move 5x between
    hosts.  The tuned allocator is the FASTER baseline, i.e. the less flattering one for the GPU."""
    global _ALLOCATOR
    if _ALLOCATOR is None:
        try:
            import ctypes
            libc = ctypes.CDLL('libc.so.6')
            ok = [libc.mallopt(-4, 0),                      # M_MMAP_MAX = 0: no mmap'ed chunks
                  libc.mallopt(-1, 2 ** 31 - 1),            # M_TRIM_THRESHOLD: the heap top is not given back
                  libc.mallopt(-2, 256 << 20)]              # M_TOP_PAD: grow the heap 256 MB at a time
            _ALLOCATOR = 'glibc mallopt(M_MMAP_MAX=0, M_TRIM_THRESHOLD=2^31-1, M_TOP_PAD=256MB)' if all(ok) else 'default (mallopt refused)'
        except (OSError, AttributeError):
            _ALLOCATOR = 'default (no glibc mallopt)'
    return _ALLOCATOR


def cpu_train_baseline(kind, batches=(32, 16), steps=3):
    """Oracle train step (stock PyTorch CPU ops) on this host's cores, per-stage breakdown; the checker timed as baseline.
    The first batch size is the headline leg: 1 warm-up + `steps` timed steps, no time cap short of a step that takes over a
    minute; the other legs are bounded samples (60 s)."""
    from oracle import frontend as ofe, models as om
    n = _cpu_threads()
    allocator = _tune_host_allocator()
    model_name, n_logical, aff = host_cpu()
    res, scan = {}, None
    for batch in batches:
        torch.manual_seed(0)
        if kind == 'deep':
            from pb_sed_amd.modules import DEEP                  # the net_config dictionary only (no device code)
            model = om.FBCRNN.build(num_events=10, hidden_size=512, net=DEEP).train()
        else:
            model = (om.FBCRNN.build() if kind == 'c2' else om.BiCRNN.build(tag_conditioning=True)).train()
        opt = torch.optim.Adam(model.parameters(), lr=5e-4)
        b = synth_batch(batch, 'cpu', kind=kind)

        def step_once(st):
            st = {} if st is None else st
            t0 = time.perf_counter()
            stft = ofe.stft(b['audio_data'])
            seq = np.array(b['seq_len'])
            x, seq_x = model.feature_extractor(stft, seq_len=seq)
            st['front_end_incl_stft'] = time.perf_counter() - t0
            opt.zero_grad()
            t1 = time.perf_counter()
            if kind in ('c2', 'deep'):
                h, seq_h = model.cnn(x, seq_x)
                st['cnn_fwd'] = time.perf_counter() - t1
                t1 = time.perf_counter()
                y_fwd, seq_y = model.fwd_tagging(h, seq_h)
                y_bwd, _ = model.bwd_tagging(h, seq_h)
                st['gru_heads_fwd'] = time.perf_counter() - t1
                out = (y_fwd, y_bwd, seq_y, x, seq_x, (b['weak_targets'], b['boundary_targets']))
            else:
                tag = b['tag_condition'].unsqueeze(-1)
                h, seq_h = model.cnn(x, seq_x, tag)
                st['cnn_fwd'] = time.perf_counter() - t1
                t1 = time.perf_counter()
                h = torch.cat([h, tag.expand(-1, -1, h.shape[-1])], dim=1)
                y, seq_y = model.rnn(h, seq_h)
                st['gru_heads_fwd'] = time.perf_counter() - t1
                out = (torch.sigmoid(y), seq_y, x, seq_x, (b['weak_targets'], b['strong_targets']))
            t1 = time.perf_counter()
            loss = model.review(b, out)['loss']
            loss.backward()
            st['loss_backward'] = time.perf_counter() - t1
            t1 = time.perf_counter()
            torch.nn.utils.clip_grad_norm_(model.parameters(), 1e10)
            opt.step()
            st['clip_adam'] = time.perf_counter() - t1
            return time.perf_counter() - t0

        times, stages = [], []
        for i in range(steps + 1):
            st = {}
            times.append(step_once(st))
            stages.append(st)
            print(f'[bench] cpu_baseline {kind} batch {batch} step {i}: {times[-1]:.2f} s on {n} threads', file=sys.stderr, flush=True)
            headline = batch == batches[0]
            if (not headline and sum(times) > 60.) or (headline and times[-1] > 60. and len(times) >= 2):
                break                                # bounded: side legs by total time; the headline leg only when ONE step takes over a minute
        if kind == 'c2' and batch == batches[-1] and aff >= 64 and sum(times) < 60.:
            # SURVEY.md 8(d) says "all host cores": one step each at 64 / 128 intra-op threads shows why the baseline stops at 32
            scan = {}
            for nt in (64, 128):
                if nt > aff:
                    break
                torch.set_num_threads(nt)
                ts_ = []
                for _ in range(2):                   # the first step after a pool resize pays for the new threads
                    t_s = time.perf_counter()
                    step_once(None)
                    ts_.append(time.perf_counter() - t_s)
                scan[f'batch{batch}_clips_per_s_at_{nt}_threads'] = round(batch / ts_[-1], 3)
                print(f'[bench] cpu_baseline {kind} batch {batch}: {ts_[-1]:.2f} s on {nt} threads', file=sys.stderr, flush=True)
            torch.set_num_threads(n)
        timed = times[1:] if len(times) > 1 else times
        dt = float(np.median(timed))
        med = stages[1 + int(np.argsort(timed)[len(timed) // 2])] if len(times) > 1 else stages[0]
        res[batch] = {'clips_per_s': round(batch / dt, 3), 's_per_step': round(dt, 3), 'timed_steps': len(timed),
                      'spread': round((max(timed) - min(timed)) / dt, 3),          # (slowest - fastest) / median of the timed steps
                      'stage_s': {k: round(v, 3) for k, v in med.items()}}
    main_b = batches[0]
    what = {'c2': 'FBCRNN', 'deep': "FBCRNN 'deep'"}.get(kind, 'tag-cond. BiCRNN')
    out = {'value': res[main_b]['clips_per_s'], 'unit': 'clips/s', 'cores': n, 'kind': 'port',
           'sample': f'oracle {what} train step incl. STFT, batch {main_b}, 1 warm-up + {res[main_b]["timed_steps"]} timed steps, median',
           'cpu_model': model_name, 'host_logical_cpus': n_logical, 'affinity_cpus': aff, 'threads_used': n, 'allocator': allocator,
           'timed_steps': res[main_b]['timed_steps'], 'spread': res[main_b]['spread'],
           'threads_note': 'intra-op threads capped at min(affinity, 32): the oracle step is many small GRU / conv ops whose CPU time stops falling (and then rises) beyond ~32 threads on this class of host; cores = threads used, not the host total',
           f'batch{main_b}': res[main_b]}
    for bb in batches[1:]:
        out[f'batch{bb}'] = res[bb]
    if scan is not None:
        out['threads_scan'] = scan
    if kind == 'c2' and 16 in res:
        out['configs0_stand_in'] = ('BASELINE.json configs[0] (reference plumbing on DESED-weak, batch 16, CPU) is not runnable '
                                    '(padertorch / DESED audio absent): batch16 above is the oracle on synthetic clips')
    return out


def cpu_inference_baseline(clips=8):
    """Oracle ensemble inference (2 FBCRNN + 3 tag-conditioned BiCRNN, median filters, event extraction) on a bounded sample."""
    from oracle import frontend as ofe, models as om, postproc as opp
    n = _cpu_threads()
    allocator = _tune_host_allocator()
    model_name, n_logical, aff = host_cpu()
    torch.manual_seed(0)
    taggers = [om.FBCRNN.build().eval() for _ in range(2)]
    detectors = [om.BiCRNN.build(tag_conditioning=True).eval() for _ in range(3)]
    b = synth_batch(clips, 'cpu')
    medfilt = np.array([[1, 3, 5, 7, 9, 11, 21, 31, 41, 51], [11] * 10, [51] * 10])
    times = []
    for i in range(3):
        t0 = time.perf_counter()
        with torch.no_grad():
            inp = {'stft': ofe.stft(b['audio_data']), 'seq_len': b['seq_len']}
            tags = (np.mean([m.tagging(dict(inp))[0].numpy() for m in taggers], 0)[..., 0] > .5).astype(np.float32)
            inp['tag_condition'] = torch.tensor(tags)
            s = np.mean([m.sound_event_detection(dict(inp))[0].numpy() for m in detectors], 0)
        s = np.stack([np.stack([opp.medfilt(s[:, k], int(nn)) for k, nn in enumerate(row)], 1) for row in medfilt], 1)
        s = s * tags[:, None, :, None]
        n_ev = sum(len(opp.scores_to_event_list(s[j, 0].T, np.round(np.arange(0, 100000) * .02, 6), .5,
                                                [f'c{k}' for k in range(10)])) for j in range(clips))
        times.append(time.perf_counter() - t0)
        print(f'[bench] cpu_baseline c5 pass {i}: {times[-1]:.2f} s on {n} threads ({n_ev} events)', file=sys.stderr, flush=True)
    dt = float(np.median(times[1:]))
    return {'value': round(clips / dt, 3), 'unit': 'clips/s', 'cores': n, 'kind': 'port',
            'sample': f'oracle ensemble inference (2 FBCRNN taggers + 3 tag-conditioned BiCRNN detectors, 3 median-filter '
                      f'variants, event extraction) on {clips} x 10 s clips, 1 warm-up + 2 timed passes, median',
            'cpu_model': model_name, 'host_logical_cpus': n_logical, 'affinity_cpus': aff, 'threads_used': n, 'allocator': allocator}


# ------------------------------------------------------------------------------------------------ per-kernel figures
def _aggregate(events, steps):
    agg = {}
    for name, tag, flops, byts, e0, e1 in events:
        a = agg.setdefault((name, tag), [0.0, 0, flops, byts])
        a[0] += e0.elapsed_time(e1)
        a[1] += 1
    by_family = {}
    for (name, tag), (ms, c, fl, by) in agg.items():
        by_family[name] = by_family.get(name, 0.0) + ms / steps
    return agg, by_family


def _executed(flops, tag):
    """fp32-equivalent flops a launch issues on the MFMA pipe: the Winograd-F(4,3) kernels (fp32 'wino' and bf16x3 'winox3')
    need 18 products per 4 outputs instead of 36."""
    return flops / 2 if tag.endswith(('wino', 'winox3')) else flops


def _x3(name, tag):
    """launches whose products are formed from exact three-way bf16 splits on the bf16 MFMA (6 part products each)"""
    return tag.endswith(('winox3', 'x3pc', 'bf16x3', 'c1x3', 's16x3')) or name in ('pbsed_gru_wgrad_multi', 'pbsed_tm_gemm')


def roofline_objects(agg, by_family, steps, batch, precision, gru_shape, kind):
    out = {}
    conv = {k: v for k, v in agg.items() if k[0].startswith('pbsed_conv') and v[2]}
    (dname, dtag), (tot_ms, cnt, flops, _) = max(conv.items(), key=lambda kv: kv[1][0])
    avg_ms = tot_ms / cnt
    bf16_launch = precision == 'bf16' and ('bf16' in dtag or dname.endswith('_bf16'))
    x3_launch = precision != 'bf16' and _x3(dname, dtag)
    # the ceiling a launch is priced against: the dense MFMA peak of its operand type; bf16x3 launches run six bf16 part
    # products per fp32-equivalent product, so their fp32-equivalent ceiling is the bf16 peak / 6
    peak = PEAK_TFLOPS['bf16'] if bf16_launch else PEAK_TFLOPS['bf16'] / 6 if x3_launch else PEAK_TFLOPS['f32']
    alg = flops / (avg_ms * 1e-3) / 1e12
    exe = _executed(flops, dtag) / (avg_ms * 1e-3) / 1e12
    traffic, source = pmc_traffic(f'{dname} {dtag}')
    wino = dtag.endswith(('wino', 'winox3'))
    out['roofline'] = {
        'bound': 'mfma', 'achieved': round(exe, 2), 'peak': round(peak, 1), 'unit': 'TFLOP/s', 'frac': round(exe / peak, 4),
        'traffic': traffic, 'traffic_unit': 'HBM bytes per launch (PMC)', 'traffic_source': source,
        'kernel': f'{dname} {dtag}', 'avg_ms': round(avg_ms, 4), 'launches': cnt, 'flops_per_launch': flops,
        'flops_executed_per_launch': _executed(flops, dtag),
        'achieved_algorithmic': round(alg, 2), 'frac_algorithmic': round(alg / peak, 4),
        'frac_of_fp32_mfma_peak': round(exe / PEAK_TFLOPS['f32'], 4),
        'operands': 'bf16' if bf16_launch else 'bf16x3 (exact three-way bf16 splits of fp32 operands, 6 bf16 MFMA products per product)' if x3_launch else 'f32',
        'note': ('achieved / frac = fp32-equivalent flops EXECUTED on the MFMA pipe over the ceiling of the operand type (utilisation, <= 1'
                 + ('; bf16x3: the pipe runs 6 bf16 products per counted product, ceiling = 2500 / 6 TFLOP/s' if x3_launch else '')
                 + '); achieved_algorithmic = 2*MACs of the direct convolution over the same time'
                 + ('; this launch is a Winograd-F(4,3) kernel and executes half of the direct products' if wino else ''))}
    # whole step / forward: algorithmic and executed (sum over the bracketed launches)
    exe_step = sum(_executed(fl, tag) * c for (name, tag), (ms, c, fl, by) in agg.items()) / steps
    out['_exe_step_flop'] = exe_step
    if gru_shape is not None:
        nch, nl, t, b, h = gru_shape
        g = {}
        for key, fam in (('forward_scan', 'pbsed_gru_stack_fwd_granule'), ('bptt_scan', 'pbsed_gru_stack_bwd_granule')):
            rows = [v for k, v in agg.items() if k[0] in (fam, fam + '_bf16')]
            if not rows:
                continue
            plain_bf16 = any(k[0] == fam + '_bf16' for k in agg)      # the bf16 training mode: one bf16 product per product
            ms = sum(r[0] for r in rows) / steps            # all scans of one step (FBCRNN: one launch; BiGRU: one per layer)
            fl = sum(r[2] * r[1] for r in rows) / steps
            x3 = h < 512 and not plain_bf16
            tf = fl / (ms * 1e-3) / 1e12
            g[key] = {'ms_per_step': round(ms, 4), 'launches_per_step': round(sum(r[1] for r in rows) / steps, 2),
                      'gflop_per_step': round(fl / 1e9, 2), 'achieved': round(tf, 2),
                      'frac': round(tf / PEAK_TFLOPS['f32'], 4),
                      'operands': ('bf16 (operands of the recurrent / projection products rounded to bf16; fp32 state and accumulation)'
                                   if plain_bf16 else
                                   'bf16x3 (exact 3-way bf16 split of the fp32 operands, 6 bf16 MFMA products per fp32 product)' if x3 else 'f32'),
                      'executed': round(tf * (6 if x3 else 1), 2),
                      'frac_executed': round(tf * (6 if x3 else 1) / PEAK_TFLOPS['bf16'] if (x3 or plain_bf16) else tf / PEAK_TFLOPS['f32'], 4),
                      'us_per_time_step': round(ms * 1e3 / (t * sum(r[1] for r in rows) / steps), 3),
                      'polled_bytes_per_launch': scan_polled_bytes(key, nch, nl, t, b, h),
                      'new_state_bytes_per_launch': int(t * nch * nl * b * h * 4)}
        if g:
            g.update(bound='mfma; latency-bound in practice: T dependent steps with an inter-workgroup hand-off each',
                     note='achieved / frac: fp32-equivalent flops of the recurrence + layer-boundary projections over the fp32-MFMA peak; '
                          'executed / frac_executed: what the MFMA pipe actually ran (bf16x3 scans: 6 bf16 products each, over the bf16 peak)',
                     peak=PEAK_TFLOPS['f32'], peak_executed=PEAK_TFLOPS['bf16'], unit='TFLOP/s',
                     shape=dict(chains=nch, layers=nl, T=t, B=b, H=h))
            pmc = _pmc_mfma_busy(('gru_granule_fwd', 'gru_granule_bwd')) if kind == 'c2' else {}     # the counter pass ran the headline config
            if pmc:
                g['mfma_pipe_busy_pmc'] = pmc
            out['roofline_gru'] = g
            # `roofline` is the DOMINANT launch of the step (largest time per launch among everything bracketed): when that is a
            # scan - it is in every config since round 3 - the conv layer's figures move to `roofline_conv`
            top = max(g[k_]['ms_per_step'] / max(g[k_]['launches_per_step'], 1e-9) for k_ in ('forward_scan', 'bptt_scan') if k_ in g)
            if top > out['roofline']['avg_ms']:
                key = max((k_ for k_ in ('forward_scan', 'bptt_scan') if k_ in g), key=lambda k_: g[k_]['ms_per_step'] / g[k_]['launches_per_step'])
                r = g[key]
                x3 = r['operands'].startswith('bf16x3')
                plain = r['operands'].startswith('bf16 (')
                peak = PEAK_TFLOPS['bf16'] if plain else PEAK_TFLOPS['bf16'] / 6 if x3 else PEAK_TFLOPS['f32']
                out['roofline_conv'] = out['roofline']
                l2_bytes, l2_src = pmc_traffic('gru_granule_fwd' if key == 'forward_scan' else 'gru_granule_bwd') if kind == 'c2' else (None, None)
                out['roofline'] = {
                    'bound': 'mfma', 'achieved': r['achieved'], 'peak': round(peak, 1), 'unit': 'TFLOP/s', 'frac': round(r['achieved'] / peak, 4),
                    'traffic': l2_bytes if l2_bytes is not None else r['polled_bytes_per_launch'],
                    'traffic_kind': ('L2 request bytes per launch (PMC TCC_REQ_sum x 128 B, own pass)' if l2_bytes is not None
                                     else 'bytes requested by the poll loads per launch (analytic, first looks only)'),
                    'traffic_source': l2_src, 'polled_bytes_per_launch': r['polled_bytes_per_launch'],
                    'new_state_bytes_per_launch': r['new_state_bytes_per_launch'], 'kernel': f'persistent GRU scan ({key}): gru_granule_{"fwd" if key == "forward_scan" else "bwd"}_gw_kernel',
                    'avg_ms': round(r['ms_per_step'] / r['launches_per_step'], 4), 'launches_per_step': r['launches_per_step'],
                    'flops_per_launch': r['gflop_per_step'] * 1e9 / r['launches_per_step'], 'us_per_time_step': r['us_per_time_step'],
                    'frac_of_fp32_mfma_peak': r['frac'], 'operands': r['operands'],
                    'note': 'the dominant launch of the step is a persistent scan: T dependent time steps with an inter-workgroup hand-off each - '
                            'latency-bound, not MFMA-bound (tools/gru_scan_prof.py: half of a step is the hand-off).  achieved = flops of the '
                            'recurrence + layer-boundary projections over the launch time; peak = the ceiling of the operand type (bf16x3: '
                            '2500 / 6 TFLOP/s); frac_of_fp32_mfma_peak = the same flops over the fp32-MFMA peak.  traffic: what the scan moves through '
                            'L2 / the fabric per launch (HBM bytes are not what bounds it); polled_bytes_per_launch against '
                            'new_state_bytes_per_launch is the poll amplification.  roofline_conv = the largest conv launch'}
    fe = [v for k, v in agg.items() if k[0] in ('pbsed_logmel_fwd', 'pbsed_logmel_from_stft')]
    if fe:
        fe_ms = fe[0][0] / fe[0][1]
        gbs = 896000.0 * batch / (fe_ms * 1e-3) / 1e9
        out['frontend_hbm'] = {'bound': 'hbm', 'achieved': round(gbs, 1), 'peak': PEAK_HBM_GBS, 'unit': 'GB/s',
                               'frac': round(gbs / PEAK_HBM_GBS, 4), 'avg_ms': round(fe_ms, 4),
                               'bytes_per_launch': int(896000 * batch)}
    return out


EVENT_FILTER = lambda n: n.startswith(('pbsed_conv', 'pbsed_gru_stack', 'pbsed_gru_wgrad', 'pbsed_tm_gemm', 'pbsed_logmel'))


def event_pass(step_fn, steps):
    """Second pass, outside the timed region: HIP events (on the launch stream) around the conv / GRU / front-end calls."""
    from pb_sed_amd import _lib
    _lib.timing, _lib.timing_filter = [], (None if os.environ.get('PBSED_BENCH_TABLE') else EVENT_FILTER)
    for i in range(steps):
        step_fn(i)
    torch.cuda.synchronize()
    events, _lib.timing, _lib.timing_filter = _lib.timing, None, None
    return events


def print_table(agg, steps):
    print('[bench] per-call table (ms/step, calls/step, TFLOP/s algorithmic):', file=sys.stderr)
    for (name, tag), (ms, c, fl, _) in sorted(agg.items(), key=lambda kv: -kv[1][0])[:48]:
        tf = fl * c / (ms * 1e-3) / 1e12 if fl else 0.
        print(f'   {ms / steps:8.3f} {c / steps:5.1f} {tf:7.1f}  {name[6:]} {tag}', file=sys.stderr)


# ------------------------------------------------------------------------------------------------ train configs
def _timed_steps(trainer, batches, steps, world, device):
    """Exactly `steps` train steps between barrier + synchronize pairs; max over ranks.  Returns (seconds, enqueue s, review)."""
    import torch.distributed as dist
    torch.cuda.synchronize()
    if world > 1:
        dist.barrier()
    torch.cuda.synchronize()
    enq, review = 0., None
    t0 = time.perf_counter()
    for i in range(steps):
        review = trainer.step(batches[i % 2])
        enq += trainer.last_enqueue_s
    trainer.finish()                      # the scan error words of the last step (Trainer looks at step n's when step n + 1 returns)
    torch.cuda.synchronize()
    if world > 1:
        dist.barrier()
    torch.cuda.synchronize()
    dt = time.perf_counter() - t0
    if world > 1:
        tt = torch.tensor([dt], device=device, dtype=torch.float64)
        dist.all_reduce(tt, op=dist.ReduceOp.MAX)
        dt = tt.item()
    return dt, enq, review


def bench_train(args, kind, world, rank, device, sustained_steps=0):
    import torch.distributed as dist
    from pb_sed_amd import _lib, ops
    from pb_sed_amd.models import strong_label, weak_label
    from pb_sed_amd.trainer import Trainer
    torch.manual_seed(0)
    precision = args.conv_precision or ('bf16' if kind == 'c3' else 'f32')
    if kind == 'deep':
        from pb_sed_amd.modules import DEEP
        model = weak_label.CRNN.build(num_events=10, hidden_size=512, net=DEEP).to(device)
    else:
        model = (weak_label.CRNN.build() if kind == 'c2' else strong_label.CRNN.build(tag_conditioning=True)).to(device)
    model.conv_precision = precision
    n_params = sum(p.numel() for p in model.parameters())
    trainer = Trainer(model, lr=5e-4, gradient_clipping=1e10)
    trainer.measure_sync = world > 1
    batches = [synth_batch(args.batch, device, seed=2 * rank + j, kind=kind) for j in range(2)]   # weak scaling: 32 clips per GPU

    for i in range(args.warmup):
        t_w = time.perf_counter()
        trainer.step(batches[i % 2])
        torch.cuda.synchronize()
        if rank == 0:
            print(f'[bench] warm-up step {i}: {(time.perf_counter() - t_w) * 1e3:.1f} ms', file=sys.stderr, flush=True)
    calls0, trainer.sync_events = _lib.n_calls, []
    dt, enq, review = _timed_steps(trainer, batches, args.steps, world, device)
    calls = (_lib.n_calls - calls0) / max(args.steps, 1)
    sync_events, trainer.sync_events = trainer.sync_events, None
    loss = float(review['loss'].item())
    sustained = None
    if sustained_steps:
        # the headline's K steps are a fraction of a second; a second, long region shows what the clocks settle at
        chunks, n_chunk = [], max(sustained_steps // 10, 1)
        for _ in range(10):
            dt_c, _, _ = _timed_steps(trainer, batches, n_chunk, world, device)
            chunks.append(args.batch * world * n_chunk / dt_c)
        sustained = {'clips_per_s': round(10 * n_chunk * args.batch * world / sum(args.batch * world * n_chunk / c for c in chunks), 2),
                     'steps': 10 * n_chunk, 'clips_per_s_median_of_10_chunks': round(float(np.median(chunks)), 2),
                     'clips_per_s_min_max_of_chunks': [round(min(chunks), 2), round(max(chunks), 2)],
                     'note': f'10 back-to-back timed regions of {n_chunk} steps each after the headline region, same bracket '
                             '(barrier + synchronize, max over ranks); clips_per_s = all clips / total time'}
    exposed_ms = (sum(a.elapsed_time(b) for a, b in sync_events) / max(args.steps, 1)) if sync_events else None
    ev_steps = min(max(args.steps, 1), 10)
    events = event_pass(lambda i: trainer.step(batches[i % 2]), ev_steps)
    # host -> device hand-over of one batch (pinned, PCIe), outside the timed region: `value` is HBM-resident
    host_batch = {k: v.cpu().pin_memory() for k, v in batches[0].items() if isinstance(v, torch.Tensor)}
    torch.cuda.synchronize()
    t_h = time.perf_counter()
    for _ in range(5):
        for k, v in host_batch.items():
            batches[0][k].copy_(v, non_blocking=True)
    torch.cuda.synchronize()
    h2d_ms = (time.perf_counter() - t_h) / 5 * 1e3
    # ... and measured end to end: the same train steps on batches that arrive from pinned host memory through
    # data.DevicePrefetcher (batch n + 1 copied on a side stream while step n runs)
    from pb_sed_amd.data import DevicePrefetcher
    host_batches = [{k: (v.cpu().pin_memory() if isinstance(v, torch.Tensor) else v) for k, v in b.items()} for b in batches]
    n_host = min(max(args.steps, 1), 20)
    for b in DevicePrefetcher([host_batches[i % 2] for i in range(3)], device):
        trainer.step(b)
    torch.cuda.synchronize()
    t_h = time.perf_counter()
    for b in DevicePrefetcher((host_batches[i % 2] for i in range(n_host)), device):
        trainer.step(b)
    torch.cuda.synchronize()
    host_step_ms = (time.perf_counter() - t_h) / n_host * 1e3
    ops.check_gru_sync()
    if rank != 0:
        return None
    agg, by_family = _aggregate(events, ev_steps)
    ms_step = dt / args.steps * 1e3
    clips = args.batch * world
    what = {'c2': 'FBCRNN weak_label_crnn.training batch 32/GPU fp32, 10 s 16 kHz clips (BASELINE.json configs[1])',
            'c3': 'tag-conditioned strong_label BiCRNN training batch 32/GPU bf16, 10 s 16 kHz clips (BASELINE.json configs[2])',
            'deep': "FBCRNN weak_label_crnn.training with net_config 'deep' (width 2: 18 conv2d layers up to 512 channels with "
                    'residual connections, 8 conv1d layers, GRU 2 x 512; training.py:170-183), batch 32/GPU fp32, 10 s 16 kHz clips'}[kind]
    gru_shape = {'c2': (2, 2, 500, args.batch, 256), 'c3': (2, 1, 500, args.batch, 256), 'deep': (2, 2, 500, args.batch, 512)}[kind]
    out = {
        'metric': {'c2': '10s@16kHz clips/sec (train step) FBCRNN batch32',
                   'c3': '10s@16kHz clips/sec (train step) tag-conditioned BiCRNN batch32 bf16',
                   'deep': "10s@16kHz clips/sec (train step) FBCRNN net_config 'deep' width 2 batch32"}[kind],
        'value': round(clips / (dt / args.steps), 2), 'unit': 'clips/s', 'n_gpus': world,
        'steps': args.steps, 'warmup': args.warmup, 'ms_per_step': round(ms_step, 3),
        'higher_is_better': True, 'scaling': 'weak', 'vs_baseline': None,
        'dtype': {'f32': 'f32', 'bf16': 'bf16 (MFMA operands of conv / projection / head launches; fp32 accumulate, BN, GRU state, '
                                        'master weights)',
                  'bf16x3': 'bf16x3 (fp32 operands split into 3 bf16 terms, fp32 accumulate)'}[precision],
        'arithmetic': ARITHMETIC_NOTE[precision],
        'data': 'synthetic (randn waveforms, random-init weights, 2 distinct resident batches in rotation)',
        'config': {'workload': what + '; full train step incl. fused log-mel front-end, loss, backward, grad-norm clip, Adam'
                                      + (', RCCL grad all-reduce' if world > 1 else ''),
                   'global_batch': clips, 'n_params': n_params, 'parallelism': f'dp{world}'},
    }
    rf = roofline_objects(agg, by_family, ev_steps, args.batch, precision, gru_shape, kind)
    exe_step = rf.pop('_exe_step_flop')
    out.update(rf)
    is_fwd = lambda name, tag: (name.startswith(('pbsed_conv_fwd', 'pbsed_gru_stack_fwd', 'pbsed_logmel'))
                                or (name == 'pbsed_tm_gemm' and tag.endswith(' fwd')))      # the GRU input projections
    fwd_ms = sum(ms for (name, tag), (ms, c, fl, by) in agg.items() if is_fwd(name, tag)) / ev_steps
    fwd_exe = sum(_executed(fl, tag) * c for (name, tag), (ms, c, fl, by) in agg.items()
                  if is_fwd(name, tag) and not name.startswith('pbsed_logmel')) / ev_steps
    fwd_alg = sum(fl * c for (name, tag), (ms, c, fl, by) in agg.items()
                  if is_fwd(name, tag) and not name.startswith('pbsed_logmel')) / ev_steps
    # BASELINE.md section 2 for the two named configs; any other net: 2*MACs of the bracketed forward launches
    fwd_tflop = FWD_GFLOP[kind] * args.batch / 1e3 if kind in FWD_GFLOP else fwd_alg / 1e12
    train_tflop = 3 * fwd_tflop
    peak = PEAK_TFLOPS['f32']
    out['step_mfma'] = {'algorithmic_tflop_per_step': round(train_tflop, 4),
                        'algorithmic_tflops_per_gpu': round(train_tflop / (ms_step * 1e-3), 2),
                        'executed_tflop_per_step_in_bracketed_launches': round(exe_step / 1e12, 4),
                        'executed_tflops_per_gpu': round(exe_step / 1e12 / (ms_step * 1e-3), 2),
                        'frac_executed_of_fp32_mfma_peak': round(exe_step / 1e12 / (ms_step * 1e-3) / peak, 4),
                        'frac_algorithmic_of_fp32_mfma_peak': round(train_tflop / (ms_step * 1e-3) / peak, 4),
                        'note': 'executed = fp32-equivalent products issued by the bracketed launches: a Winograd launch counts half of '
                                'the direct products; bf16x3 launches count each fp32-equivalent product once although the bf16 pipe '
                                'runs six part products for it'}
    out['forward_conv_gru'] = {'ms_per_step': round(fwd_ms, 3), 'algorithmic_tflop': round(fwd_tflop, 4),
                               'algorithmic_tflops': round(fwd_tflop / (fwd_ms * 1e-3), 2),
                               'executed_tflops': round(fwd_exe / 1e12 / (fwd_ms * 1e-3), 2),
                               'frac_executed_of_fp32_mfma_peak': round(fwd_exe / 1e12 / (fwd_ms * 1e-3) / peak, 4),
                               'frac_algorithmic_of_fp32_mfma_peak': round(fwd_tflop / (fwd_ms * 1e-3) / peak, 4)}
    if precision == 'bf16':
        # a bf16 line is priced against the bf16 peak first; the fp32-peak fractions above are kept for comparison with c2 only
        pb = PEAK_TFLOPS['bf16']
        out['step_mfma']['frac_algorithmic_of_bf16_mfma_peak'] = round(train_tflop / (ms_step * 1e-3) / pb, 4)
        out['step_mfma']['frac_executed_of_bf16_mfma_peak'] = round(exe_step / 1e12 / (ms_step * 1e-3) / pb, 4)
        out['forward_conv_gru']['frac_algorithmic_of_bf16_mfma_peak'] = round(fwd_tflop / (fwd_ms * 1e-3) / pb, 4)
        out['forward_conv_gru']['frac_executed_of_bf16_mfma_peak'] = round(fwd_exe / 1e12 / (fwd_ms * 1e-3) / pb, 4)
        out['step_mfma']['honest_reading'] = ('dtype is bf16: the step runs at the frac_*_of_bf16_mfma_peak fractions of what the chip '
                                              'can do with bf16 operands; it is latency / HBM / fp32-kernel bound, see ms_per_step_by_entry_point')
    if sustained is not None:
        out['sustained'] = sustained
    out['ms_per_step_by_entry_point'] = {k: round(v, 3) for k, v in sorted(by_family.items(), key=lambda kv: -kv[1])}
    out['host'] = {'enqueue_ms_per_step': round(enq / args.steps * 1e3, 3), 'c_abi_calls_per_step': round(calls, 1),
                   'note': 'host time of a step up to (not including) the wait for its deferred review summary'}
    out['h2d'] = {'ms_per_batch': round(h2d_ms, 3), 'bytes': int(sum(v.numel() * v.element_size() for v in host_batch.values())),
                  'clips_per_s_if_serialised': round(clips / (dt / args.steps + h2d_ms * 1e-3), 2),
                  'ms_per_step_batches_from_pinned_host': round(host_step_ms, 3),
                  'clips_per_s_batches_from_pinned_host': round(clips / (host_step_ms * 1e-3), 2),
                  'note': 'PCIe-inclusive, measured: every batch copied from pinned host memory by data.DevicePrefetcher (side stream, one batch ahead); never the headline value'}
    out['loss'] = loss
    if world > 1:
        s_bytes = 4 * n_params
        out['allreduce'] = {
            'implementation': trainer.allreduce, 'bytes_per_step': s_bytes, 'buckets': trainer.bucket_names,
            'exposed_ms_per_step': None if exposed_ms is None else round(exposed_ms, 4),
            'bus_bytes_per_step': round(s_bytes * 2 * (world - 1) / world),
            'busbw_GBs_if_fully_exposed': None if not exposed_ms else round(s_bytes * 2 * (world - 1) / world / (exposed_ms * 1e-3) / 1e9, 1),
            'time_at_ring_bound_ms': round(s_bytes * 2 * (world - 1) / world / (XGMI_LINK_GBS * 1e9) * 1e3, 4),
            'time_at_direct_bound_ms': round(s_bytes * 2 * (world - 1) / world / (XGMI_LINKS * XGMI_LINK_GBS * 1e9) * 1e3, 4),
            'note': 'exposed = time the compute stream waits in GradSync.finish() (collectives are issued per bucket from inside '
                    'backward and overlap the remaining conv launches); bounds: 153 GB/s per xGMI link (ring) and 7 links (direct)'}
    print('[bench] gpu: ' + json.dumps({k: out[k] for k in ('value', 'ms_per_step', 'roofline')}), file=sys.stderr, flush=True)
    if os.environ.get('PBSED_BENCH_TABLE'):
        print_table(agg, ev_steps)
    if world == 1 and not args.no_cpu_baseline:
        out['cpu_baseline'] = cpu_train_baseline(kind, batches={'c2': (32, 16), 'deep': (8,)}.get(kind, (32,)))
    return out


# ------------------------------------------------------------------------------------------------ ensemble inference
def bench_inference(args, world, rank, device):
    import torch.distributed as dist
    from pb_sed_amd import inference as inf, ops
    from pb_sed_amd.models import strong_label, weak_label
    torch.manual_seed(0)
    precision = args.conv_precision or 'f32'
    taggers = [weak_label.CRNN.build() for _ in range(2)]
    detectors = [strong_label.CRNN.build(tag_conditioning=True) for _ in range(3)]
    for m in taggers + detectors:
        m.conv_precision = precision
    classes = [f'class{i}' for i in range(10)]
    total = args.batch if args.batch != 32 else 64                 # configs[4]: batch 64 (the default --batch is the train size)
    assert total % world == 0, (total, world)
    per = total // world
    data_dev = []
    for j in range(2):                                              # two distinct resident batches, this rank's shard of each
        b = synth_batch(total, 'cpu', seed=10 + j)
        ids = [f'clip{j}_{i}' for i in range(total)]
        sl = slice(rank * per, (rank + 1) * per)
        data_dev.append({'audio_data': b['audio_data'][sl].to(device), 'seq_len': b['seq_len'][sl], 'example_id': ids[sl]})
    medfilt = np.array([[1, 3, 5, 7, 9, 11, 21, 31, 41, 51], [11] * 10, [51] * 10])
    ts = np.round(np.arange(0, 100000) * .02, 6)

    def job(first, count, data=None):
        """`count` batches through the reference's two dataset passes (experiments/strong_label_crnn/inference.py: tags of the
        whole dataset first, then tag-conditioned detection), then the event lists.  inference() keeps one batch in flight."""
        data = data_dev if data is None else data
        batches = [dict(data[(first + i) % 2], example_id=[f'{a}_{first + i}' for a in data[(first + i) % 2]['example_id']])
                   for i in range(count)]
        tag_scores = inf.tagging(taggers, batches, device)
        tags = {a: (s[0] > .5).astype(np.float32) for a, s in tag_scores.items()}

        def conditioned():
            for b in batches:
                cond = ops.host_to_device(np.stack([tags[a] for a in b['example_id']]), device)     # pinned staging: no host stall
                yield dict(b, tag_condition=cond)
        sed = inf.sound_event_detection(detectors, conditioned(), device, medfilt_length=medfilt, apply_mask=True, masks=tags)
        events = inf.scores_to_event_list({a: s[0] for a, s in sed.items()}, .5, classes, ts, device=device)
        assert len(events) == count * per, (len(events), count, per)
        return len(events)

    def run(i):
        return job(i, 1)

    if args.warmup:
        job(0, args.warmup)
    torch.cuda.synchronize()
    if world > 1:
        dist.barrier()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    n = job(0, args.steps)
    torch.cuda.synchronize()
    if world > 1:
        dist.barrier()
    torch.cuda.synchronize()
    dt = time.perf_counter() - t0
    if world > 1:
        tt = torch.tensor([dt], device=device, dtype=torch.float64)
        dist.all_reduce(tt, op=dist.ReduceOp.MAX)
        dt = tt.item()
    t1 = time.perf_counter()
    for i in range(min(args.steps, 10)):
        run(i)
    torch.cuda.synchronize()
    serial_ms = (time.perf_counter() - t1) / min(args.steps, 10) * 1e3        # one batch at a time (round 3's timed loop)
    # the same job with the waveforms handed over from PINNED HOST memory batch by batch (both passes copy them over PCIe,
    # asynchronously, in stream order): the rate a caller sees whose clips are not resident - `value` is HBM-resident
    data_host = [dict(d, audio_data=d['audio_data'].cpu().pin_memory()) for d in data_dev]
    n_host = min(args.steps, 20)
    job(0, 2, data_host)
    torch.cuda.synchronize()
    t1 = time.perf_counter()
    job(0, n_host, data_host)
    torch.cuda.synchronize()
    host_ms = (time.perf_counter() - t1) / n_host * 1e3
    ev_steps = min(max(args.steps, 1), 5)
    events = event_pass(run, ev_steps)
    ops.check_gru_sync()
    if rank != 0:
        return None
    agg, by_family = _aggregate(events, ev_steps)
    ms = dt / args.steps * 1e3
    out = {
        'metric': '10s@16kHz clips/sec, 5-model FBCRNN+BiCRNN ensemble inference, batch 64',
        'value': round(total / (dt / args.steps), 2), 'unit': 'clips/s', 'n_gpus': world, 'steps': args.steps,
        'warmup': args.warmup, 'ms_per_step': round(ms, 3), 'higher_is_better': True, 'scaling': 'strong',
        'vs_baseline': None, 'dtype': precision,
        'arithmetic': ARITHMETIC_NOTE[precision],
        'data': 'synthetic (randn waveforms, random-init weights, 2 distinct resident batches in rotation)',
        'config': {'workload': 'strong_label_crnn_inference: 2 FBCRNN taggers -> tags -> 3 tag-conditioned BiCRNN detectors, '
                               'ensemble mean, 3 per-class median-filter variants, tag masking, event lists on the host '
                               f'(BASELINE.json configs[4]); batch {total} sharded over {world} rank(s), no collective; the timed '
                               f'job is {args.steps} batches through the tagging pass, then the detection pass (one batch in '
                               'flight inside inference()), then the event lists',
                   'global_batch': total, 'clips_per_rank': per, 'models': 5, 'parallelism': f'clips/{world}'},
    }
    rf = roofline_objects(agg, by_family, ev_steps, per, precision, None, 'c5')
    rf.pop('_exe_step_flop')
    out.update(rf)
    fwd_tflop = (2 * FWD_GFLOP['c2'] + 3 * FWD_GFLOP['c3']) * per / 1e3
    out['step_mfma'] = {'algorithmic_tflop_per_step_per_gpu': round(fwd_tflop, 4),
                        'algorithmic_tflops_per_gpu': round(fwd_tflop / (ms * 1e-3), 2),
                        'frac_algorithmic_of_fp32_mfma_peak': round(fwd_tflop / (ms * 1e-3) / PEAK_TFLOPS['f32'], 4)}
    out['ms_per_step_by_entry_point'] = {k: round(v, 3) for k, v in sorted(by_family.items(), key=lambda kv: -kv[1])}
    out['ms_per_step_one_batch_at_a_time'] = round(serial_ms, 3)
    out['h2d'] = {'ms_per_step_waveforms_from_pinned_host': round(host_ms, 3),
                  'clips_per_s_waveforms_from_pinned_host': round(total / (host_ms * 1e-3), 2),
                  'bytes_per_step': int(2 * data_dev[0]['audio_data'].numel() * 4),
                  'note': 'PCIe-inclusive: both dataset passes copy the batch from pinned host memory; never the headline value'}
    if os.environ.get('PBSED_BENCH_TABLE'):
        print_table(agg, ev_steps)
    if world == 1 and not args.no_cpu_baseline:
        out['cpu_baseline'] = cpu_inference_baseline()
    return out


def _free_port():
    import socket
    with socket.socket() as sk:
        sk.bind(('127.0.0.1', 0))
        return sk.getsockname()[1]


def self_launch(n):
    """`python bench.py --gpus N` without a launcher: re-run this command as N ranks (one per GPU) under
    torch.distributed.run on 127.0.0.1 and hand its exit code back.  stdout of the ranks passes through, so rank 0's JSON
    line is this process' JSON line."""
    import subprocess
    cmd = [sys.executable, '-m', 'torch.distributed.run', '--nnodes=1', f'--nproc-per-node={n}', '--master-addr', '127.0.0.1',
           '--master-port', str(_free_port()), os.path.abspath(__file__)] + sys.argv[1:]
    print(f'[bench] --gpus {n} without WORLD_SIZE: launching {n} ranks: {" ".join(cmd)}', file=sys.stderr, flush=True)
    env = dict(os.environ, HSA_ENABLE_IPC_MODE_LEGACY=os.environ.get('HSA_ENABLE_IPC_MODE_LEGACY', '0'))
    return subprocess.run(cmd, env=env).returncode


def rendezvous_check(world, rank, device, backend):
    """Every rank contributes rank + 1 to a sum-all-reduce: the collective backend really spans `world` ranks."""
    import torch.distributed as dist
    t = torch.tensor([float(rank + 1)], device=device if backend == 'nccl' else 'cpu')
    dist.all_reduce(t)
    got, want = t.item(), world * (world + 1) / 2
    assert dist.get_world_size() == world and got == want, (dist.get_world_size(), world, got, want)
    return {'backend': 'nccl (RCCL)' if backend == 'nccl' else backend, 'ranks_seen': dist.get_world_size(),
            'allreduce_check': got}


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument('--gpus', type=int, default=1)
    ap.add_argument('--steps', type=int, default=50)
    ap.add_argument('--warmup', type=int, default=10)
    ap.add_argument('--batch', type=int, default=32, help='clips per GPU (c5: total clips, default 64)')
    ap.add_argument('--config', default='c2', choices=['c2', 'c3', 'c5', 'deep'])
    ap.add_argument('--no-cpu-baseline', action='store_true')
    ap.add_argument('--headline-only', action='store_true',
                    help='default run: skip the sustained figure and the other_configs (c3, c5, deep) legs')
    ap.add_argument('--sustained-steps', type=int, default=500)
    ap.add_argument('--conv-precision', default=None, choices=['f32', 'bf16', 'bf16x3'],
                    help="operand format of the conv MFMAs; default: f32 (c2, c5, deep), bf16 (c3)")
    ap.add_argument('--rendezvous-only', default=None, choices=['gloo', 'nccl'],
                    help='launch / rendezvous / all-reduce check only, no GPU work (tests/test_dp_gloo.py drives it with gloo on CPU)')
    args = ap.parse_args()

    if 'WORLD_SIZE' not in os.environ and args.gpus > 1:
        sys.exit(self_launch(args.gpus))

    import torch.distributed as dist
    world = int(os.environ.get('WORLD_SIZE', '1'))
    rank = int(os.environ.get('RANK', '0'))
    local_rank = int(os.environ.get('LOCAL_RANK', '0'))
    assert world == args.gpus, f'--gpus {args.gpus} but WORLD_SIZE={world}'
    if args.rendezvous_only:
        dist.init_process_group(args.rendezvous_only)
        info = rendezvous_check(world, rank, f'cuda:{local_rank}', args.rendezvous_only)
        if rank == 0:
            print(json.dumps({'rendezvous': info, 'n_gpus': world}))
        dist.destroy_process_group()
        return
    torch.cuda.set_device(local_rank)
    device = f'cuda:{local_rank}'
    rendezvous = None
    if world > 1:
        dist.init_process_group('nccl', device_id=torch.device(device))
        rendezvous = rendezvous_check(world, rank, device, 'nccl')
        if rank == 0:
            print(f'[bench] RCCL spans {world} ranks: {rendezvous}', file=sys.stderr, flush=True)
    full = args.config == 'c2' and not args.headline_only and args.conv_precision is None and args.batch == 32
    out = run_config(args, args.config, world, rank, device, sustained_steps=args.sustained_steps if full else 0)
    if full and world == 1:
        # the driver runs the default command only: carry the sustained figure and the other BASELINE configs on the same line
        # (N = 1 only: the scaling runs at N > 1 time the headline config and its sustained figure, nothing a rank could trip over)
        others = {}
        for kind in ('c3', 'c5', 'deep'):
            sub = argparse.Namespace(**vars(args))
            sub.steps, sub.warmup, sub.no_cpu_baseline = (10, 3, True) if kind == 'c5' else (20, 5, True)
            try:
                res = run_config(sub, kind, world, rank, device)
            except Exception as e:                       # a leg that fails must not take the headline with it
                res = {'error': f'{type(e).__name__}: {e}'}
            if rank == 0:
                others[kind] = res
        if rank == 0:
            out['other_configs'] = others
    if rank == 0:
        if rendezvous is not None:
            out['rendezvous'] = rendezvous
        emit(out)
    if world > 1:
        dist.destroy_process_group()


# ------------------------------------------------------------------------------------------------ the contract line
LINE_LIMIT = 6000           # bytes; tests/test_host_logic.py::test_bench_contract_line_is_compact holds the line to it


def _clean(x):
    """strict JSON: no NaN / Infinity (json.loads of a strict parser refuses them)"""
    if isinstance(x, float):
        return x if np.isfinite(x) else None
    if isinstance(x, dict):
        return {k: _clean(v) for k, v in x.items()}
    if isinstance(x, (list, tuple)):
        return [_clean(v) for v in x]
    return x


def _pick(d, keys):
    return {k: d[k] for k in keys if d is not None and k in d}


ROOFLINE_KEYS = ('bound', 'achieved', 'peak', 'unit', 'frac', 'traffic', 'traffic_kind', 'polled_bytes_per_launch', 'new_state_bytes_per_launch',
                 'kernel', 'avg_ms', 'flops_per_launch', 'operands', 'us_per_time_step')


def _short_roofline(r):
    r = _pick(r, ROOFLINE_KEYS)
    if 'operands' in r:
        r['operands'] = r['operands'].split(' ')[0]           # 'bf16x3', 'bf16', 'f32'
    if 'kernel' in r:
        r['kernel'] = r['kernel'][:96]
    return r


def contract_line(out):
    """The ONE stdout line the driver parses: the contract keys + compact roofline / cpu_baseline / other_configs objects.
    Everything else of `out` (notes, per-entry-point tables, sub-config details) goes to stderr and gpurun_out/bench_detail.json."""
    line = _pick(out, ('metric', 'value', 'unit', 'n_gpus', 'steps', 'warmup', 'ms_per_step', 'higher_is_better', 'scaling',
                       'vs_baseline', 'data'))
    line['dtype'] = str(out.get('dtype', '')).split(' ')[0]
    cfg = out.get('config', {})
    line['config'] = _pick(cfg, ('global_batch', 'n_params', 'parallelism', 'clips_per_rank', 'models'))
    line['config']['workload'] = str(cfg.get('workload', ''))[:160]
    line['roofline'] = _short_roofline(out.get('roofline'))
    for key in ('roofline_conv', 'frontend_hbm'):
        if key in out:
            line[key] = _short_roofline(out[key])
    if 'roofline_gru' in out:
        line['roofline_gru'] = {k: _pick(v, ('ms_per_step', 'achieved', 'frac', 'us_per_time_step', 'polled_bytes_per_launch'))
                                for k, v in out['roofline_gru'].items() if k in ('forward_scan', 'bptt_scan')}
    if 'cpu_baseline' in out:
        cb = out['cpu_baseline']
        c = _pick(cb, ('value', 'unit', 'cores', 'kind', 'cpu_model', 'host_logical_cpus', 'timed_steps', 'spread'))
        c['sample'] = str(cb.get('sample', ''))[:100]
        for k, v in cb.items():
            if k.startswith('batch') and isinstance(v, dict) and 'clips_per_s' in v:
                c[k + '_clips_per_s'] = v['clips_per_s']
        if 'threads_scan' in cb:
            c['threads_scan'] = cb['threads_scan']
        line['cpu_baseline'] = c
    if 'sustained' in out:
        line['sustained'] = _pick(out['sustained'], ('clips_per_s', 'steps'))
    if 'step_mfma' in out:
        line['step_mfma'] = _pick(out['step_mfma'], ('algorithmic_tflops_per_gpu', 'frac_algorithmic_of_fp32_mfma_peak',
                                                     'frac_algorithmic_of_bf16_mfma_peak'))
    if 'loss' in out:
        line['loss'] = out['loss']
    if 'other_configs' in out:
        oc = {}
        for k, v in out['other_configs'].items():
            if 'error' in v:
                oc[k] = {'error': str(v['error'])[:120]}
                continue
            rf = v.get('roofline', {})
            oc[k] = {'value': v.get('value'), 'ms_per_step': v.get('ms_per_step'), 'dtype': str(v.get('dtype', '')).split(' ')[0],
                     'roofline_frac': rf.get('frac'), 'kernel': str(rf.get('kernel', ''))[:64]}
        line['other_configs'] = oc
    if 'rendezvous' in out:
        line['rendezvous'] = out['rendezvous']
    if 'allreduce' in out:
        line['allreduce'] = _pick(out['allreduce'], ('implementation', 'bytes_per_step', 'exposed_ms_per_step',
                                                     'busbw_GBs_if_fully_exposed', 'time_at_ring_bound_ms'))
    line['detail'] = 'gpurun_out/bench_detail.json'
    return _clean(line)


def emit(out):
    """Full result -> stderr + gpurun_out/bench_detail.json; the compact contract line -> stdout (one line, < LINE_LIMIT bytes)."""
    detail = json.dumps(_clean(out), allow_nan=False)
    try:
        os.makedirs(os.path.join(ROOT, 'gpurun_out'), exist_ok=True)
        with open(os.path.join(ROOT, 'gpurun_out', 'bench_detail.json'), 'w') as f:
            f.write(detail + '\n')
    except OSError as e:
        print(f'[bench] bench_detail.json not written: {e}', file=sys.stderr)
    print('[bench] detail: ' + detail, file=sys.stderr, flush=True)
    print(fit_line(contract_line(out)), flush=True)


# optional objects of the contract line, least important first: what an oversized line loses before anything else does.  The
# contract keys themselves, `roofline` and `cpu_baseline` are never dropped.
DROP_ORDER = ('rendezvous', 'loss', 'sustained', 'step_mfma', 'other_configs', 'roofline_gru', 'frontend_hbm', 'roofline_conv', 'allreduce')


def fit_line(line):
    """The contract line as ONE text line under LINE_LIMIT bytes whatever the run put into it: optional objects go first
    (DROP_ORDER, named in `dropped`), then unbounded strings inside what is left are cut.  A result always reaches stdout -
    an oversized line used to be an AssertionError AFTER the whole benchmark had run (nothing for the driver to parse)."""
    dumps = lambda d: json.dumps(d, allow_nan=False, separators=(',', ':'))
    line = dict(line)
    text = dumps(line)
    for key in DROP_ORDER:
        if len(text) < LINE_LIMIT:
            break
        if key in line:
            del line[key]
            line.setdefault('dropped', []).append(key)
            text = dumps(line)
    if len(text) >= LINE_LIMIT:                              # still too long: strings of the kept objects (sample, workload, kernel ...)
        def cut(x, n):
            if isinstance(x, str):
                return x[:n]
            if isinstance(x, dict):
                return {k: cut(v, n) for k, v in list(x.items())[:24]}
            if isinstance(x, (list, tuple)):
                return [cut(v, n) for v in x[:16]]
            return x
        for n in (64, 24, 8):
            text = dumps(cut(line, n))
            if len(text) < LINE_LIMIT:
                break
    return text.replace('\n', ' ')


def run_config(args, kind, world, rank, device, sustained_steps=0):
    if kind == 'c5':
        return bench_inference(args, world, rank, device)
    return bench_train(args, kind, world, rank, device, sustained_steps=sustained_steps)


if __name__ == '__main__':
    main()
