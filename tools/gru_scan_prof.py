"""Time the persistent GRU scans at a BASELINE shape and print the per-step timeline of one workgroup.

    python tools/gru_scan_prof.py [--shape c2|c3|deep] [--batch 32] [--block N] [--no-prof] [--delays f,fg,b,bg] [--reps 20]

c2: 2 chains x 2 layers, H 256, T 500 (FBCRNN stacks);  c3: 2 chains x 1 layer (one BiGRU layer);  deep: 2 x 2, H 512.
Durations: HIP events around `reps` launches (after the in-place delay tuning of ops.gru_stack_*).  Timeline: shader-clock
stamps of workgroup --block (pbsed_gru_set_prof), steps 200..231, averaged; clocks, and microseconds at the measured
clocks-per-step / wall-time-per-step ratio.  The library reads its PBSED_* switches once per process: run variants as
separate processes (tools/gru_scan_variants.sh).
"""
import argparse
import ctypes as C
import os
import sys

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from pb_sed_amd import _lib, ops  # noqa: E402

ap = argparse.ArgumentParser()
ap.add_argument('--shape', default='c2')
ap.add_argument('--batch', type=int, default=32)
ap.add_argument('--block', type=int, default=0)
ap.add_argument('--no-prof', action='store_true')
ap.add_argument('--delays', default=None)
ap.add_argument('--reps', type=int, default=20)
ap.add_argument('--precision', default='f32')
args = ap.parse_args()

dev = 'cuda:0'
torch.manual_seed(0)
nch, nl, h = {'c2': (2, 2, 256), 'c3': (2, 1, 256), 'deep': (2, 2, 512)}[args.shape]
t, b = 500, args.batch
seq = torch.full((b,), t, dtype=torch.int32, device=dev)
k = 1. / np.sqrt(h)
u = lambda *s: (torch.rand(*s, device=dev) * 2 - 1) * k
gi0 = [torch.randn(t, b, 3 * h, device=dev) * .5 for _ in range(nch)]
idx = [(c, l) for c in range(nch) for l in range(nl)]
w_ih = [u(3 * h, h) if l else None for c, l in idx]
b_ih = [u(3 * h) if l else None for c, l in idx]
w_hh = [u(3 * h, h) for _ in idx]
b_hh = [u(3 * h) for _ in idx]
reverse = [bool(c & 1) for c in range(nch)]
if args.delays:
    os.environ['PBSED_GRU_POLL_DELAYS'] = args.delays


def fwd():
    return ops.gru_stack_fwd(gi0, w_ih, b_ih, w_hh, b_hh, reverse, seq, nl, save=True, precision=args.precision)


hs, save = fwd()                                   # tunes the first-poll delay of this shape in place
w_hh_t = [ops.transpose2d(w) for w in w_hh]
w_ih_up_t = [ops.transpose2d(w_ih[i + 1]) if l + 1 < nl else None for i, (c, l) in enumerate(idx)]
dy = [torch.randn(t, b, h, device=dev) * .01 for _ in range(nch)]


def bwd():
    return ops.gru_stack_bwd(w_hh_t, w_ih_up_t, hs, save, dy, reverse, seq, nl, precision=args.precision)


bwd()
ops.check_gru_sync()


def time_it(fn):
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(args.reps):
        fn()
    e1.record()
    e1.synchronize()
    return e0.elapsed_time(e1) / args.reps


ms_f, ms_b = time_it(fwd), time_it(bwd)
ops.check_gru_sync()
tuned = {k_[1:3]: v for k_, v in ops._POLL_TUNED.items()}
print(f'{args.shape} B{b} H{h} {nch}x{nl} {args.precision}: forward {ms_f:.3f} ms ({ms_f * 1e3 / t:.2f} us/step)  '
      f'BPTT {ms_b:.3f} ms ({ms_b * 1e3 / t:.2f} us/step)   tuned first-poll delays {tuned}  '
      f'env {dict((k_, v) for k_, v in os.environ.items() if k_.startswith("PBSED_"))}')
if args.no_prof:
    sys.exit(0)

NAMES_C = ['top', 'poll issued', 'poll satisfied', 'partials written', 'past barrier']
NAMES_G = ['top', 'past barrier', 'reduced', 'published', 'stored']
for name, fn, ms in (('forward', fwd, ms_f), ('BPTT', bwd, ms_b)):
    for block in sorted({args.block, args.block + 8 * 16}):      # a ring block and (two-layer stacks) a later slot of the same XCD
        buf = torch.zeros(32 * 16, dtype=torch.int64, device=dev)
        _lib.call('pbsed_gru_set_prof', buf.data_ptr(), block)
        fn()
        torch.cuda.synchronize()
        _lib.call('pbsed_gru_set_prof', None, 0)
        p = buf.cpu().numpy().reshape(32, 16).astype(np.int64)
        if not p[:, 8].any():
            print(f'  {name} block {block}: no stamps (idle slot)')
            continue
        step_clk = np.diff(p[:, 8]).mean()
        us = ms * 1e3 / t / step_clk                 # microseconds per shader clock at this launch's pace
        print(f'  {name} block {block}: {step_clk:.0f} clocks per step = {ms * 1e3 / t:.2f} us  ({1 / us / 1e3:.2f} GHz)')
        if p[:, 0].any():
            c = p[1:-1, :5] - p[1:-1, 8:9]           # relative to the gate wave's step top
            line = '   '.join(f'{n} {c[:, i].mean():+.0f}' for i, n in enumerate(NAMES_C))
            print(f'    contraction wave 0 (clocks after the gate wave\'s step top): {line};  missed polls per step {p[1:-1, 5].mean():.2f}')
            d = np.diff(p[1:-1, :5], axis=1).mean(0)
            print(f'      sleep {d[0]:.0f}  poll {d[1]:.0f}  split + MFMA + LDS write {d[2]:.0f}  barrier wait {d[3]:.0f}')
        g = p[1:-1, 8:13]
        d = np.diff(g, axis=1).mean(0)
        print(f'    gate wave 0: wait at barrier {d[0]:.0f}  reduction {d[1]:.0f}  gate maths + publish {d[2]:.0f}  stores {d[3]:.0f}  '
              f'(barrier -> published {d[1] + d[2]:.0f} clocks = {(d[1] + d[2]) * us:.2f} us)')
        # hand-off: from this block's publish of step s to its own contraction wave's satisfied poll of step s + 1
        if p[:, 0].any():
            ho = (p[2:, 2] - p[1:-1, 11]).mean()
            print(f'    published(s) -> poll satisfied(s + 1): {ho:.0f} clocks = {ho * us:.2f} us')
