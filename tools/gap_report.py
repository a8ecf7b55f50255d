"""Where the device idles inside a train step: the rocprofv3 kernel / memory-copy traces of a bench run, one step's timeline.

    rocprofv3 --kernel-trace --memory-copy-trace --output-format csv -d OUT -- python bench.py --headline-only --no-cpu-baseline ...
    python tools/gap_report.py OUT [first_kernel_name_fragment]

Takes the LAST complete step (from one `logmel_kernel` launch to the next), merges kernel and copy records into one timeline
ordered by start time, and prints: busy time, idle time, the idle intervals by the pair of operations around them (summed over
the step), and the copies.  Durations are the trace's own begin / end timestamps (ns).
"""
import csv
import glob
import os
import sys
from collections import defaultdict


def load(pattern, kind):
    rows = []
    for path in glob.glob(pattern, recursive=True):
        with open(path) as f:
            for r in csv.DictReader(f):
                name = r.get('Kernel_Name') or r.get('Name') or r.get('Direction') or kind
                start = int(r.get('Start_Timestamp') or r.get('Start') or 0)
                end = int(r.get('End_Timestamp') or r.get('End') or 0)
                if end > start:
                    rows.append((start, end, kind, name))
    return rows


def short(name):
    name = name.replace('pbsed::', '').replace('void ', '')
    cut = name.find('(')
    return (name if cut < 0 else name[:cut])[:70]


out = sys.argv[1]
marker = sys.argv[2] if len(sys.argv) > 2 else 'logmel_kernel'
ops = load(os.path.join(out, '**', '*kernel_trace.csv'), 'kernel') + load(os.path.join(out, '**', '*memory_copy_trace.csv'), 'copy')
ops.sort()
marks = [i for i, o in enumerate(ops) if marker in o[3]]
if len(marks) < 3:
    sys.exit(f'fewer than 3 launches of {marker} in the trace ({len(ops)} records)')
# the step with the most device operations among the last marker-to-marker intervals (bench.py also launches the front-end on
# its own at the end)
a, b = max(zip(marks[-12:-1], marks[-11:]), key=lambda ab: ab[1] - ab[0])
step = ops[a:b]
t0, t1 = step[0][0], ops[b][0]
busy_until, busy, gaps = t0, 0, defaultdict(lambda: [0, 0])
prev = None
for s, e, kind, name in step:
    if s > busy_until:
        if prev is not None:
            g = gaps[(short(prev), short(name))]
            g[0] += s - busy_until
            g[1] += 1
        busy_until = s
    if e > busy_until:
        busy += e - busy_until
        busy_until = e
    prev = name
tail = t1 - busy_until
print(f'step: {(t1 - t0) / 1e6:.3f} ms from {short(step[0][3])} to the next one; {len(step)} device operations '
      f'({sum(1 for o in step if o[2] == "kernel")} kernels, {sum(1 for o in step if o[2] == "copy")} copies); '
      f'busy {busy / 1e6:.3f} ms, idle {(t1 - t0 - busy) / 1e6:.3f} ms (of it {tail / 1e6:.3f} ms behind the last operation)')
print('idle intervals by neighbours (ms total, count, us each):')
for (p, n), (ns, cnt) in sorted(gaps.items(), key=lambda kv: -kv[1][0])[:25]:
    print(f'  {ns / 1e6:7.3f}  x{cnt:<3d} {ns / cnt / 1e3:6.1f}   {p}  ->  {n}')
copies = defaultdict(lambda: [0, 0])
for s, e, kind, name in step:
    if kind == 'copy':
        copies[name][0] += e - s
        copies[name][1] += 1
for name, (ns, cnt) in copies.items():
    print(f'copies {name}: {cnt} per step, {ns / 1e3:.1f} us in all')
sizes = sorted(((e - s) / 1e3, short(n)) for s, e, k, n in step if k == 'kernel')[::-1]
print('longest kernels of the step (us): ' + ', '.join(f'{n} {d:.0f}' for d, n in sizes[:8]))
