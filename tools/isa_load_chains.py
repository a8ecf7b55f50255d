"""Static look (CPU, cross-compile) for SERIALISED load groups: `load .. s_waitcnt vmcnt(0) .. load` inside one loop trip.

Written as `off = ok ? offset : OOB; load(off)` a raw buffer load can come back as TWO load sites in divergent branches (the
out-of-range one and the real one, same destination registers); the real one's address is then computed into a register the
other site's load is still writing, and the compiler puts `s_waitcnt vmcnt(0)` in front of it - inside the group of loads that was
meant to be in flight together: every load of the group waits for the round trip of the one before (found in conv_s16's
load_tile: four drains per tile, three of them on loads issued a few instructions earlier).  The cure is a branch-free select
(`(off & ok) | (OOB & ~ok)` with `ok` built from `&`, not `&&`).  For every kernel: the `vmcnt(0)` waits that have a VMEM load within
`--near` instructions on BOTH sides, inside a loop.

    python tools/isa_load_chains.py pb_sed_amd/csrc/conv_s16.hip [--near 30] [-k name]
"""
import argparse
import os
import re
import subprocess
import sys
import tempfile

ap = argparse.ArgumentParser()
ap.add_argument('src')
ap.add_argument('-k', default='')
ap.add_argument('--near', type=int, default=30)
ap.add_argument('-D', action='append', default=[])
args = ap.parse_args()
src = os.path.abspath(args.src)
with tempfile.TemporaryDirectory() as tmp:
    out = os.path.join(tmp, 'k.s')
    cmd = ['/opt/rocm/bin/hipcc', '--offload-arch=gfx950', '-O3', '-std=c++17', '-fPIC', '-Wno-unused-result', '-Wno-unused-value',
           '-S', '--cuda-device-only', src, '-o', out] + [f'-D{d}' for d in args.D]
    r = subprocess.run(cmd, cwd=os.path.dirname(src), capture_output=True, text=True)
    if r.returncode:
        sys.exit(r.stderr[-2000:])
    text = open(out).read()
is_load = re.compile(r'^(buffer_load|global_load|flat_load)')
is_wide = re.compile(r'^(buffer_load|global_load|flat_load)_dwordx[234]')      # the bulk loads (byte / dword side loads of untaken paths are noise)
print(f'{"kernel":70s} {"VMEM loads":>10s} {"vmcnt(0)":>8s} {"between loads, in a loop":>25s}')
for m in re.finditer(r'\n(_Z\w+):[^\n]*\n', text):
    name = m.group(1)
    end = text.find('.Lfunc_end', m.end())
    body = [l.strip() for l in text[m.end():end].split('\n')]
    ins = [l for l in body if l and not l.startswith(';') and (not l.startswith('.') or l.startswith('.LBB'))]
    if not any('s_endpgm' in l for l in ins):
        continue
    dem = subprocess.run(['c++filt', name], capture_output=True, text=True).stdout.strip() or name
    dem = re.sub(r'\(.*', '', dem).replace('void pbsed::', '')
    if args.k not in dem:
        continue
    label_at = {l.split(':')[0]: i for i, l in enumerate(ins) if l.startswith('.LBB')}
    loops = []
    for i, l in enumerate(ins):
        t = l.split()
        if t[0].startswith(('s_cbranch', 's_branch')) and len(t) > 1 and t[1] in label_at and label_at[t[1]] < i:
            loops.append((label_at[t[1]], i))
    loads = [i for i, l in enumerate(ins) if is_load.match(l)]
    waits = [i for i, l in enumerate(ins) if l.startswith('s_waitcnt') and 'vmcnt(0)' in l]
    wide = [i for i in loads if is_wide.match(ins[i])]
    chained = [w for w in waits if any(w - args.near <= x < w for x in wide) and any(w < x <= w + args.near for x in wide)
               and any(h <= w <= t for h, t in loops)]
    print(f'{dem[:70]:70s} {len(loads):10d} {len(waits):8d} {len(chained):25d}')
