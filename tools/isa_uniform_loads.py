"""Static look (CPU, cross-compile) for VMEM loads of per-tile UNIFORM values inside the loops of a kernel.

A `global_load_dword` through a pointer (`a.seq_len[b]`, `a.bias[c]`) whose index went through the VALU (an integer division has no
scalar form) is a vector-memory load even when every lane reads the same word, and the wait for it is `s_waitcnt vmcnt(..)` - in a
wave that also has stores or prefetch loads in flight that is `vmcnt(0)`: a full drain of the wave's memory queue once per trip
(loads and stores share vmcnt on gfx9).  For every kernel of a .hip file this prints the non-buffer
loads that sit inside a loop (a label that a later branch jumps back to), the loop's size and what else it holds, and the first
vmcnt wait behind each of them.

    python tools/isa_uniform_loads.py pb_sed_amd/csrc/conv_s16.hip [-k conv_s16_kernel] [-D NAME=VALUE]
"""
import argparse
import os
import re
import subprocess
import sys
import tempfile

ap = argparse.ArgumentParser()
ap.add_argument('src')
ap.add_argument('-k', default='', help='only kernels whose demangled name contains this')
ap.add_argument('-D', action='append', default=[])
ap.add_argument('--summary', action='store_true', help='one line per kernel')
ap.add_argument('--min-loop', type=int, default=150, help='ignore loops shorter than this many instructions (copy / spin loops)')
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

is_store = re.compile(r'^(buffer_store|global_store|flat_store|buffer_atomic|global_atomic|flat_atomic)')
is_bufload = re.compile(r'^buffer_load')
is_ptrload = re.compile(r'^(global_load|flat_load)')
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
    loops = []                                           # (head, tail) instruction ranges of back edges
    for i, l in enumerate(ins):
        t = l.split()
        if t[0].startswith(('s_cbranch', 's_branch')) and t[1] in label_at and label_at[t[1]] < i:
            loops.append((label_at[t[1]], i))
    loops = [lp for lp in loops if lp[1] - lp[0] >= args.min_loop]
    rows = []
    for i, l in enumerate(ins):
        if not is_ptrload.match(l):
            continue
        inside = [lp for lp in loops if lp[0] <= i <= lp[1]]
        if not inside:
            continue
        h, t = min(inside, key=lambda lp: lp[1] - lp[0])
        seg = ins[h:t + 1]
        wpos = next((j for j in range(i + 1, min(i + 400, len(ins))) if ins[j].startswith('s_waitcnt') and 'vmcnt' in ins[j]), -1)
        wait = ins[wpos].split(';')[0].strip() if wpos >= 0 else '-'
        rows.append((i, l.split()[0], t - h + 1, sum(1 for x in seg if is_bufload.match(x)), sum(1 for x in seg if is_store.match(x)),
                     sum(1 for x in seg if 'v_mfma' in x), wait, wpos))
    if args.summary:
        if rows:
            big = [r_ for r_ in rows if r_[5] > 0]                         # inside a loop that also runs MFMAs: the tile / chunk loop
            sites = {r_[7] for r_ in big if r_[6].endswith('vmcnt(0)')}      # distinct drains (loads requested together share one)
            print(f'{dem[:66]:66s} pointer loads in loops {len(rows):3d}, in MFMA loops {len(big):3d}, distinct vmcnt(0) waits behind them '
                  f'{len(sites):3d}')
        continue
    if rows:
        print(f'== {dem}')
        print(f'   {"at":>6s} {"load":22s} {"loop instr":>10s} {"buffer loads":>12s} {"stores":>6s} {"mfma":>5s}  first vmcnt wait behind it')
        for r_ in rows:
            print(f'   {r_[0]:6d} {r_[1]:22s} {r_[2]:10d} {r_[3]:12d} {r_[4]:6d} {r_[5]:5d}  {r_[6]}')
