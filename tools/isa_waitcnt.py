"""Static look at a kernel file's ISA for the gfx9 load/store counter hazard: loads and stores of a
wave count on ONE `vmcnt`, in issue order (no separate store counter before gfx10), so the wait for a load that has stores in
front of it also sits out the acknowledgement of those stores - and where the compiler cannot count them (branches, loop
headers) the wait is `s_waitcnt vmcnt(0)`.  For every kernel
of a .hip file: the number of VMEM loads / stores / MFMAs and the `vmcnt(0)` waits that have a store within the preceding
`--window` instructions (a store right in front of the wait: the wait pays its round trip).  CPU only (cross-compiles).

    python tools/isa_waitcnt.py pb_sed_amd/csrc/conv_winox3.hip [--window 120] [-D WX_DBG=0]
"""
import argparse
import os
import re
import subprocess
import sys
import tempfile

ap = argparse.ArgumentParser()
ap.add_argument('src')
ap.add_argument('--window', type=int, default=120)
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
is_store = re.compile(r'\b(buffer_store|global_store|flat_store|buffer_atomic|global_atomic|flat_atomic)')
is_load = re.compile(r'\b(buffer_load|global_load|flat_load)')
print(f'{"kernel":70s} {"loads":>6s} {"stores":>6s} {"mfma":>6s} {"vmcnt(0)":>8s} {"behind a store":>15s}')
for m in re.finditer(r'\n(_Z\w+):[^\n]*\n', text):
    name = m.group(1)
    end = text.find('.Lfunc_end', m.end())
    body = [l.strip() for l in text[m.end():end].split('\n')]
    ins = [l for l in body if l and not l.startswith((';', '.'))]
    if not any('s_endpgm' in l for l in ins):
        continue
    stores = [i for i, l in enumerate(ins) if is_store.search(l)]
    waits = [i for i, l in enumerate(ins) if 's_waitcnt' in l and 'vmcnt(0)' in l]
    hot = [w for w in waits if any(w - args.window <= s < w for s in stores)]
    try:
        demangled = subprocess.run(['c++filt', name], capture_output=True, text=True).stdout.strip() or name
    except OSError:
        demangled = name
    demangled = re.sub(r'\(.*', '', demangled).replace('void pbsed::', '')
    print(f'{demangled[:70]:70s} {sum(1 for l in ins if is_load.search(l)):6d} {len(stores):6d} {sum(1 for l in ins if "v_mfma" in l):6d} '
          f'{len(waits):8d} {len(hot):15d}')
