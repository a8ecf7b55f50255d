"""Drive EVERY entry point of the C-ABI library with host-valid arguments, no GPU needed - meant to run on an AddressSanitizer build of
the HOST side of the library (tools/asan_host_check.sh; SURVEY.md section 5: sanitizers).  Pointer parameters that the ABI reads or
writes on the host (int / float / double / byte arrays, tables of device pointers) get real 4 096-element buffers, device pointers
are NULL, every integer parameter is set to one value per sweep (0, 1, 4, 17, 1000, -3): the launchers must reject the shape, or
get as far as their first HIP call and report the missing device - and on the way fill their argument structs, pointer tables,
tile lists and error strings, which is what the sanitizer watches.  PBSED_DRIVE_THREADS=4: four threads sweep concurrently (the
ThreadSanitizer pass: per-device tables, the scratch registry, one-time attribute settings are shared host state; the error
string is thread-local).  Prints DONE and the tally of return codes.

    LD_PRELOAD=<libclang_rt.asan> PBSED_LIB=<asan build> python tools/asan_host_drive.py
"""
import collections
import ctypes as C
import faulthandler
import os
import sys

faulthandler.enable()
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from pb_sed_amd import _lib     # noqa: E402

lib = _lib.lib()
INTS = (C.c_int, C.c_size_t, C.c_uint, C.c_long, C.c_ulong, C.c_longlong, C.c_ulonglong)
SIGNED = (C.c_int, C.c_long, C.c_longlong)


def argument(t, ival):
    if t in INTS:
        return ival if (ival >= 0 or t in SIGNED) else 0
    if t in (C.c_float, C.c_double):
        return .5
    if t is C.c_void_p:
        return None                                  # a device pointer: nothing may dereference it on the host
    if hasattr(t, '_type_') and not isinstance(t._type_, str):
        return (t._type_ * 4096)()                   # host arrays / pointer tables
    return None


tally = collections.Counter()
THREADS = int(os.environ.get('PBSED_DRIVE_THREADS', '1'))     # > 1: the ThreadSanitizer pass (ctypes releases the GIL in the calls)


def sweep(quiet):
    for name in sorted(_lib.SIGNATURES):
        fn = getattr(lib, name)
        for ival in (0, 1, 4, 17, 1000, -3):
            if not quiet:
                print('CALL', name, ival, flush=True)    # (the last line before a sanitizer report names the culprit)
            r = fn(*[argument(t, ival) for t in _lib.SIGNATURES[name]])
            tally[r if isinstance(r, int) and r <= 0 else 'value'] += 1


if THREADS > 1:
    import threading
    ts = [threading.Thread(target=lambda: [sweep(True) for _ in range(2)]) for _ in range(THREADS)]
    [t.start() for t in ts]
    [t.join() for t in ts]
else:
    sweep(False)
print('DONE', len(_lib.SIGNATURES), 'entry points;', dict(tally))
