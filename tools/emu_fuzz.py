#!/usr/bin/env python
"""The randomised sweeps of tests/sweeps/*.py on the EMULATED device (no GPU needed): random shapes through the real kernel source on
the CPU (tests/emu), e.g. under AddressSanitizer -

    tools/emu_fuzz.py fuzz_conv.py 6 3
    ASAN_OPTIONS=detect_leaks=0 LD_PRELOAD=$(ls /opt/rocm/lib/llvm/lib/clang/*/lib/linux/libclang_rt.asan-x86_64.so) \\
        PBSED_EMU_OPT="-O0 -g -fsanitize=address -fno-omit-frame-pointer" tools/emu_fuzz.py fuzz_postproc.py 20 1

The sweep's source runs unchanged but for its device name ('cuda' -> 'cpu').  Slow (full-size random shapes take minutes per case):
a tool, not part of the test suites."""
import os
import sys
import tempfile

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import pytest  # noqa: E402

from tests.emu import cpu_device  # noqa: E402

path = sys.argv[1] if os.path.exists(sys.argv[1]) else os.path.join(ROOT, 'tests', 'sweeps', sys.argv[1])
src = open(path).read().replace("'cuda:0'", "'cpu'").replace("'cuda'", "'cpu'")
sys.argv = [path] + sys.argv[2:]
mp = pytest.MonkeyPatch()
for name in ('tests.test_gpu_ops', 'tests.test_gpu_model', 'tests.test_gpu_postproc', 'tests.test_gpu_configs'):     # sweeps that re-use a GPU test's body
    mod = __import__(name, fromlist=['DEV'])
    if getattr(mod, 'DEV', None) == 'cuda:0':
        mod.DEV = 'cpu'
with tempfile.TemporaryDirectory() as d, cpu_device.emulated_device(mp, cpu_device.EmulatedLibrary(d)):
    exec(compile(src, path, 'exec'), {'__name__': '__main__', '__file__': path})
mp.undo()
