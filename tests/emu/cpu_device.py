"""TEST INFRASTRUCTURE: the whole product path - pb_sed_amd's Python host side over the C-ABI - on the CPU, without a GPU.

``EmulatedLibrary`` serves the entry points of ``include/pbsed.h`` from the emulated translation units of this directory (every
``csrc/*.hip`` file compiled for x86 against ``shim/hip/hip_runtime.h``: HIP threads are fibers, MFMA / DPP / barriers are
rendezvous points); ``emulated_device`` swaps it in for ``libpbsed_mi355.so`` and stands a few no-op objects in for the
``torch.cuda`` stream / event calls of the host side, so that the models, the engine and the trainer run unchanged on CPU tensors:
the SAME Python code and the SAME kernel source as on the MI355X, only the instruction set differs.

This is a checker that lives under tests/ - the package itself knows nothing about it, has no CPU path and still refuses CPU tensors
(``_lib.require_gpu``) and a missing library.  What it proves: index maps, masks, fragment layouts, launch arguments and the order of
launches - the arithmetic results of the device code.  What it cannot prove: anything about timing, the GPU's memory model, or
concurrency between streams (a "stream" here runs in program order).
"""
import contextlib
import ctypes as C
import os
import subprocess
import time
from concurrent.futures import ThreadPoolExecutor

import torch

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
CLANG = '/opt/rocm/lib/llvm/bin/clang++'
OPT = os.environ.get('PBSED_EMU_OPT', '-O0').split()
UNITS = ('api', 'conv_s16', 'conv_winox3', 'conv_wino', 'conv_bf16', 'conv1d_pc', 'logmel', 'gru', 'gru_stack', 'rnn_gemms', 'postproc')


def compile_unit(unit, csrc, out, defines=()):
    # -O0: the units are template-heavy (tens of seconds each at -O1, seconds at -O0) and the emulated launches are small
    subprocess.run([CLANG, '-x', 'c++', '-std=c++20', *OPT, '-fPIC', '-shared', '-w', *[f'-D{d}' for d in defines],
                    '-I', os.path.join(HERE, 'shim'), '-I', csrc, os.path.join(HERE, f'emu_{unit}.cpp'),
                    os.path.join(HERE, 'hipemu_runtime.cpp'), '-o', out], check=True)
    return out


class EmulatedLibrary:
    """Attribute access like a ``ctypes.CDLL`` of libpbsed_mi355.so, resolved over the emulated units (argtypes from _lib.SIGNATURES)."""

    def __init__(self, outdir, csrc=None, reuse=False):
        """``reuse``: load the units another EmulatedLibrary has built in ``outdir`` (worker processes of a multi-rank test)."""
        csrc = csrc or os.path.join(ROOT, 'pb_sed_amd', 'csrc')
        self.outdir = str(outdir)
        paths = [os.path.join(self.outdir, f'libemu_{u}.so') for u in UNITS]
        if not (reuse and all(os.path.exists(p) for p in paths)):
            with ThreadPoolExecutor(8) as ex:
                list(ex.map(lambda u: compile_unit(u, csrc, os.path.join(self.outdir, f'libemu_{u}.so')), UNITS))
        self._units = {u: C.CDLL(p) for u, p in zip(UNITS, paths)}
        self._units['gru_stack'].hipemu_set_concurrent(1)      # the persistent scans: every workgroup on an OS thread of its own
        self._fns = {}
        self._last = 'api'
        self.calls = []                                        # entry points in call order (a test may look at what ran)
        self.seconds = {}                                      # entry point -> wall-clock spent in it

    def _resolve(self, name):
        from pb_sed_amd import _lib
        for unit, dll in self._units.items():
            try:
                fn = getattr(dll, name)
            except AttributeError:
                continue
            fn.argtypes = _lib.SIGNATURES[name]
            fn.restype = _lib._NON_STATUS.get(name, C.c_int)
            return unit, fn
        raise AttributeError(f'{name}: not exported by an emulated unit')

    def __getattr__(self, name):
        if name == 'pbsed_last_error':
            unit = self._last
            fn = getattr(self._units[unit], 'pbsed_last_error' if unit == 'api' else 'emu_last_error')
            fn.restype = C.c_char_p
            return fn
        if name not in self._fns:
            self._fns[name] = self._resolve(name)
        unit, fn = self._fns[name]

        def call(*args):
            self._last = unit
            self.calls.append(name)
            t0 = time.perf_counter()
            try:
                return fn(*args)
            finally:
                self.seconds[name] = self.seconds.get(name, 0.) + time.perf_counter() - t0
        return call


class _Stream:
    cuda_stream = 0

    def __init__(self, *a, **k):
        pass

    def wait_stream(self, other):
        pass

    def wait_event(self, ev):
        pass

    def synchronize(self):
        pass


class _Event:
    def __init__(self, *a, **k):
        pass

    def record(self, stream=None):
        pass

    def wait(self, stream=None):
        pass

    def synchronize(self):
        pass

    def query(self):
        return True

    def elapsed_time(self, other):
        return 0.


@contextlib.contextmanager
def emulated_device(monkeypatch, library, to_copies=True):
    """Inside: pb_sed_amd runs on CPU tensors through ``library``.  Everything patched is restored by ``monkeypatch``.
    ``to_copies``: ``tensor.to('cpu')`` hands back a COPY, as ``.to('cuda:0')`` of a host tensor does (ops.host_to_device stages through pooled
    pinned buffers that are re-used after the copy; the GPU tests keep host originals as references while a launch updates the device
    copy in place)."""
    from pb_sed_amd import _lib, ops
    stream = _Stream()
    monkeypatch.setattr(ops, 'ensure_scratch', lambda device: None)      # the library's own (shim-malloc'ed) scratch serves the launches
    monkeypatch.setattr(_lib, '_lib', library)
    monkeypatch.setattr(_lib, 'stream', lambda: None)
    monkeypatch.setattr(_lib, 'require_gpu', lambda t: None)
    for name, obj in (('Stream', _Stream), ('Event', _Event), ('current_stream', lambda device=None: stream),
                      ('stream', lambda s: contextlib.nullcontext()), ('device', lambda d: contextlib.nullcontext()),
                      ('current_device', lambda: 0), ('synchronize', lambda device=None: None)):
        monkeypatch.setattr(torch.cuda, name, obj)
    # uninitialised memory is POISON here (NaN in floating-point tensors, 0xAA.. in integer ones): a launch that leaves part of its
    # output unwritten, or a consumer that reads what no launch wrote, shows up in the results instead of hiding behind the zeros of
    # a fresh allocation
    def poisoned(t):
        if t.numel() and os.environ.get('PBSED_EMU_POISON', '1') == '1':
            if t.dtype.is_floating_point:
                t.fill_(float('nan'))
            elif t.dtype in (torch.uint8, torch.int8, torch.int16, torch.int32, torch.int64):
                t.fill_({torch.uint8: 0xAA, torch.int8: -86, torch.int16: -21846, torch.int32: -1431655766}.get(t.dtype, -6148914691236517206))
        return t
    empty, empty_like, new_empty = torch.empty, torch.empty_like, torch.Tensor.new_empty
    monkeypatch.setattr(torch, 'empty', lambda *a, pin_memory=False, **k: poisoned(empty(*a, **k)))
    monkeypatch.setattr(torch, 'empty_like', lambda *a, pin_memory=False, **k: poisoned(empty_like(*a, **k)))
    monkeypatch.setattr(torch.Tensor, 'new_empty', lambda self, *a, pin_memory=False, **k: poisoned(new_empty(self, *a, **k)))
    monkeypatch.setattr(torch.Tensor, 'pin_memory', lambda self, *a, **k: self)
    if to_copies:
        to = torch.Tensor.to

        def to_device(self, *a, **k):
            out = to(self, *a, **k)
            first = a[0] if a else k.get('device')
            return out.clone() if out is self and isinstance(first, (str, torch.device)) and str(first) == 'cpu' else out
        monkeypatch.setattr(torch.Tensor, 'to', to_device)
    yield library
