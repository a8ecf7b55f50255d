"""CPU-side checks: the C-ABI library builds/loads and exports every symbol include/pbsed.h declares
(no compute calls without a GPU), and the product path refuses to run on CPU tensors."""
import ctypes
import os
import re

import pytest
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


@pytest.fixture(scope='module')
def libpath():
    from pb_sed_amd import _lib
    if not os.path.exists(_lib.LIB_PATH):
        import __graft_entry__
        __graft_entry__.build()
    return _lib.LIB_PATH


def test_header_symbols_exported(libpath):
    hdr = open(os.path.join(ROOT, 'include', 'pbsed.h')).read()
    names = sorted(set(re.findall(r'\b(pbsed_[a-z0-9_]+)\s*\(', hdr)))
    assert len(names) >= 20
    lib = ctypes.CDLL(libpath)
    missing = [n for n in names if not hasattr(lib, n)]
    assert not missing, missing
    from pb_sed_amd import _lib
    assert set(_lib.SIGNATURES) == set(names), set(_lib.SIGNATURES) ^ set(names)


def test_binding_loads_and_reports_version(libpath):
    from pb_sed_amd import _lib
    assert _lib.lib().pbsed_version() >= 1
    assert _lib.lib().pbsed_last_error() is not None


def test_no_cpu_fallback():
    from pb_sed_amd import ops
    from pb_sed_amd.modules import get_fbanks
    with pytest.raises(RuntimeError, match='no CPU fallback'):
        ops.conv_fwd(torch.zeros(1, 1, 4, 8), ops.PackedConv(torch.zeros(16, 1, 3, 3)), None)
    with pytest.raises(RuntimeError, match='no CPU fallback'):
        ops.logmel_fwd(torch.zeros(1, 16000), None, None, None, 50)


def test_pack_dims_host_logic(libpath):
    from pb_sed_amd import _lib
    i, o = ctypes.c_int(), ctypes.c_int()
    for (kh, kw, cin, cout, dg), exp in {(3, 3, 1, 16, 0): (4, 16), (3, 3, 16, 32, 0): (16, 32), (3, 3, 1, 32, 0): (8, 32), (3, 3, 3, 160, 0): (8, 192),
                                          (3, 3, 128, 256, 0): (128, 256), (1, 1, 2048, 256, 0): (2048, 256),
                                          (1, 1, 256, 10, 0): (256, 16), (3, 3, 128, 256, 1): (256, 128),
                                          (1, 1, 266, 768, 0): (272, 768)}.items():
        _lib.lib().pbsed_conv_pack_dims(kh, kw, cin, cout, dg, ctypes.byref(i), ctypes.byref(o))
        assert (i.value, o.value) == exp, ((kh, kw, cin, cout, dg), i.value, o.value)


def test_model_structure_matches_oracle():
    from oracle import models as om
    from pb_sed_amd.models import strong_label, weak_label
    a, b = weak_label.CRNN.build(), om.FBCRNN.build()
    assert [(k, tuple(v.shape)) for k, v in a.state_dict().items()] == \
           [(k, tuple(v.shape)) for k, v in b.state_dict().items()]
    assert sum(p.numel() for p in a.parameters()) == 3493188
    a, b = strong_label.CRNN.build(), om.BiCRNN.build()
    assert [(k, tuple(v.shape)) for k, v in a.state_dict().items()] == \
           [(k, tuple(v.shape)) for k, v in b.state_dict().items()]


def test_oversized_clip_is_refused_before_any_launch(libpath):
    """The conv loaders / epilogues address one clip with 32-bit offsets; the entry points refuse larger clips with
    PBSED_E_ARG and a message instead of launching (argument check only: null pointers, no GPU needed)."""
    from pb_sed_amd import _lib
    L = _lib.lib()
    # x, w_packed, bias, scale, shift, relu, seq_len, y, pool_idx, stats, stats_per_cf, B, Cin, Cout, F, T, KH, KW, pool, stream
    rc = L.pbsed_conv_fwd(None, None, None, None, None, 1, None, None, None, None, 0, 1, 4096, 64, 512, 100000, 3, 3, 0, None)
    assert rc != 0 and b'clip' in L.pbsed_last_error()
    rc = L.pbsed_conv_fwd_wino(None, None, None, None, None, 1, None, None, None, None, 0, 1, 4096, 64, 512, 100000, 0, None)
    assert rc != 0 and b'clip' in L.pbsed_last_error()
    # x, g, unpool_idx, ..., see include/pbsed.h: B, Cin, Cout, F, T, KH, KW
    rc = L.pbsed_conv_bwd_weight(None, None, None, 1, None, None, None, None, None, 1, 4096, 64, 512, 100000, 3, 3, None)
    assert rc != 0 and b'clip' in L.pbsed_last_error()


@pytest.mark.parametrize('mode', ['asan', 'tsan'])
def test_host_side_of_the_library_is_clean_under_the_sanitizers(tmp_path, mode):
    """SURVEY.md section 5 (sanitizers): a sanitizer build of the HOST side of the library (device code untouched), every one
    of its entry points driven through its argument checks, argument structs, pointer tables, tile lists and error strings with
    host-valid arguments (tools/asan_host_check.sh, tools/asan_host_drive.py; no GPU needed: the launchers reject the shape or
    stop at their first HIP call).  'asan': AddressSanitizer + UndefinedBehaviorSanitizer, one thread; 'tsan': ThreadSanitizer, four
    threads sweeping concurrently (the prefetch thread of the trainer calls into the library beside the main thread).  ~30 s each."""
    import subprocess
    r = subprocess.run(['bash', os.path.join(ROOT, 'tools', 'asan_host_check.sh'), str(tmp_path), mode], capture_output=True, text=True,
                       timeout=900)
    if r.returncode == 77:
        pytest.skip('no AddressSanitizer runtime in this image')
    assert r.returncode == 0, (r.stdout[-3000:], r.stderr[-1500:])
    assert 'DONE 9' in r.stdout or 'DONE 1' in r.stdout, r.stdout[-500:]          # DONE <n entry points>; ...
