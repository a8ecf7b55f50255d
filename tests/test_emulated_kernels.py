"""Kernel translation units EXECUTED ON THE CPU (no GPU needed): ``tests/emu/shim/hip/hip_runtime.h`` stands in for the HIP runtime -
every HIP thread of a block is a fiber, cross-lane instructions (MFMA, DPP, shuffles, readfirstlane) and ``__syncthreads`` are
rendezvous points, the amdgcn builtins are ordinary functions - so a ``csrc/*.hip`` file compiles for x86 as it is and its
``extern "C"`` entry points run on host pointers.  Two uses:

* the kernels of the TREE against a plain float64 restatement of the op (a CPU-side sanity check of the device code itself:
  index maps, masks, fragment layouts, the bf16x3 arithmetic);
* the tree + the PARKED PATCHES of ``tools/micro/attic`` (re-orderings / equivalent re-writings of loads that wait
  for a GPU measurement) against the tree: the same emulator, the same inputs - outputs, pool indices and statistics must be
  BIT-IDENTICAL.  That is the functional half of "the patch changes nothing but the waits"; the timing half needs hardware.

Reference op sites: the few-channel 3x3 layers of pb_sed/experiments/weak_label_crnn/training.py:161-168.
"""
import ctypes as C
import os
import re
import shutil
import subprocess

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
EMU = os.path.join(ROOT, 'tests', 'emu')
CLANG = '/opt/rocm/lib/llvm/bin/clang++'
pytestmark = pytest.mark.skipif(not os.path.exists(CLANG), reason='needs the ROCm clang++ (ext_vector_type, __bf16)')


def _build(unit, csrc, out):
    # -O0: these translation units are template-heavy (28 s at -O1, 2 s at -O0) and the emulated launches are tiny
    subprocess.run([CLANG, '-x', 'c++', '-std=c++20', *os.environ.get('PBSED_EMU_OPT', '-O0').split(), '-fPIC', '-shared', '-w',
                    '-I', os.path.join(EMU, 'shim'), '-I', csrc, os.path.join(EMU, unit), os.path.join(EMU, 'hipemu_runtime.cpp'),
                    '-o', out], check=True)
    return out



@pytest.fixture(scope='module')
def patched_csrc(tmp_path_factory):
    """csrc of the tree with the whole parked patch stack applied (the order of tools/build_variants.sh)."""
    work = tmp_path_factory.mktemp('patched')
    (work / 'pb_sed_amd').mkdir()
    shutil.copytree(os.path.join(ROOT, 'pb_sed_amd', 'csrc'), work / 'pb_sed_amd' / 'csrc', ignore=shutil.ignore_patterns('build'))
    script = open(os.path.join(ROOT, 'tools', 'build_variants.sh')).read()
    for p in re.findall(r'attic/(\w+\.patch)\)', script):
        subprocess.run(['patch', '-s', '-p1', '-i', os.path.join(ROOT, 'tools', 'micro', 'attic', p)], cwd=work, check=True)
    return str(work / 'pb_sed_amd' / 'csrc')


_PATCHED_UNITS = ('emu_conv_s16.cpp', 'emu_conv_winox3.cpp', 'emu_logmel.cpp', 'emu_conv_bf16.cpp', 'emu_conv1d_pc.cpp', 'emu_conv_wgrad.cpp')
_TREE_UNITS = _PATCHED_UNITS + ('emu_gru_stack.cpp', 'emu_rnn_gemms.cpp', 'emu_misc.cpp', 'emu_postproc.cpp')


@pytest.fixture(scope='module')
def built(tmp_path_factory, patched_csrc):
    """Every emulated unit of this module, from the tree and (where a parked patch touches it) from the patched tree, compiled ONCE
    and in parallel; built(unit, 'tree' | 'patched') -> CDLL."""
    from concurrent.futures import ThreadPoolExecutor
    d = tmp_path_factory.mktemp('emu_units')
    tree = os.path.join(ROOT, 'pb_sed_amd', 'csrc')
    jobs = [(u, tree, 'tree') for u in _TREE_UNITS] + [(u, patched_csrc, 'patched') for u in _PATCHED_UNITS]
    path = lambda u, which: str(d / f'{u[:-4]}_{which}.so')
    with ThreadPoolExecutor(8) as ex:
        list(ex.map(lambda j: _build(j[0], j[1], path(j[0], j[2])), jobs))
    return lambda unit, which='tree': C.CDLL(path(unit, which))


@pytest.fixture(scope='module')
def s16_libs(built):
    return built('emu_conv_s16.cpp', 'tree'), built('emu_conv_s16.cpp', 'patched')


def P(a):
    return None if a is None else a.ctypes.data_as(C.c_void_p)


def _bits(a):
    return a.view(np.uint32 if a.dtype == np.float32 else np.uint64 if a.dtype == np.float64 else a.dtype)


def _conv3x3_f64(x, w):
    b, cin, f, t = x.shape
    xp = np.pad(x, ((0, 0), (0, 0), (1, 1), (1, 1)))
    y = np.zeros((b, w.shape[0], f, t))
    for kh in range(3):
        for kw in range(3):
            y += np.einsum('oc,bcft->boft', w[:, :, kh, kw], xp[:, :, kh:kh + f, kw:kw + t])
    return y


def _pack(lib, w, dgrad):
    cout, cin = w.shape[:2]
    outp = 16 if (cin if dgrad else cout) <= 16 else 32
    up = np.zeros(5 * 3 * (outp // 16) * 512, np.uint16)
    assert lib.pbsed_pack_conv_weights_s16(P(w), P(up), cout, cin, dgrad, None) == 0
    return up


S16_FWD = [  # (B, Cin, Cout, F, T, pool, prologue, per_cf)
    (2, 16, 16, 8, 128, 0, True, 0),          # layer 2 of 'shallow' in small
    (2, 16, 16, 8, 100, 1, True, 0),          # + (2,1) pool, T no multiple of the 64-wide tile, ragged lengths
    (1, 16, 32, 6, 68, 0, True, 1),           # 32 output channels (2 x 2 waves), F no multiple of the 4-row tile, per-(channel, row) statistics
    (2, 11, 16, 4, 64, 0, False, 0),          # the tag-conditioned first layer of the BiCRNN: 11 input channels, no prologue
]


@pytest.mark.parametrize('case', S16_FWD, ids=lambda c: 'x'.join(str(v) for v in c))
def test_conv_s16_forward_on_the_cpu_tree_vs_float64_and_patched_vs_tree(s16_libs, case):
    b, cin, cout, f, t, pool, pro, per_cf = case
    rng = np.random.RandomState(sum(case))
    x = rng.randn(b, cin, f, t).astype(np.float32)
    w = (rng.randn(cout, cin, 3, 3) * .1).astype(np.float32)
    bias = rng.randn(cout).astype(np.float32)
    scale = (rng.rand(cin) + .5).astype(np.float32) if pro else None
    shift = (rng.randn(cin) * .1).astype(np.float32) if pro else None
    seq = np.array([t] + [int(t * .7)] * (b - 1), np.int32)
    fo = f // 2 if pool else f
    outs = []
    for lib in s16_libs:
        up = _pack(lib, w, 0)
        y = np.full((b, cout, fo, t), np.nan, np.float32)
        idx = np.full((b, cout, fo, t), 7, np.uint8) if pool else None
        stats = np.zeros((32, cout * (fo if per_cf else 1), 2), np.float64)
        rc = lib.pbsed_conv_fwd_s16(P(x), P(up), P(bias), P(scale), P(shift), 1, P(seq), P(y), P(idx), P(stats), per_cf, b, cin, cout,
                                    f, t, pool, None)
        assert rc == 0, lib.emu_last_error()
        outs.append((y, idx, stats))
    (y, idx, stats), (y2, idx2, stats2) = outs
    # --- tree vs float64
    xa = x.astype(np.float64)
    if pro:
        xa = np.maximum(xa * scale[None, :, None, None] + shift[None, :, None, None], 0)
        for i in range(b):
            xa[i, :, :, seq[i]:] = 0                             # Normalization re-masks its output; zero padding is post-activation
    ref = _conv3x3_f64(xa, w.astype(np.float64)) + bias[None, :, None, None]
    if pool:
        pair = ref.reshape(b, cout, fo, 2, t)
        ref_idx = (pair[:, :, :, 1] > pair[:, :, :, 0]).astype(np.uint8)
        ref = pair.max(3)
        near_tie = np.abs(pair[:, :, :, 1] - pair[:, :, :, 0]) < 1e-5
        assert np.array_equal(idx[~near_tie], ref_idx[~near_tie])
    assert not np.isnan(y).any()
    assert np.abs(y - ref).max() < 2e-5 * max(1., np.abs(ref).max())
    masked = ref.copy()
    for i in range(b):
        masked[i, :, :, seq[i]:] = 0
    s_ref = masked.sum((0, 3)) if per_cf else masked.sum((0, 2, 3))
    s_got = stats.sum(0)[:, 0].reshape(s_ref.shape)
    assert np.abs(s_got - s_ref).max() < 1e-3 * max(1., np.abs(s_ref).max())
    # --- patched vs tree: bit for bit
    assert np.array_equal(_bits(y2), _bits(y))
    assert np.array_equal(_bits(stats2), _bits(stats))
    if pool:
        assert np.array_equal(idx2, idx)


S16_BWD = [  # (B, Cin (produced), Cout (contracted), F, T, unpool, bn)
    (2, 16, 16, 8, 128, False, True),         # data gradient through a norm + ReLU
    (2, 16, 16, 8, 100, True, True),          # ... of a pooled layer (un-pooling through the argmax bytes)
    (1, 32, 16, 4, 64, False, True),          # 32 produced channels: the variant that keeps its constants in the channel loop
    (2, 11, 16, 4, 64, False, False),         # plain store (the first layer: no norm below)
]


@pytest.mark.parametrize('case', S16_BWD, ids=lambda c: 'x'.join(str(int(v)) for v in c))
def test_conv_s16_data_gradient_on_the_cpu_tree_vs_float64_and_patched_vs_tree(s16_libs, case):
    b, cin, cout, f, t, unpool, bn = case
    rng = np.random.RandomState(sum(int(v) for v in case) + 1)
    w = (rng.randn(cout, cin, 3, 3) * .1).astype(np.float32)
    fg = f // 2 if unpool else f
    g = rng.randn(b, cout, fg, t).astype(np.float32)
    uidx = (rng.rand(b, cout, fg, t) < .5).astype(np.uint8) if unpool else None
    seq = np.array([t] + [int(t * .8)] * (b - 1), np.int32)
    bx = rng.randn(b, cin, f, t).astype(np.float32) if bn else None
    bmean = (rng.randn(cin) * .1).astype(np.float32) if bn else None
    binv = (rng.rand(cin) + .5).astype(np.float32) if bn else None
    bscale = (rng.rand(cin) + .5).astype(np.float32) if bn else None
    bshift = (rng.randn(cin) * .1).astype(np.float32) if bn else None
    outs = []
    for lib in s16_libs:
        up = _pack(lib, w, 1)
        dz = np.full((b, cin, f, t), np.nan, np.float32)
        stats = np.zeros((32, cin, 2), np.float64)
        rc = lib.pbsed_conv_bwd_data_s16(P(g), P(up), P(uidx), P(seq), P(dz), P(bx), P(bmean), P(binv), P(bscale), P(bshift), 1, P(stats),
                                         b, cin, cout, f, t, None)
        assert rc == 0, lib.emu_last_error()
        outs.append((dz, stats))
    (dz, stats), (dz2, stats2) = outs
    # --- tree vs float64: dx = conv_transpose(g_unpooled, w) = conv3x3 with flipped, transposed weights
    gu = g.astype(np.float64)
    if unpool:
        full = np.zeros((b, cout, f, t))
        full[:, :, 0::2] = np.where(uidx == 0, gu, 0)
        full[:, :, 1::2] = np.where(uidx == 1, gu, 0)
        gu = full
    wt = np.flip(w.astype(np.float64), (2, 3)).transpose(1, 0, 2, 3)
    ref = _conv3x3_f64(gu, wt)
    if bn:
        z = bx.astype(np.float64) * bscale[None, :, None, None] + bshift[None, :, None, None]
        keep = z > 0
        for i in range(b):
            keep[i, :, :, seq[i]:] = False
        near = np.abs(z) < 1e-6
        ref = np.where(keep, ref, 0)
        ok = ~near
        assert np.abs(dz - ref)[ok].max() < 2e-5 * max(1., np.abs(ref).max())
        xhat = (bx.astype(np.float64) - bmean[None, :, None, None]) * binv[None, :, None, None]
        s1, s2 = ref.sum((0, 2, 3)), (ref * xhat).sum((0, 2, 3))
        got = stats.sum(0)
        assert np.abs(got[:, 0] - s1).max() < 1e-3 * max(1., np.abs(s1).max())
        assert np.abs(got[:, 1] - s2).max() < 1e-3 * max(1., np.abs(s2).max())
    else:
        assert np.abs(dz - ref).max() < 2e-5 * max(1., np.abs(ref).max())
    # --- patched vs tree: bit for bit
    assert np.array_equal(_bits(dz2), _bits(dz))
    assert np.array_equal(_bits(stats2), _bits(stats))


# ------------------------------------------------------------------------------------------------ conv_winox3 (Winograd F(4,3), bf16x3)
@pytest.fixture(scope='module')
def wx_libs(built):
    return built('emu_conv_winox3.cpp', 'tree'), built('emu_conv_winox3.cpp', 'patched')


def _pack_wx(lib, w, dgrad):
    cout, cin = w.shape[:2]
    inp, outp = C.c_int(), C.c_int()
    lib.pbsed_conv_pack_dims_winox3(cin, cout, dgrad, C.byref(inp), C.byref(outp))
    up = np.zeros(18 * inp.value * outp.value * 3, np.uint16)
    assert lib.pbsed_pack_conv_weights_winox3(P(w), P(up), cout, cin, dgrad, None) == 0
    return up


WX_FWD = [  # (B, Cin, Cout, F, T, pool, per_cf)
    (1, 32, 64, 4, 64, 0, 0),                 # one chunk, one cout tile
    (2, 64, 64, 8, 100, 1, 0),                # two chunks, (2,1) pool, ragged, T no multiple of the tile
    (1, 32, 32, 6, 68, 0, 1),                 # 32-cout blocks (two waves per cout tile), F no multiple of 4, per-(channel, row) statistics
    (1, 64, 128, 4, 64, 0, 0),                # two cout tiles per spatial tile: the item walk of the persistent blocks
]


@pytest.mark.parametrize('case', WX_FWD, ids=lambda c: 'x'.join(str(v) for v in c))
def test_conv_winox3_forward_on_the_cpu_tree_vs_float64_and_patched_vs_tree(wx_libs, case):
    b, cin, cout, f, t, pool, per_cf = case
    rng = np.random.RandomState(sum(case) + 7)
    x = rng.randn(b, cin, f, t).astype(np.float32)
    w = (rng.randn(cout, cin, 3, 3) * .1).astype(np.float32)
    bias = rng.randn(cout).astype(np.float32)
    scale = (rng.rand(cin) + .5).astype(np.float32)
    shift = (rng.randn(cin) * .1).astype(np.float32)
    seq = np.array([t] + [int(t * .7)] * (b - 1), np.int32)
    fo = f // 2 if pool else f
    outs = []
    for lib in wx_libs:
        up = _pack_wx(lib, w, 0)
        y = np.full((b, cout, fo, t), np.nan, np.float32)
        idx = np.full((b, cout, fo, t), 7, np.uint8) if pool else None
        stats = np.zeros((32, cout * (fo if per_cf else 1), 2), np.float64)
        rc = lib.pbsed_conv_fwd_winox3(P(x), P(up), P(bias), P(scale), P(shift), 1, P(seq), P(y), P(idx), P(stats), per_cf, b, cin, cout,
                                       f, t, pool, None)
        assert rc == 0
        outs.append((y, idx, stats))
    (y, idx, stats), (y2, idx2, stats2) = outs
    xa = np.maximum(x.astype(np.float64) * scale[None, :, None, None] + shift[None, :, None, None], 0)
    for i in range(b):
        xa[i, :, :, seq[i]:] = 0
    ref = _conv3x3_f64(xa, w.astype(np.float64)) + bias[None, :, None, None]
    if pool:
        pair = ref.reshape(b, cout, fo, 2, t)
        ref_idx = (pair[:, :, :, 1] > pair[:, :, :, 0]).astype(np.uint8)
        ref = pair.max(3)
        near_tie = np.abs(pair[:, :, :, 1] - pair[:, :, :, 0]) < 1e-4
        assert np.array_equal(idx[~near_tie], ref_idx[~near_tie])
    assert not np.isnan(y).any()
    assert np.abs(y - ref).max() < 5e-5 * max(1., np.abs(ref).max())          # Winograd F(4,3) in fp32-class arithmetic
    masked = ref.copy()
    for i in range(b):
        masked[i, :, :, seq[i]:] = 0
    s_ref = masked.sum((0, 3)) if per_cf else masked.sum((0, 2, 3))
    s_got = stats.sum(0)[:, 0].reshape(s_ref.shape)
    assert np.abs(s_got - s_ref).max() < 2e-3 * max(1., np.abs(s_ref).max())
    assert np.array_equal(_bits(y2), _bits(y))
    assert np.array_equal(_bits(stats2), _bits(stats))
    if pool:
        assert np.array_equal(idx2, idx)


WX_BWD = [  # (B, Cin (produced), Cout (contracted), F, T, unpool)
    (1, 64, 64, 4, 64, False),                # data gradient through a norm + ReLU (the re-load of the layer input, one pair ahead)
    (2, 64, 32, 8, 100, True),                # ... of a pooled layer, ragged
    (1, 32, 64, 4, 68, False),                # 32 produced channels: the 32-cout block form
]


@pytest.mark.parametrize('case', WX_BWD, ids=lambda c: 'x'.join(str(int(v)) for v in c))
def test_conv_winox3_data_gradient_on_the_cpu_tree_vs_float64_and_patched_vs_tree(wx_libs, case):
    b, cin, cout, f, t, unpool = case
    rng = np.random.RandomState(sum(int(v) for v in case) + 11)
    w = (rng.randn(cout, cin, 3, 3) * .1).astype(np.float32)
    fg = f // 2 if unpool else f
    g = rng.randn(b, cout, fg, t).astype(np.float32)
    uidx = (rng.rand(b, cout, fg, t) < .5).astype(np.uint8) if unpool else None
    seq = np.array([t] + [int(t * .8)] * (b - 1), np.int32)
    bx = rng.randn(b, cin, f, t).astype(np.float32)
    bmean = (rng.randn(cin) * .1).astype(np.float32)
    binv = (rng.rand(cin) + .5).astype(np.float32)
    bscale = (rng.rand(cin) + .5).astype(np.float32)
    bshift = (rng.randn(cin) * .1).astype(np.float32)
    outs = []
    for lib in wx_libs:
        up = _pack_wx(lib, w, 1)
        dz = np.full((b, cin, f, t), np.nan, np.float32)
        stats = np.zeros((32, cin, 2), np.float64)
        rc = lib.pbsed_conv_bwd_data_winox3(P(g), P(up), P(uidx), P(seq), P(dz), P(bx), P(bmean), P(binv), P(bscale), P(bshift), 1,
                                            P(stats), b, cin, cout, f, t, None)
        assert rc == 0
        outs.append((dz, stats))
    (dz, stats), (dz2, stats2) = outs
    gu = g.astype(np.float64)
    if unpool:
        full = np.zeros((b, cout, f, t))
        full[:, :, 0::2] = np.where(uidx == 0, gu, 0)
        full[:, :, 1::2] = np.where(uidx == 1, gu, 0)
        gu = full
    ref = _conv3x3_f64(gu, np.flip(w.astype(np.float64), (2, 3)).transpose(1, 0, 2, 3))
    z = bx.astype(np.float64) * bscale[None, :, None, None] + bshift[None, :, None, None]
    keep = z > 0
    for i in range(b):
        keep[i, :, :, seq[i]:] = False
    ref = np.where(keep, ref, 0)
    ok = np.abs(z) > 1e-6
    assert not np.isnan(dz).any()
    assert np.abs(dz - ref)[ok].max() < 5e-5 * max(1., np.abs(ref).max())
    xhat = (bx.astype(np.float64) - bmean[None, :, None, None]) * binv[None, :, None, None]
    got = stats.sum(0)
    assert np.abs(got[:, 0] - ref.sum((0, 2, 3))).max() < 2e-3 * max(1., np.abs(ref.sum((0, 2, 3))).max())
    assert np.abs(got[:, 1] - (ref * xhat).sum((0, 2, 3))).max() < 2e-3 * max(1., np.abs((ref * xhat).sum((0, 2, 3))).max())
    assert np.array_equal(_bits(dz2), _bits(dz))
    assert np.array_equal(_bits(stats2), _bits(stats))


# ------------------------------------------------------------------------------------------------ logmel (the fused front-end)
@pytest.fixture(scope='module')
def lm_libs(built):
    return built('emu_logmel.cpp', 'tree'), built('emu_logmel.cpp', 'patched')


def _logmel_tables(fb):
    """pb_sed_amd/ops.py::LogMelTables on the host (window, twiddles, packed sparse filterbank)."""
    fb = np.asarray(fb, np.float32)
    nz = fb > 0
    start = np.array([int(np.argmax(r)) if r.any() else 0 for r in nz], np.int32)
    end = np.array([len(r) - int(np.argmax(r[::-1])) if r.any() else 0 for r in nz], np.int32)
    length = (end - start).astype(np.int32)
    off = np.concatenate([[0], np.cumsum(length)[:-1]]).astype(np.int32)
    w = np.concatenate([fb[m, start[m]:end[m]] for m in range(fb.shape[0])]).astype(np.float32)
    k = np.arange(960, dtype=np.float64)
    win = (0.42 - 0.5 * np.cos(2 * np.pi * k / 960) + 0.08 * np.cos(4 * np.pi * k / 960)).astype(np.float32)
    q = np.arange(1024, dtype=np.float64)
    tw = np.stack([np.cos(-2 * np.pi * q / 1024), np.sin(-2 * np.pi * q / 1024)], -1).astype(np.float32)
    return win, tw, start, length, off, w


@pytest.mark.parametrize('warped', [False, True], ids=['static_filterbank', 'warped_mel'])
def test_logmel_front_end_on_the_cpu_tree_vs_oracle_and_patched_vs_tree(lm_libs, warped):
    """The fused STFT -> power -> mel -> log -> norm -> clamp -> mask launch (two clips, 23 frames: one full 16-frame tile and a
    ragged one), with the statistics sums of a training step, against oracle/frontend.py in float64; then the patched kernel
    (one request group in the set-up, split descriptor load / finish, template on WARPED) bit for bit against the tree's."""
    import torch
    from oracle import frontend as ofe
    rng = np.random.RandomState(3 + warped)
    b, n = 2, 320 * 22 + 1
    t = ofe.num_frames(n)
    wav = rng.randn(b, n).astype(np.float32)
    seq = np.array([t, t - 5], np.int32)
    fe = ofe.LogMelExtractor().double().eval()
    f = fe.number_of_filters
    win, tw, start, length, off, w = _logmel_tables(fe.fbanks.numpy())
    mean = (rng.randn(f) * .1 - 3).astype(np.float32)
    inv_std = (rng.rand(f) * .2 + .4).astype(np.float32)
    pts = None
    if warped:                                   # per-clip filter edges: monotone fractional bin positions
        base = np.linspace(2., 500., f + 2)
        pts = np.stack([base * s for s in (1., .93)]).astype(np.float32)
    outs = []
    for lib in lm_libs:
        out = np.full((b, 1, f, t), np.nan, np.float32)
        stats = np.zeros((32, f, 2), np.float64)
        rc = lib.pbsed_logmel_fwd(P(wav), b, n, t, P(seq), P(win), P(tw), P(start), P(length), P(off), P(w), len(w), f, P(mean), P(inv_std),
                                  C.c_float(1e-18), C.c_float(6.), P(out), P(stats), 320, P(pts), None)
        assert rc == 0
        outs.append((out, stats))
    (out, stats), (out2, stats2) = outs
    fe.mean.copy_(torch.from_numpy(mean).double())
    fe.inv_std.copy_(torch.from_numpy(inv_std).double())
    ref = fe(ofe.stft(torch.from_numpy(wav).double()), seq_len=torch.from_numpy(seq.astype(np.int64)),
             mel_points=None if pts is None else pts.astype(np.float64))[0].numpy()
    assert not np.isnan(out).any()
    assert np.abs(out - ref).max() < 2e-4                         # the bound of tests/test_gpu_ops.py::test_logmel_vs_oracle
    got = stats.sum(0)
    assert np.abs(got[:, 0] - ref.sum((0, 1, 3))).max() < 1e-2
    assert np.array_equal(_bits(out2), _bits(out))
    assert np.array_equal(_bits(stats2), _bits(stats))


# ------------------------------------------------------------------------------------------------ conv_bf16 (configs[2]; the 1x1 layers of 'deep')
@pytest.fixture(scope='module')
def b16_libs(built):
    return built('emu_conv_bf16.cpp', 'tree'), built('emu_conv_bf16.cpp', 'patched')


def _pack_b16(lib, w, dgrad, nsplit):
    cout, cin, kh, kw = w.shape
    inp, outp = C.c_int(), C.c_int()
    lib.pbsed_conv_pack_dims_bf16(cin, cout, dgrad, C.byref(inp), C.byref(outp))
    up = np.zeros(nsplit * kh * kw * outp.value * inp.value, np.uint16)
    assert lib.pbsed_pack_conv_weights_bf16(P(w), P(up), cout, cin, kh, kw, dgrad, nsplit, None) == 0
    return up


def _conv_f64(x, w):
    kh, kw = w.shape[2:]
    b, cin, f, t = x.shape
    xp = np.pad(x, ((0, 0), (0, 0), (kh // 2, kh // 2), (kw // 2, kw // 2)))
    y = np.zeros((b, w.shape[0], f, t))
    for i in range(kh):
        for j in range(kw):
            y += np.einsum('oc,bcft->boft', w[:, :, i, j], xp[:, :, i:i + f, j:j + t])
    return y


B16_FWD = [  # (B, Cin, Cout, F, T, K, pool, nsplit, residual)
    (1, 64, 128, 4, 128, 1, 0, 3, True),      # a 1x1 layer of 'deep' with a residual connection ending at it (bf16x3: fp32-class)
    (2, 64, 128, 4, 100, 1, 1, 3, True),      # ... under a (2,1) pool, ragged
    (1, 64, 64, 4, 64, 3, 0, 1, False),       # configs[2]: 3x3, plain bf16 operands
    (1, 32, 64, 4, 68, 3, 1, 1, False),       # ... with a pool
]


@pytest.mark.parametrize('case', B16_FWD, ids=lambda c: 'x'.join(str(int(v)) for v in c))
def test_conv_bf16_forward_on_the_cpu_tree_vs_float64_and_patched_vs_tree(b16_libs, case):
    b, cin, cout, f, t, k, pool, nsplit, with_res = case
    rng = np.random.RandomState(sum(int(v) for v in case) + 21)
    x = rng.randn(b, cin, f, t).astype(np.float32)
    w = (rng.randn(cout, cin, k, k) * .1).astype(np.float32)
    bias = rng.randn(cout).astype(np.float32)
    scale = (rng.rand(cin) + .5).astype(np.float32)
    shift = (rng.randn(cin) * .1).astype(np.float32)
    seq = np.array([t] + [int(t * .7)] * (b - 1), np.int32)
    fo = f // 2 if pool else f
    res = rng.randn(b, cout, fo, t).astype(np.float32) if with_res else None
    outs = []
    for lib in b16_libs:
        up = _pack_b16(lib, w, 0, nsplit)
        y = np.full((b, cout, fo, t), np.nan, np.float32)
        idx = np.full((b, cout, fo, t), 7, np.uint8) if pool else None
        stats = np.zeros((32, cout, 2), np.float64)
        if with_res:
            rc = lib.pbsed_conv_fwd_bf16_res(P(x), P(up), P(bias), P(scale), P(shift), 1, P(seq), P(y), P(idx), P(stats), 0, b, cin, cout,
                                             f, t, k, k, pool, nsplit, P(res), None)
        else:
            rc = lib.pbsed_conv_fwd_bf16(P(x), P(up), P(bias), P(scale), P(shift), 1, P(seq), P(y), P(idx), P(stats), 0, b, cin, cout,
                                         f, t, k, k, pool, nsplit, None)
        assert rc == 0, lib.emu_last_error()
        outs.append((y, idx, stats))
    (y, idx, stats), (y2, idx2, stats2) = outs
    xa = np.maximum(x.astype(np.float64) * scale[None, :, None, None] + shift[None, :, None, None], 0)
    for i in range(b):
        xa[i, :, :, seq[i]:] = 0
    ref = _conv_f64(xa, w.astype(np.float64)) + bias[None, :, None, None]
    if pool:
        ref = ref.reshape(b, cout, fo, 2, t).max(3)
    if with_res:
        ref = ref + res
    tol = 2e-5 if nsplit == 3 else 3e-2                           # bf16x3 is fp32-class; plain bf16 operands round to 8 bits
    assert not np.isnan(y).any()
    assert np.abs(y - ref).max() < tol * max(1., np.abs(ref).max())
    assert np.array_equal(_bits(y2), _bits(y))
    assert np.array_equal(_bits(stats2), _bits(stats))
    if pool:
        assert np.array_equal(idx2, idx)


B16_BWD = [  # (B, Cin (produced), Cout (contracted), F, T, K, unpool, nsplit)
    (1, 64, 64, 4, 64, 3, False, 1),          # configs[2]'s data gradient through a norm + ReLU
    (2, 64, 32, 8, 100, 3, True, 1),          # ... of a pooled layer, ragged
    (1, 128, 64, 4, 128, 1, False, 3),        # a 1x1 layer of 'deep' (bf16x3)
]


@pytest.mark.parametrize('case', B16_BWD, ids=lambda c: 'x'.join(str(int(v)) for v in c))
def test_conv_bf16_data_gradient_on_the_cpu_tree_vs_float64_and_patched_vs_tree(b16_libs, case):
    b, cin, cout, f, t, k, unpool, nsplit = case
    rng = np.random.RandomState(sum(int(v) for v in case) + 23)
    w = (rng.randn(cout, cin, k, k) * .1).astype(np.float32)
    fg = f // 2 if unpool else f
    g = rng.randn(b, cout, fg, t).astype(np.float32)
    uidx = (rng.rand(b, cout, fg, t) < .5).astype(np.uint8) if unpool else None
    seq = np.array([t] + [int(t * .8)] * (b - 1), np.int32)
    bx = rng.randn(b, cin, f, t).astype(np.float32)
    bmean = (rng.randn(cin) * .1).astype(np.float32)
    binv = (rng.rand(cin) + .5).astype(np.float32)
    bscale = (rng.rand(cin) + .5).astype(np.float32)
    bshift = (rng.randn(cin) * .1).astype(np.float32)
    outs = []
    for lib in b16_libs:
        up = _pack_b16(lib, w, 1, nsplit)
        dz = np.full((b, cin, f, t), np.nan, np.float32)
        stats = np.zeros((32, cin, 2), np.float64)
        rc = lib.pbsed_conv_bwd_data_bf16(P(g), P(up), P(uidx), P(seq), P(dz), P(bx), P(bmean), P(binv), P(bscale), P(bshift), 1, P(stats),
                                          b, cin, cout, f, t, k, k, nsplit, None)
        assert rc == 0, lib.emu_last_error()
        outs.append((dz, stats))
    (dz, stats), (dz2, stats2) = outs
    gu = g.astype(np.float64)
    if unpool:
        full = np.zeros((b, cout, f, t))
        full[:, :, 0::2] = np.where(uidx == 0, gu, 0)
        full[:, :, 1::2] = np.where(uidx == 1, gu, 0)
        gu = full
    ref = _conv_f64(gu, np.flip(w.astype(np.float64), (2, 3)).transpose(1, 0, 2, 3))
    z = bx.astype(np.float64) * bscale[None, :, None, None] + bshift[None, :, None, None]
    keep = z > 0
    for i in range(b):
        keep[i, :, :, seq[i]:] = False
    ref = np.where(keep, ref, 0)
    ok = np.abs(z) > 1e-6
    tol = 2e-5 if nsplit == 3 else 3e-2
    assert not np.isnan(dz).any()
    assert np.abs(dz - ref)[ok].max() < tol * max(1., np.abs(ref).max())
    assert np.array_equal(_bits(dz2), _bits(dz))
    assert np.array_equal(_bits(stats2), _bits(stats))


# ------------------------------------------------------------------------------------------------ conv1d_pc (Conv1d k = 1 / 3, bf16x3, producer / consumer)
@pytest.fixture(scope='module')
def c1_libs(built):
    return built('emu_conv1d_pc.cpp', 'tree'), built('emu_conv1d_pc.cpp', 'patched')


def _pack_c1(lib, w, dgrad):
    cout, cin, kw = w.shape
    inp, outp = C.c_int(), C.c_int()
    lib.pbsed_conv1d_pack_dims_x3(cin, cout, dgrad, C.byref(inp), C.byref(outp))
    up = np.zeros(kw * inp.value * outp.value * 3, np.uint16)
    assert lib.pbsed_pack_conv1d_weights_x3(P(w), P(up), cout, cin, kw, dgrad, None) == 0
    return up


def _conv1d_f64(x, w):
    kw = w.shape[2]
    xp = np.pad(x, ((0, 0), (0, 0), (kw // 2, kw // 2)))
    return sum(np.einsum('oc,bct->bot', w[:, :, j], xp[:, :, j:j + x.shape[2]]) for j in range(kw))


@pytest.mark.parametrize('case', [(2, 64, 128, 200, 3), (2, 96, 256, 132, 1)], ids=lambda c: 'x'.join(str(v) for v in c))
def test_conv1d_pc_on_the_cpu_tree_vs_float64_and_patched_vs_tree(c1_libs, case):
    """Forward (BN-apply + ReLU + mask prologue, bias, statistics) and the data gradient through the layer's norm + ReLU of the
    Conv1d kernels (CNN1d, the output nets): ragged lengths, T no multiple of the 128-wide tile, Cin no multiple of 64."""
    b, cin, cout, t, kw = case
    rng = np.random.RandomState(sum(case) + 31)
    x = rng.randn(b, cin, t).astype(np.float32)
    w = (rng.randn(cout, cin, kw) * .1).astype(np.float32)
    bias = rng.randn(cout).astype(np.float32)
    scale = (rng.rand(cin) + .5).astype(np.float32)
    shift = (rng.randn(cin) * .1).astype(np.float32)
    seq = np.array([t, int(t * .7)], np.int32)
    g = rng.randn(b, cout, t).astype(np.float32)
    bmean = (rng.randn(cin) * .1).astype(np.float32)
    binv = (rng.rand(cin) + .5).astype(np.float32)
    outs = []
    for lib in c1_libs:
        y = np.full((b, cout, t), np.nan, np.float32)
        stats = np.zeros((32, cout, 2), np.float64)
        rc = lib.pbsed_conv1d_fwd_x3(P(x), P(_pack_c1(lib, w, 0)), P(bias), P(scale), P(shift), 1, P(seq), P(y), P(stats), b, cin, cout, t,
                                     kw, None)
        assert rc == 0, lib.emu_last_error()
        dz = np.full((b, cin, t), np.nan, np.float32)
        dstats = np.zeros((32, cin, 2), np.float64)
        rc = lib.pbsed_conv1d_bwd_data_x3(P(g), P(_pack_c1(lib, w, 1)), P(seq), P(dz), P(x), P(bmean), P(binv), P(scale), P(shift), 1,
                                          P(dstats), b, cin, cout, t, kw, None)
        assert rc == 0, lib.emu_last_error()
        outs.append((y, stats, dz, dstats))
    (y, stats, dz, dstats), (y2, stats2, dz2, dstats2) = outs
    z = x.astype(np.float64) * scale[None, :, None] + shift[None, :, None]
    xa = np.maximum(z, 0)
    keep = z > 0
    for i in range(b):
        xa[i, :, seq[i]:] = 0
        keep[i, :, seq[i]:] = False
    ref = _conv1d_f64(xa, w.astype(np.float64)) + bias[None, :, None]
    assert not np.isnan(y).any() and not np.isnan(dz).any()
    assert np.abs(y - ref).max() < 2e-5 * max(1., np.abs(ref).max())
    dref = np.where(keep, _conv1d_f64(g.astype(np.float64), np.flip(w.astype(np.float64), 2).transpose(1, 0, 2)), 0)
    ok = np.abs(z) > 1e-6
    assert np.abs(dz - dref)[ok].max() < 2e-5 * max(1., np.abs(dref).max())
    for a_, b_ in ((y2, y), (stats2, stats), (dz2, dz), (dstats2, dstats)):
        assert np.array_equal(_bits(a_), _bits(b_))


# ------------------------------------------------------------------------------------------------ conv_wgrad (every conv weight-gradient kernel)
@pytest.fixture(scope='module')
def wg_libs(built):
    return built('emu_conv_wgrad.cpp', 'tree'), built('emu_conv_wgrad.cpp', 'patched')


WGRAD = [  # (B, Cin, Cout, F, T, KH, KW, unpool, bf16) -> the kernel conv_wgrad_launch picks
    (1, 64, 64, 4, 64, 3, 3, False, 0),       # conv_wgrad_pc_kernel (bf16x3, producer / consumer, column walk): the largest item of a step
    (2, 16, 16, 8, 100, 3, 3, True, 0),       # conv_wgrad_s16_kernel under a pool, ragged
    (1, 32, 64, 4, 64, 3, 3, False, 0),       # conv_wgrad_wino_kernel (fp32 MFMA, Winograd domain)
    (2, 64, 64, 4, 68, 3, 3, True, 1),        # conv_wgrad_bf16_kernel<3,3,..> (configs[2]; its loader reads seq_len: scalar_tile_loads.patch)
    (2, 64, 64, 1, 200, 1, 3, False, 0),      # conv1d_wgrad_pc_kernel<3> (Conv1d k = 3)
    (2, 96, 64, 1, 132, 1, 1, False, 0),      # conv_wgrad_bf16_kernel<1,1,2,3> (Conv1d k = 1 below 512 inputs, bf16x3)
]


@pytest.mark.parametrize('case', WGRAD, ids=lambda c: 'x'.join(str(int(v)) for v in c))
def test_conv_weight_gradients_on_the_cpu_tree_vs_float64_and_patched_vs_tree(wg_libs, case):
    b, cin, cout, f, t, kh, kw, unpool, bf16 = case
    rng = np.random.RandomState(sum(int(v) for v in case) + 41)
    x = rng.randn(b, cin, f, t).astype(np.float32)
    scale = (rng.rand(cin) + .5).astype(np.float32)
    shift = (rng.randn(cin) * .1).astype(np.float32)
    seq = np.array([t] + [int(t * .7)] * (b - 1), np.int32)
    fg = f // 2 if unpool else f
    g = rng.randn(b, cout, fg, t).astype(np.float32)
    uidx = (rng.rand(b, cout, fg, t) < .5).astype(np.uint8) if unpool else None
    outs = []
    for lib in wg_libs:
        dw = np.zeros((cout, cin, kh, kw), np.float32)
        db = np.zeros(cout, np.float32)
        rc = lib.emu_conv_bwd_weight(P(x), P(scale), P(shift), 1, P(seq), P(g), P(uidx), P(dw), P(db), b, cin, cout, f, t, kh, kw, bf16)
        assert rc == 0, lib.emu_last_error()
        outs.append((dw, db))
    (dw, db), (dw2, db2) = outs
    xa = np.maximum(x.astype(np.float64) * scale[None, :, None, None] + shift[None, :, None, None], 0)
    for i in range(b):
        xa[i, :, :, seq[i]:] = 0
    gu = g.astype(np.float64)
    if unpool:
        full = np.zeros((b, cout, f, t))
        full[:, :, 0::2] = np.where(uidx == 0, gu, 0)
        full[:, :, 1::2] = np.where(uidx == 1, gu, 0)
        gu = full
    xp = np.pad(xa, ((0, 0), (0, 0), (kh // 2, kh // 2), (kw // 2, kw // 2)))
    ref = np.zeros((cout, cin, kh, kw))
    for i in range(kh):
        for j in range(kw):
            ref[:, :, i, j] = np.einsum('boft,bcft->oc', gu, xp[:, :, i:i + f, j:j + t])
    tol = 3e-2 if bf16 else 3e-5
    assert np.abs(dw - ref).max() < tol * max(1., np.abs(ref).max())
    assert np.abs(db - gu.sum((0, 2, 3))).max() < tol * max(1., np.abs(gu.sum((0, 2, 3))).max())
    assert np.array_equal(_bits(dw2), _bits(dw))
    assert np.array_equal(_bits(db2), _bits(db))


def test_wgrad_column_walk_with_xcd_contiguous_positions_gives_the_same_gradient(wg_libs, monkeypatch):
    """PBSED_WGRAD_XCD_COLS=1 (opt-in, unmeasured): the blocks of conv_wgrad_pc_kernel take XCD-contiguous list positions - a
    permutation of who walks which (clip, t range) column.  Same columns, same rows, same products: the gradient is that of the
    default map up to the order of the final atomic adds, and within the float64 bar.  16 blocks over 24 columns (ragged)."""
    lib = wg_libs[0]
    b, cin, cout, f, t = 6, 64, 64, 3, 100            # nTt = 4 -> 24 columns; 64 'CUs' / (1 x 1 tiles) -> split = 16 after the column cap
    rng = np.random.RandomState(77)
    x = rng.randn(b, cin, f, t).astype(np.float32)
    scale = (rng.rand(cin) + .5).astype(np.float32)
    shift = (rng.randn(cin) * .1).astype(np.float32)
    seq = np.array([t, 90, 81, 70, 64, 33], np.int32)
    g = rng.randn(b, cout, f, t).astype(np.float32)
    res = {}
    lib.emu_set_cus(16)
    try:
        for flag in ('0', '1'):
            monkeypatch.setenv('PBSED_WGRAD_XCD_COLS', flag)
            dw = np.zeros((cout, cin, 3, 3), np.float32)
            db = np.zeros(cout, np.float32)
            rc = lib.emu_conv_bwd_weight(P(x), P(scale), P(shift), 1, P(seq), P(g), None, P(dw), P(db), b, cin, cout, f, t, 3, 3, 0)
            assert rc == 0, lib.emu_last_error()
            res[flag] = (dw, db)
    finally:
        lib.emu_set_cus(2)
    xa = np.maximum(x.astype(np.float64) * scale[None, :, None, None] + shift[None, :, None, None], 0)
    for i in range(b):
        xa[i, :, :, seq[i]:] = 0
    xp = np.pad(xa, ((0, 0), (0, 0), (1, 1), (1, 1)))
    ref = np.zeros((cout, cin, 3, 3))
    for i in range(3):
        for j in range(3):
            ref[:, :, i, j] = np.einsum('boft,bcft->oc', g.astype(np.float64), xp[:, :, i:i + f, j:j + t])
    for flag in ('0', '1'):
        assert np.abs(res[flag][0] - ref).max() < 3e-5 * np.abs(ref).max(), flag
    assert np.abs(res['1'][0] - res['0'][0]).max() < 2e-6 * np.abs(ref).max()        # the order of the atomics only
    assert np.abs(res['1'][1] - res['0'][1]).max() < 2e-6 * np.abs(res['0'][1]).max()
    print("bit-identical to the default map:", np.array_equal(res["1"][0], res["0"][0]))


# ------------------------------------------------------------------------------------------------ gru_stack (the persistent scans, blocks concurrent)
@pytest.fixture(scope='module')
def gru_lib(built):
    lib = built('emu_gru_stack.cpp')
    lib.hipemu_set_concurrent(1)                 # every workgroup an OS thread: the rings and projection groups hand words to each other
    return lib


def _table(arrs):
    return (C.c_void_p * len(arrs))(*[P(a) for a in arrs])


def _gru_reference(gi0, w_ih, b_ih, w_hh, b_hh, rev, seq, dy_top):
    """float64 forward + BPTT of nchains unidirectional stacks with the scans' masking (state and output 0 from seq_len on)."""
    nch, nl = len(w_hh), len(w_hh[0])
    t_, b_, h3 = gi0[0].shape
    h_ = h3 // 3
    sig = lambda v: 1 / (1 + np.exp(-v))
    hs = [[None] * nl for _ in range(nch)]
    keep = [[None] * nl for _ in range(nch)]
    for c in range(nch):
        below = None
        for l in range(nl):
            whh, bhh = w_hh[c][l].astype(np.float64), b_hh[c][l].astype(np.float64)
            h = np.zeros((b_, h_))
            out = np.zeros((t_, b_, h_))
            rec = {}
            for s in range(t_):
                t = t_ - 1 - s if rev[c] else s
                gi = gi0[c][t].astype(np.float64) if l == 0 else below[t] @ w_ih[c][l].astype(np.float64).T + b_ih[c][l]
                gh = h @ whh.T + bhh
                r, z = sig(gi[:, :h_] + gh[:, :h_]), sig(gi[:, h_:2 * h_] + gh[:, h_:2 * h_])
                n = np.tanh(gi[:, 2 * h_:] + r * gh[:, 2 * h_:])
                live = (t < seq)[:, None]
                rec[t] = (r, z, n, gh[:, 2 * h_:], h, live)
                h = np.where(live, (1 - z) * n + z * h, 0.)
                out[t] = h
            hs[c][l], keep[c][l], below = out, rec, out
    dgi = [[None] * nl for _ in range(nch)]
    dgh = [[None] * nl for _ in range(nch)]
    for c in range(nch):
        dy = dy_top[c].astype(np.float64)
        for l in reversed(range(nl)):
            whh = w_hh[c][l].astype(np.float64)
            gi_g, gh_g = np.zeros((t_, b_, 3 * h_)), np.zeros((t_, b_, 3 * h_))
            carry = np.zeros((b_, h_))
            for s in reversed(range(t_)):
                t = t_ - 1 - s if rev[c] else s
                r, z, n, ghn, hp, live = keep[c][l][t]
                dh = np.where(live, dy[t] + carry, 0.)
                dn = dh * (1 - z) * (1 - n * n)
                dz = dh * (hp - n) * z * (1 - z)
                dr = dn * ghn * r * (1 - r)
                gi_g[t] = np.concatenate([dr, dz, dn], 1)
                gh_g[t] = np.concatenate([dr, dz, dn * r], 1)
                carry = gh_g[t] @ whh + dh * z
            dgi[c][l], dgh[c][l] = gi_g, gh_g
            if l > 0:
                dy = gi_g @ w_ih[c][l].astype(np.float64)
    return hs, dgi, dgh


@pytest.mark.parametrize('case', [(2, 2, 5, 64, 10), (2, 1, 20, 64, 9)], ids=['two_2layer_stacks_ragged', 'bigru_layer_two_batch_tiles'])
def test_persistent_scans_on_the_cpu_forward_and_bptt_vs_float64(gru_lib, case):
    """The persistent forward scan and BPTT of csrc/gru_stack.hip with every workgroup on its own OS thread (rings + projection
    groups: 40 resp. 32 workgroups of 512 fibers): the tagged-word exchange - publish, paced polls, parity, time-out word - runs for
    real.  Two forward calls on ONE workspace (the parity flips: the first call's words must never satisfy the second call's
    polls), then BPTT from the second call's saved factors, against a float64 GRU.  (The host's memory model is stronger than the
    GPU's: this checks the protocol's logic and the arithmetic; the index maps are enumerated exhaustively by test_scan_protocol.)"""
    nch, nl, b, h, t = case
    rng = np.random.RandomState(sum(case))
    seq = np.sort(rng.randint(max(t - 5, 1), t + 1, b))[::-1].astype(np.int32).copy()
    seq[0] = t
    rev = np.array([c & 1 for c in range(nch)], np.int32)
    k = 1 / np.sqrt(h)
    u = lambda *s: rng.uniform(-k, k, s).astype(np.float32)
    w_ih = [[None if l == 0 else u(3 * h, h) for l in range(nl)] for _ in range(nch)]
    b_ih = [[None if l == 0 else u(3 * h) for l in range(nl)] for _ in range(nch)]
    w_hh = [[u(3 * h, h) for _ in range(nl)] for _ in range(nch)]
    b_hh = [[u(3 * h) for _ in range(nl)] for _ in range(nch)]
    flat = lambda m: [m[c][l] for c in range(nch) for l in range(nl)]
    bp = (b + 15) // 16 * 16
    gran = np.zeros(nch * t * bp * h * (nl + 3 * (nl - 1)), np.uint32)
    err = np.zeros(1, np.uint32)
    for epoch in (1, 2):                         # odd on the first use of a workspace, then alternating
        gi0 = [(rng.randn(t, b, 3 * h) * .5).astype(np.float32) for _ in range(nch)]
        hs = [[np.full((t, b, h), np.nan, np.float32) for _ in range(nl)] for _ in range(nch)]
        save = [[np.full((t, b, 5 * h), np.nan, np.float32) for _ in range(nl)] for _ in range(nch)]
        rc = gru_lib.pbsed_gru_stack_fwd_granule(nch, nl, _table(gi0), _table(flat(w_ih)), _table(flat(b_ih)), _table(flat(w_hh)),
                                                 _table(flat(b_hh)), _table(flat(hs)), _table(flat(save)), P(rev), P(seq), b, h, t,
                                                 P(gran), epoch, P(err), None)
        assert rc == 0 and err[0] == 0, (rc, err[0])
        dy_top = [rng.randn(t, b, h).astype(np.float32) for _ in range(nch)]
        ref_hs, ref_dgi, ref_dgh = _gru_reference(gi0, w_ih, b_ih, w_hh, b_hh, rev, seq, dy_top)
        for c in range(nch):
            for l in range(nl):
                assert np.abs(hs[c][l] - ref_hs[c][l]).max() < 2e-6, (epoch, c, l)
    # BPTT from the second call's saved factors
    w_hh_t = [[np.ascontiguousarray(w_hh[c][l].T) for l in range(nl)] for c in range(nch)]
    w_ih_up_t = [[np.ascontiguousarray(w_ih[c][l + 1].T) if l + 1 < nl else None for l in range(nl)] for c in range(nch)]
    dgi = [[np.full((t, b, 3 * h), np.nan, np.float32) for _ in range(nl)] for _ in range(nch)]
    dgh = [[np.full((t, b, 3 * h), np.nan, np.float32) for _ in range(nl)] for _ in range(nch)]
    gran_b = np.zeros(nch * t * bp * h * (2 * nl - 1), np.uint32)
    rc = gru_lib.pbsed_gru_stack_bwd_granule(nch, nl, _table(flat(w_hh_t)), _table(flat(w_ih_up_t)), _table(flat(hs)), _table(flat(save)),
                                             _table(dy_top), _table(flat(dgi)), _table(flat(dgh)), P(rev), P(seq), b, h, t, P(gran_b), 1,
                                             P(err), None)
    assert rc == 0 and err[0] == 0, (rc, err[0])
    for c in range(nch):
        for l in range(nl):
            scale = max(1., np.abs(ref_dgi[c][l]).max())
            assert np.abs(dgi[c][l] - ref_dgi[c][l]).max() < 1e-5 * scale, (c, l)
            assert np.abs(dgh[c][l] - ref_dgh[c][l]).max() < 1e-5 * scale, (c, l)


def test_xcd_local_exchange_is_gated_by_the_placement_probe_and_checked_in_the_kernel(gru_lib):
    """The XCD-local BPTT exchange (GruStackArgs::local) is correct only if all workgroups of a ring share an XCD.  The library
    OBSERVES that: a probe launch per device records every workgroup's XCC_ID (here: the shim's placement models,
    hipemu_set_xcc_mode) and the exchange is enabled only for an 8-XCD placement that is a function of block id mod 8; the ring's
    workgroups re-check their own XCC_ID at every launch and set bit 1 of the error word when it is not what the probe saw.
    (i) round-robin placement: local on, no error; (ii) a one-XCD partition: the probe refuses, the scan runs with the
    placement-independent exchange and gives the SAME bits; (iii) a dispatcher whose XCD pointer carries over between launches:
    refused too (the scans' own check needs a map that holds for every launch); (iv) placement changes behind a verified probe:
    error word 2."""
    nch, nl, b, h, t = 2, 2, 5, 64, 6
    rng = np.random.RandomState(11)
    seq = np.array([6, 6, 5, 4, 3], np.int32)
    rev = np.array([0, 1], np.int32)
    k = 1 / np.sqrt(h)
    u = lambda *s: rng.uniform(-k, k, s).astype(np.float32)
    flat = lambda m: [m[c][l] for c in range(nch) for l in range(nl)]
    w_hh_t = [[u(h, 3 * h) for _ in range(nl)] for _ in range(nch)]
    w_ih_up_t = [[u(h, 3 * h) if l + 1 < nl else None for l in range(nl)] for _ in range(nch)]
    hs = [[u(t, b, h) for _ in range(nl)] for _ in range(nch)]
    save = [[u(t, b, 5 * h) for _ in range(nl)] for _ in range(nch)]
    dy_top = [rng.randn(t, b, h).astype(np.float32) for _ in range(nch)]
    bp = 16

    def bptt():
        dgi = [[np.full((t, b, 3 * h), np.nan, np.float32) for _ in range(nl)] for _ in range(nch)]
        dgh = [[np.full((t, b, 3 * h), np.nan, np.float32) for _ in range(nl)] for _ in range(nch)]
        gran = np.zeros(nch * t * bp * h * (2 * nl - 1), np.uint32)
        err = np.zeros(1, np.uint32)
        rc = gru_lib.pbsed_gru_stack_bwd_granule(nch, nl, _table(flat(w_hh_t)), _table(flat(w_ih_up_t)), _table(flat(hs)),
                                                 _table(flat(save)), _table(dy_top), _table(flat(dgi)), _table(flat(dgh)), P(rev), P(seq),
                                                 b, h, t, P(gran), 1, P(err), None)
        assert rc == 0
        return int(err[0]), np.stack(flat(dgi) + flat(dgh))

    try:
        gru_lib.hipemu_set_xcc_mode(0)
        gru_lib.pbsed_gru_set_xcd_local(1)                # allowed, probed anew at the next BPTT scan
        e0, r0 = bptt()
        assert e0 == 0 and np.isfinite(r0).all()
        gru_lib.hipemu_set_xcc_mode(1)                    # every workgroup reports XCD 0
        gru_lib.pbsed_gru_set_xcd_local(1)
        e1, r1 = bptt()
        assert e1 == 0 and np.array_equal(r0, r1)         # probe refused: sc1 exchange, the same truncated states
        gru_lib.hipemu_set_xcc_mode(3)                    # round-robin whose start carries over from launch to launch: every launch
        gru_lib.pbsed_gru_set_xcd_local(1)                # is periodic, but the map the scans check themselves against would rotate
        e4, r4 = bptt()
        assert e4 == 0 and np.array_equal(r0, r4)         # refused by the probe's odd-sized middle launch: no false alarm later
        gru_lib.hipemu_set_xcc_mode(0)
        gru_lib.pbsed_gru_set_xcd_local(1)
        assert bptt()[0] == 0                             # verified under the round-robin placement ...
        gru_lib.hipemu_set_xcc_mode(2)                    # ... which then changes behind the library's back
        e2, _ = bptt()
        assert e2 & 2, e2
        assert gru_lib.pbsed_gru_set_xcd_local(0) == 1    # what ops.gru_flags_raise does on seeing bit 1
        e3, r3 = bptt()
        assert e3 == 0 and np.array_equal(r0, r3)
    finally:
        gru_lib.hipemu_set_xcc_mode(0)
        gru_lib.pbsed_gru_set_xcd_local(1)


# ------------------------------------------------------------------------------------------------ tm_gemm / gru_wgrad (the products around the scans)
@pytest.fixture(scope='module')
def rg_lib(built):
    return built('emu_rnn_gemms.cpp')


@pytest.mark.parametrize('bf16', [0, 1], ids=['bf16x3', 'bf16'])
def test_time_major_projection_on_the_cpu_vs_float64(rg_lib, bf16):
    """pbsed_tm_gemm: y [R, N] = bias + sum_i x_i [R, k_i] w_i [N, k_i]^T - the input projection of a BiGRU layer from both directions
    of the layer below (two sources, nothing concatenated), R no multiple of the 128-row block, N no multiple of 128."""
    rng = np.random.RandomState(5 + bf16)
    r, n, ks = 200, 192, [64, 96]
    xs = [rng.randn(r, k).astype(np.float32) for k in ks]
    ws = [(rng.randn(n, k) * .1).astype(np.float32) for k in ks]
    bias = rng.randn(n).astype(np.float32)
    y = np.full((r, n), np.nan, np.float32)
    rc = rg_lib.pbsed_tm_gemm(2, _table(xs), _table(ws), P(np.array(ks, np.int32)), P(bias), P(y), r, n, bf16, None)
    assert rc == 0, rg_lib.emu_last_error()
    ref = bias[None] + sum(x.astype(np.float64) @ w.astype(np.float64).T for x, w in zip(xs, ws))
    assert np.abs(y - ref).max() < (3e-2 if bf16 else 2e-5) * max(1., np.abs(ref).max())


def test_gru_weight_gradients_on_the_cpu_vs_float64(rg_lib):
    """pbsed_gru_wgrad_multi: dW_hh (x = the layer's own states one step back: shift -1 forward chain, +1 reversed chain, rows outside
    [0, T) zero), dW_ih (shift 0, another input width in the same launch), db - straight from time-major buffers."""
    rng = np.random.RandomState(9)
    t, b, h, kin = 12, 5, 64, 96
    g3 = 3 * h
    dg = [(rng.randn(t, b, g3) * .3).astype(np.float32) for _ in range(3)]
    xs = [rng.randn(t, b, h).astype(np.float32), rng.randn(t, b, h).astype(np.float32), rng.randn(t, b, kin).astype(np.float32)]
    shift = np.array([-1, 1, 0], np.int32)
    ks = np.array([h, h, kin], np.int32)
    dws = [np.zeros((g3, k), np.float32) for k in ks]
    dbs = [np.zeros(g3, np.float32) for _ in ks]
    rc = rg_lib.pbsed_gru_wgrad_multi(3, _table(dg), _table(xs), P(shift), _table(dws), _table(dbs), t, b, g3, P(ks), 0, None)
    assert rc == 0, rg_lib.emu_last_error()
    for i in range(3):
        xsft = np.zeros_like(xs[i], dtype=np.float64)
        for tt in range(t):
            if 0 <= tt + shift[i] < t:
                xsft[tt] = xs[i][tt + shift[i]]
        ref = np.einsum('tbg,tbk->gk', dg[i].astype(np.float64), xsft)
        assert np.abs(dws[i] - ref).max() < 3e-5 * max(1., np.abs(ref).max()), i
        assert np.abs(dbs[i] - dg[i].astype(np.float64).sum((0, 1))).max() < 3e-5 * max(1., np.abs(dg[i].sum((0, 1))).max()), i


# ------------------------------------------------------------------------------------------------------------------------------
# misc.hip (losses, optimiser) and postproc.hip (filters, event extraction): the reference-pinned golden vectors of the GPU tests
# (tests/test_gpu_ops.py, tests/test_gpu_postproc.py), through the same kernels, on the CPU.  Reference: pb_sed/models/weak_label/
# crnn.py:214-300 (fwd_bwd loss), pb_sed/models/strong_label/crnn.py:94-112, pb_sed/models/base/inference.py:229-283 (filters).
@pytest.fixture(scope='module')
def misc_lib(built):
    return built('emu_misc.cpp')


@pytest.fixture(scope='module')
def pp_lib(built):
    return built('emu_postproc.cpp')


F = C.c_float


@pytest.mark.parametrize('name', ['ragged_strong', 'full_len', 'no_bwd', 'slat', 'weak_only', 'half_weight_smooth', 'class_weights'])
def test_fbcrnn_loss_kernel_on_the_cpu_vs_reference_golden(misc_lib, golden, name):
    import ast
    g = golden('ref_fbcrnn_loss.npz')
    kw = ast.literal_eval(str(g[f'{name}/kw']))
    f32 = lambda a: np.ascontiguousarray(a, dtype=np.float32)
    yf = f32(g[f'{name}/y_fwd'])
    yb = f32(g[f'{name}/y_bwd']) if f'{name}/y_bwd' in g else None
    weak, bnd = f32(g[f'{name}/weak_targets']), f32(g[f'{name}/boundary_targets'])
    cw = f32(kw['class_weights']) if 'class_weights' in kw else None
    seq = np.ascontiguousarray(g[f'{name}/seq_len'], dtype=np.int32)
    b, k, t = yf.shape
    o_f, o_b, d_f, d_b = (np.full_like(yf, np.nan) for _ in range(4))
    loss = np.full(1, np.nan, np.float32)
    summary = np.full(3 * b * k + 1, np.nan, np.float32)
    rc = misc_lib.pbsed_fbcrnn_loss(P(yf), P(yb), P(weak), P(bnd), P(cw), P(seq), P(o_f), P(o_b if yb is not None else None),
                                    P(d_f), P(d_b if yb is not None else None), P(loss), b, k, t, F(1e-5),
                                    F(kw.get('strong_fwd_bwd_loss_weight', 1.)), int(kw.get('slat', False)),
                                    F(kw.get('label_smoothing', 0.)), 1, P(summary), None)
    assert rc == 0
    assert loss[0] == pytest.approx(float(g[f'{name}/loss']), rel=2e-5)
    np.testing.assert_allclose(d_f, g[f'{name}/grad_y_fwd'], atol=1e-6, rtol=2e-4)
    if yb is not None:
        np.testing.assert_allclose(d_b, g[f'{name}/grad_y_bwd'], atol=1e-6, rtol=2e-4)
    assert np.isfinite(summary).all()


@pytest.mark.parametrize('name', ['a', 'b'])
def test_bicrnn_loss_kernel_on_the_cpu_vs_reference_golden(misc_lib, golden, name):
    g = golden('ref_bicrnn_loss.npz')
    y = np.ascontiguousarray(g[f'{name}/y'], dtype=np.float32)
    tg = np.ascontiguousarray(g[f'{name}/strong_targets'], dtype=np.float32)
    seq = np.ascontiguousarray(g[f'{name}/seq_len'], dtype=np.int32)
    b, k, t = y.shape
    o, d = np.full_like(y, np.nan), np.full_like(y, np.nan)
    loss, scratch = np.full(1, np.nan, np.float32), np.zeros(1, np.float64)
    assert misc_lib.pbsed_bicrnn_loss(P(y), P(tg), P(seq), P(o), P(d), P(loss), P(scratch), b, k, t, 1, None) == 0
    assert loss[0] == pytest.approx(float(g[f'{name}/loss']), rel=2e-5)
    np.testing.assert_allclose(d, g[f'{name}/grad_y'], atol=1e-7, rtol=2e-4)


def test_adam_and_the_gradient_norm_on_the_cpu_vs_float64(misc_lib):
    """padertorch's Adam(lr, gradient_clipping) (pb_sed/experiments/weak_label_crnn/training.py:264-269) restated in float64:
    clip by the global norm, then torch.optim.Adam's update; three steps, an odd length (tail lanes), a skip flag at the end."""
    rng = np.random.default_rng(3)
    n = 10007
    p = rng.standard_normal(n).astype(np.float32)
    m, v = np.zeros(n, np.float32), np.zeros(n, np.float32)
    pr, mr, vr = p.astype(np.float64), np.zeros(n), np.zeros(n)
    lr, b1, b2, eps, max_norm = 5e-4, .9, .999, 1e-8, 5.
    ss, norm = np.zeros(1, np.float64), np.zeros(1, np.float32)
    for step, s in enumerate((1., .1, 3.), 1):
        g = (rng.standard_normal(n) * s).astype(np.float32)
        ref_norm = np.sqrt((g.astype(np.float64) ** 2).sum())
        gc = g.astype(np.float64) * min(1., max_norm / (ref_norm + 1e-6))
        mr = b1 * mr + (1 - b1) * gc
        vr = b2 * vr + (1 - b2) * gc * gc
        pr = pr - lr / (1 - b1 ** step) * mr / (np.sqrt(vr / (1 - b2 ** step)) + eps)
        ss[0] = 0.
        assert misc_lib.pbsed_grad_sumsq(P(g), C.c_size_t(n), P(ss), None) == 0
        assert ss[0] == pytest.approx(ref_norm ** 2, rel=1e-12)
        assert misc_lib.pbsed_adam_step(P(p), P(g), P(m), P(v), C.c_size_t(n), F(lr), F(b1), F(b2), F(eps), step, F(1.), F(max_norm),
                                        P(ss), P(norm), None, 0, None) == 0
        assert norm[0] == pytest.approx(ref_norm, rel=1e-6)
    np.testing.assert_allclose(p, pr, atol=1e-6, rtol=1e-5)
    before = p.copy(), m.copy(), v.copy()
    flags = np.array([0, 3], np.int32)                                   # a scan of this step raised its error word
    assert misc_lib.pbsed_adam_step(P(p), P(g), P(m), P(v), C.c_size_t(n), F(lr), F(b1), F(b2), F(eps), 4, F(1.), F(max_norm),
                                    P(ss), P(norm), P(flags), 2, None) == 0
    for a, bfr in zip((p, m, v), before):
        np.testing.assert_array_equal(a, bfr)


def _rows_n(x, per_row):
    return np.ascontiguousarray(np.broadcast_to(np.asarray(per_row), x.shape[:-1]).reshape(-1), dtype=np.int32)


def _medfilt(lib, x, n):
    x = np.ascontiguousarray(x, dtype=np.float32)
    out = np.full_like(x, np.nan)
    nr = _rows_n(x, n)
    assert lib.pbsed_medfilt(P(x), P(out), P(nr), nr.size, x.shape[-1], None) == 0
    return out


def test_median_filters_on_the_cpu_bit_exact_vs_reference_golden(pp_lib, golden):
    """scipy.signal.medfilt as the reference calls it (pb_sed/models/base/inference.py:229-251), incl. per-row lengths and the
    long filters of the tuning range (bisection-select kernel)."""
    g = golden('ref_filters.npz')
    for n in (1, 3, 5, 11, 41, 101):
        np.testing.assert_array_equal(_medfilt(pp_lib, g['x'], n), g[f'medfilt_{n}'])
    np.testing.assert_array_equal(_medfilt(pp_lib, g['x'], g['len_1d']), g['filtering_med_1d'])
    for n in (5, 151, 301):
        np.testing.assert_array_equal(_medfilt(pp_lib, g['x_long'], n), g[f'medfilt_long_{n}'])


def test_boundaries_filter_on_the_cpu_vs_reference_golden(pp_lib, golden):
    g = golden('ref_filters.npz')
    x = np.ascontiguousarray(g['x'], dtype=np.float32)

    def run(n, f64):
        out, out64 = np.full_like(x, np.nan), np.full(x.shape, np.nan)
        nr = _rows_n(x, n)
        assert pp_lib.pbsed_boundariesfilt(P(x), P(out), P(out64 if f64 else None), P(nr), nr.size, x.shape[-1], None) == 0
        return out64 if f64 else out
    np.testing.assert_array_equal(run(0, False), g['boundariesfilt_0'])
    for n in (2, 6, 20):
        np.testing.assert_allclose(run(n, True), g[f'boundariesfilt_{n}'], rtol=0, atol=1e-15)
    np.testing.assert_array_equal(run(g['steplen_1d'], False), g['filtering_bnd_1d'])


def test_event_frames_on_the_cpu_vs_a_plain_scan(pp_lib):
    """Onsets / offsets of (score > threshold) per class row (pb_sed/models/base/inference.py scores_to_event_list via
    sed_scores_eval's thresholding): events running to the row's end, empty rows, single frames, rows shorter than T."""
    rng = np.random.default_rng(5)
    r, t = 9, 70
    x = rng.random((r, t)).astype(np.float32)
    x[6] = 1.
    x[7] = 0.
    thr = np.array([.5, .3, .9, .1, .7, .5, .5, .5, .99], np.float32)
    ln = np.array([70, 37, 1, 64, 70, 2, 20, 20, 70], np.int32)
    mx = t // 2 + 1
    ev, cnt = np.zeros((r, mx, 2), np.int32), np.zeros(r, np.int32)
    assert pp_lib.pbsed_event_frames(P(x), P(thr), P(ln), P(ev), P(cnt), r, t, mx, None) == 0
    for i in range(r):
        act = np.concatenate([[False], x[i, :ln[i]] > thr[i], [False]])
        on, off = np.flatnonzero(act[1:] & ~act[:-1]), np.flatnonzero(~act[1:] & act[:-1])
        assert cnt[i] == len(on), i
        np.testing.assert_array_equal(ev[i, :len(on)], np.stack([on, off], 1).reshape(-1, 2))


def test_address_sanitizer_sees_a_kernel_leave_its_tensor(tmp_path):
    """tools/emu_asan.sh: the emulated units compiled with -fsanitize=address are a memory checker for the DEVICE code (plain global
    accesses; raw-buffer accesses are range-checked by the shim).  Here: the detector detects - a median-filter launch whose output
    tensor is 64 floats short is reported at the kernel's own store (postproc.hip) - and the same launch on a full tensor is clean."""
    import glob
    import sys
    rt = glob.glob('/opt/rocm/lib/llvm/lib/clang/*/lib/linux/libclang_rt.asan-x86_64.so')
    if not rt:
        pytest.skip('no libclang_rt.asan in this image')
    so = str(tmp_path / 'pp_asan.so')
    subprocess.run([CLANG, '-x', 'c++', '-std=c++20', '-O0', '-g', '-fsanitize=address', '-fno-omit-frame-pointer', '-fPIC', '-shared', '-w',
                    '-I', os.path.join(EMU, 'shim'), '-I', os.path.join(ROOT, 'pb_sed_amd', 'csrc'), os.path.join(EMU, 'emu_postproc.cpp'),
                    os.path.join(EMU, 'hipemu_runtime.cpp'), '-o', so], check=True)
    drive = ("import ctypes as C, numpy as np, sys\n"
             f"lib = C.CDLL({so!r}); r, t = 64, 500\n"
             "x = np.random.rand(r, t).astype(np.float32); out = np.zeros(r * t - int(sys.argv[1]), np.float32); n = np.full(r, 5, np.int32)\n"
             "P = lambda a: a.ctypes.data_as(C.c_void_p)\n"
             "print('rc', lib.pbsed_medfilt(P(x), P(out), P(n), r, t, None))\n")
    env = dict(os.environ, LD_PRELOAD=rt[0], ASAN_OPTIONS='detect_leaks=0')
    clean = subprocess.run([sys.executable, '-c', drive, '0'], env=env, capture_output=True, text=True)
    assert clean.returncode == 0 and 'rc 0' in clean.stdout and 'AddressSanitizer' not in clean.stderr, clean.stderr[-2000:]
    short = subprocess.run([sys.executable, '-c', drive, '64'], env=env, capture_output=True, text=True)
    assert short.returncode != 0 and 'heap-buffer-overflow' in short.stderr and 'medfilt_kernel' in short.stderr, short.stderr[-2000:]
    assert re.search(r'postproc\.hip:\d+', short.stderr)


def test_other_fiber_schedules_expose_a_missing_barrier(tmp_path):
    """tools/emu_schedules.sh: HIPEMU_SCHEDULE = 1 (descending thread order) / >= 2 (seeded shuffles) are other legal executions of
    a block.  The detector detects: a purpose-written kernel in which thread t reads the LDS word thread t - 1 wrote, WITHOUT a
    barrier, happens to be right in ascending order and is wrong in the others; with the barrier every order agrees."""
    src = tmp_path / 'race.cpp'
    src.write_text('''
#include <hip/hip_runtime.h>
static thread_local int lds[256];
template <bool BARRIER> static void k(int* out) {
    const int t = threadIdx.x;
    lds[t] = 1000 + t;
    if (BARRIER) __syncthreads();
    out[t] = t ? lds[t - 1] : -1;
    __syncthreads();
    lds[t] = 0;
}
extern "C" void run(int* out, int barrier) {
    if (barrier) hipLaunchKernelGGL(k<true>, dim3(1), dim3(256), 0, nullptr, out);
    else hipLaunchKernelGGL(k<false>, dim3(1), dim3(256), 0, nullptr, out);
}
''')
    so = str(tmp_path / 'race.so')
    subprocess.run([CLANG, '-x', 'c++', '-std=c++20', '-O0', '-fPIC', '-shared', '-w', '-I', os.path.join(EMU, 'shim'), str(src),
                    os.path.join(EMU, 'hipemu_runtime.cpp'), '-o', so], check=True)
    lib = C.CDLL(so)
    want = np.concatenate([[-1], 1000 + np.arange(255)]).astype(np.int32)
    got = {}
    try:
        for sched in ('0', '1', '2', '3'):
            os.environ['HIPEMU_SCHEDULE'] = sched
            for barrier in (0, 1):
                out = np.zeros(256, np.int32)
                lib.run(P(out), barrier)
                got[sched, barrier] = np.array_equal(out, want)
    finally:
        del os.environ['HIPEMU_SCHEDULE']
    assert all(got[s, 1] for s in '0123'), got
    assert got['0', 0] and not got['1', 0] and not got['2', 0] and not got['3', 0], got
