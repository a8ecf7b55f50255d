"""Thin functional wrappers over the C-ABI (one call = one or a few HIP launches, no autograd).

Tensors are torch CUDA(HIP) tensors used purely as device memory; everything runs on torch's
current stream.  Shapes follow the reference layouts: 2-D activations ``[B, C, F, T]``, 1-D
``[B, C, T]`` (treated as F = 1), GRU scan buffers time-major ``[T, B, *]``.
"""
import ctypes as C
import os
import struct
import weakref

import numpy as np
import torch

from . import _lib
from ._lib import call, ptr, stream


STAT_SLOTS = 32     # PBSED_STAT_SLOTS in csrc/common.h / include/pbsed.h


def _dims4(x):
    if x.dim() == 3:
        b, c, t = x.shape
        return b, c, 1, t
    return tuple(x.shape)


def _conv_tag(b, cin, pc, f, t):
    return f'{cin}->{pc.cout} k{pc.kh}x{pc.kw} B{b} F{f} T{t}'


def _conv_flops(b, cin, pc, f, t):
    return 2 * b * pc.cout * cin * pc.kh * pc.kw * f * t


NSPLIT = {'bf16': 1, 'bf16x3': 3}      # bf16-MFMA operand formats (csrc/conv_bf16.hip)
PACK_EPOCH = [0]      # bump (invalidate_packed) whenever parameters are changed behind torch's back (fused Adam)


def invalidate_packed():
    PACK_EPOCH[0] += 1


_PACKS = {}          # (src data_ptr, mode) -> _PackEntry: every fp32 packed copy made so far (for refresh_packs)
_PACK_DESC = [None, None]     # (tuple of registry keys, device descriptor array)


class _PackEntry:
    def __init__(self, owner, src_ptr, dst, dims, mode, cache_key):
        self.owner, self.src_ptr, self.dst, self.dims, self.mode, self.cache_key = weakref.ref(owner), src_ptr, dst, dims, mode, cache_key


def _register_pack(owner, weight, dst, dims, mode, cache_key):
    if weight.is_contiguous():
        _PACKS[(weight.data_ptr(), mode)] = _PackEntry(owner, weight.data_ptr(), dst, dims, mode, cache_key)


def refresh_packs():
    """Re-pack every registered fp32 / Winograd weight copy in ONE launch and mark the per-parameter caches current
    (called by Trainer.step after the fused Adam changed the parameters in place): ~40 pack launches -> 1."""
    dead = [k for k, e in _PACKS.items() if e.owner() is None or e.owner().data_ptr() != e.src_ptr]
    for k in dead:
        del _PACKS[k]
    if not _PACKS:
        return
    keys = tuple(_PACKS)
    if _PACK_DESC[0] != keys:
        raw = b''.join(struct.pack('<QQiiiiiiii', e.src_ptr, e.dst.data_ptr(), *e.dims, e.mode, 0) for e in _PACKS.values())
        dev = next(iter(_PACKS.values())).dst.device
        _PACK_DESC[0], _PACK_DESC[1] = keys, torch.frombuffer(bytearray(raw), dtype=torch.uint8).to(dev)
    call('pbsed_pack_conv_weights_batched', ptr(_PACK_DESC[1]), len(keys), stream())
    for e in _PACKS.values():
        owner = e.owner()
        key = (owner._version, PACK_EPOCH[0], owner.data_ptr())
        cache = getattr(owner, '_pbsed_pack', None)
        if cache is None or cache.get('key') != key:
            cache = {'key': key}
            owner._pbsed_pack = cache
        cache[e.cache_key] = e.dst


class PackedConv:
    """Weights of one conv layer packed for the forward and data-gradient kernels (cached per parameter
    version, so inference loops pack each layer once)."""

    def __init__(self, weight, owner=None):
        """``owner``: the Parameter a (reshaped) ``weight`` view belongs to; the packed copies are cached ON that
        object (so they die with it - device addresses are recycled) keyed by its version and PACK_EPOCH."""
        self.owner = weight if owner is None else owner
        w = weight.detach()
        if w.dim() == 3:
            cout, cin, kw = w.shape
            kh = 1
        else:
            cout, cin, kh, kw = w.shape
        self.cout, self.cin, self.kh, self.kw = cout, cin, kh, kw
        self.weight = weight

    def _pack(self, dgrad):
        key = (self.owner._version, PACK_EPOCH[0], self.owner.data_ptr(), dgrad)
        cache = getattr(self.owner, '_pbsed_pack', None)
        if cache is None or cache.get('key') != key[:3]:
            cache = {'key': key[:3]}
            try:
                self.owner._pbsed_pack = cache
            except AttributeError:
                pass
        if dgrad in cache:
            return cache[dgrad]
        inp, outp = C.c_int(), C.c_int()
        _lib.lib().pbsed_conv_pack_dims(self.kh, self.kw, self.cin, self.cout, dgrad,
                                        C.byref(inp), C.byref(outp))
        wp = torch.empty(self.kh * self.kw * inp.value * outp.value, device=self.weight.device,
                         dtype=torch.float32)
        w = self.weight.detach().contiguous()
        call('pbsed_pack_conv_weights', ptr(w), ptr(wp), self.cout, self.cin, self.kh, self.kw, dgrad, stream())
        cache[dgrad] = wp
        _register_pack(self.owner, self.weight, wp, (self.cout, self.cin, self.kh, self.kw, inp.value, outp.value), dgrad, dgrad)
        return wp

    def _pack_bf16(self, dgrad, nsplit):
        key = (self.owner._version, PACK_EPOCH[0], self.owner.data_ptr())
        cache = getattr(self.owner, '_pbsed_pack', None)
        if cache is None or cache.get('key') != key:
            cache = {'key': key}
            try:
                self.owner._pbsed_pack = cache
            except AttributeError:
                pass
        ck = ('bf16', dgrad, nsplit)
        if ck in cache:
            return cache[ck]
        inp, outp = C.c_int(), C.c_int()
        _lib.lib().pbsed_conv_pack_dims_bf16(self.cin, self.cout, dgrad, C.byref(inp), C.byref(outp))
        wp = torch.empty(nsplit * self.kh * self.kw * inp.value * outp.value, device=self.weight.device,
                         dtype=torch.int16)
        w = self.weight.detach().contiguous()
        call('pbsed_pack_conv_weights_bf16', ptr(w), ptr(wp), self.cout, self.cin, self.kh, self.kw, dgrad, nsplit,
             stream())
        cache[ck] = wp
        # the bf16 copies (one part, or the three parts of the fp32-class format) join the one-launch refresh after every
        # optimiser step
        if nsplit in (1, 3):
            _register_pack(self.owner, self.weight, wp, (self.cout, self.cin, self.kh, self.kw, inp.value, outp.value),
                           (4 if nsplit == 1 else 6) + dgrad, ck)
        return wp

    def _pack_wino(self, dgrad):
        assert self.kh == 3 and self.kw == 3
        key = (self.owner._version, PACK_EPOCH[0], self.owner.data_ptr())
        cache = getattr(self.owner, '_pbsed_pack', None)
        if cache is None or cache.get('key') != key:
            cache = {'key': key}
            try:
                self.owner._pbsed_pack = cache
            except AttributeError:
                pass
        ck = ('wino', dgrad)
        if ck in cache:
            return cache[ck]
        inp, outp = C.c_int(), C.c_int()
        _lib.lib().pbsed_conv_pack_dims_wino(self.cin, self.cout, dgrad, C.byref(inp), C.byref(outp))
        up = torch.empty(18 * inp.value * outp.value, device=self.weight.device, dtype=torch.float32)
        w = self.weight.detach().contiguous()
        call('pbsed_pack_conv_weights_wino', ptr(w), ptr(up), self.cout, self.cin, dgrad, stream())
        cache[ck] = up
        _register_pack(self.owner, self.weight, up, (self.cout, self.cin, 3, 3, inp.value, outp.value), 2 + dgrad, ck)
        return up

    def _pack_winox3(self, dgrad):
        """Fragment-ordered three-part Winograd weights of csrc/conv_winox3.hip (uint16)."""
        assert self.kh == 3 and self.kw == 3
        key = (self.owner._version, PACK_EPOCH[0], self.owner.data_ptr())
        cache = getattr(self.owner, '_pbsed_pack', None)
        if cache is None or cache.get('key') != key:
            cache = {'key': key}
            try:
                self.owner._pbsed_pack = cache
            except AttributeError:
                pass
        ck = ('winox3', dgrad)
        if ck in cache:
            return cache[ck]
        inp, outp = C.c_int(), C.c_int()
        _lib.lib().pbsed_conv_pack_dims_winox3(self.cin, self.cout, dgrad, C.byref(inp), C.byref(outp))
        up = torch.empty(18 * inp.value * outp.value * 3, device=self.weight.device, dtype=torch.int16)
        w = self.weight.detach().contiguous()
        call('pbsed_pack_conv_weights_winox3', ptr(w), ptr(up), self.cout, self.cin, dgrad, stream())
        cache[ck] = up
        _register_pack(self.owner, self.weight, up, (self.cout, self.cin, 3, 3, inp.value, outp.value), 8 + dgrad, ck)
        return up

    def _pack_c1x3(self, dgrad):
        """Fragment-ordered three-part Conv1d weights of csrc/conv1d_pc.hip (uint16)."""
        assert self.kh == 1 and self.kw in (1, 3)
        key = (self.owner._version, PACK_EPOCH[0], self.owner.data_ptr())
        cache = getattr(self.owner, '_pbsed_pack', None)
        if cache is None or cache.get('key') != key:
            cache = {'key': key}
            try:
                self.owner._pbsed_pack = cache
            except AttributeError:
                pass
        ck = ('c1x3', dgrad)
        if ck in cache:
            return cache[ck]
        inp, outp = C.c_int(), C.c_int()
        _lib.lib().pbsed_conv1d_pack_dims_x3(self.cin, self.cout, dgrad, C.byref(inp), C.byref(outp))
        up = torch.empty(self.kw * inp.value * outp.value * 3, device=self.weight.device, dtype=torch.int16)
        w = self.weight.detach().contiguous()
        call('pbsed_pack_conv1d_weights_x3', ptr(w), ptr(up), self.cout, self.cin, self.kw, dgrad, stream())
        cache[ck] = up
        _register_pack(self.owner, self.weight, up, (self.cout, self.cin, 1, self.kw, inp.value, outp.value), 10 + dgrad, ck)
        return up

    def _pack_s16(self, dgrad):
        """Fragment-ordered three-part weights of the few-channel 3x3 kernels of csrc/conv_s16.hip (uint16)."""
        assert self.kh == 3 and self.kw == 3
        key = (self.owner._version, PACK_EPOCH[0], self.owner.data_ptr())
        cache = getattr(self.owner, '_pbsed_pack', None)
        if cache is None or cache.get('key') != key:
            cache = {'key': key}
            try:
                self.owner._pbsed_pack = cache
            except AttributeError:
                pass
        ck = ('s16x3', dgrad)
        if ck in cache:
            return cache[ck]
        inp, outp = C.c_int(), C.c_int()
        _lib.lib().pbsed_conv_pack_dims_s16(self.cin, self.cout, dgrad, C.byref(inp), C.byref(outp))
        up = torch.empty(5 * 3 * (outp.value // 16) * 512, device=self.weight.device, dtype=torch.int16)
        w = self.weight.detach().contiguous()
        call('pbsed_pack_conv_weights_s16', ptr(w), ptr(up), self.cout, self.cin, dgrad, stream())
        cache[ck] = up
        _register_pack(self.owner, self.weight, up, (self.cout, self.cin, 3, 3, inp.value, outp.value), 14 + dgrad, ck)
        return up

    def fwd(self, precision='f32'):
        if precision == 's16x3':
            return self._pack_s16(0)
        if precision == 'c1x3':
            return self._pack_c1x3(0)
        if precision == 'winox3':
            return self._pack_winox3(0)
        if precision == 'wino':
            return self._pack_wino(0)
        return self._pack(0) if precision == 'f32' else self._pack_bf16(0, NSPLIT[precision])

    def dgrad(self, precision='f32'):
        if precision == 's16x3':
            return self._pack_s16(1)
        if precision == 'c1x3':
            return self._pack_c1x3(1)
        if precision == 'winox3':
            return self._pack_winox3(1)
        if precision == 'wino':
            return self._pack_wino(1)
        return self._pack(1) if precision == 'f32' else self._pack_bf16(1, NSPLIT[precision])


class _ZeroArena:
    """Zero-initialised float64 scratch handed out in slices that are each used once (the statistics accumulators of
    the conv epilogues): one 32 MB fill every few steps instead of one fill launch per conv layer and pass."""

    def __init__(self):
        self.buf, self.off = None, 0

    def take(self, n, device):
        if self.buf is None or self.buf.device != device or self.off + n > self.buf.numel():
            self.buf = torch.zeros(max(1 << 22, n), dtype=torch.float64, device=device)
            self.off = 0
        out = self.buf[self.off:self.off + n]
        self.off += (n + 15) // 16 * 16
        return out


_STATS_ARENA = _ZeroArena()


def _zero_stats(c, device):
    return _STATS_ARENA.take(STAT_SLOTS * c * 2, torch.device(device)).view(STAT_SLOTS, c, 2)


def conv_fwd(x, pc, wp, bias=None, scale=None, shift=None, relu=True, seq_len=None, pool=False,
             want_stats=False, stats_per_cf=False, precision='f32', residual=None):
    """y = conv(prologue(x)) [+pool] [+residual].  Returns (y, pool_idx|None, stats|None).  ``wp`` must have been packed
    with the same ``precision`` ('f32' | 'bf16' | 'bf16x3').  ``residual`` [like y]: added before the store and the
    statistics (fp32 direct kernels only)."""
    _lib.require_gpu(x)
    b, cin, f, t = _dims4(x)
    assert cin == pc.cin, (cin, pc.cin)
    fo = f // 2 if pool else f
    shape = (b, pc.cout, t) if x.dim() == 3 else (b, pc.cout, fo, t)
    y = torch.empty(shape, device=x.device, dtype=torch.float32)
    idx = torch.empty(shape, device=x.device, dtype=torch.uint8) if pool else None
    stats = None
    if want_stats:
        stats = _zero_stats(pc.cout * fo if stats_per_cf else pc.cout, x.device)
    assert residual is None or precision in ('f32', 'bf16x3', 'bf16'), 'a residual add needs the direct fp32 or bf16-MFMA kernel'
    if precision == 'winox3' and t % 4:               # the bf16x3 kernel needs 16-byte aligned rows: same result from the fp32 form
        precision, wp = 'wino', pc.fwd('wino')
    if precision == 's16x3' and t % 4:                # likewise the few-channel kernel: the direct fp32 kernel takes any T
        precision, wp = 'f32', pc.fwd()
    if precision == 'c1x3':
        assert f == 1 and pc.kh == 1 and not pool, 'the producer / consumer Conv1d kernel takes [B, C, T] tensors'
        call('pbsed_conv1d_fwd_x3', ptr(x), ptr(wp), ptr(bias), ptr(scale), ptr(shift), int(relu), ptr(seq_len), ptr(y),
             ptr(stats), b, cin, pc.cout, t, pc.kw, stream(), tag=_conv_tag(b, cin, pc, f, t) + ' c1x3',
             flops=_conv_flops(b, cin, pc, f, t))
        return y, idx, stats
    if precision in ('wino', 'winox3'):
        call('pbsed_conv_fwd_' + precision, ptr(x), ptr(wp), ptr(bias), ptr(scale), ptr(shift), int(relu), ptr(seq_len),
             ptr(y), ptr(idx), ptr(stats), int(stats_per_cf), b, cin, pc.cout, f, t, int(pool), stream(),
             tag=_conv_tag(b, cin, pc, f, t) + ' ' + precision, flops=_conv_flops(b, cin, pc, f, t))
        return y, idx, stats
    if precision == 's16x3':
        call('pbsed_conv_fwd_s16', ptr(x), ptr(wp), ptr(bias), ptr(scale), ptr(shift), int(relu), ptr(seq_len),
             ptr(y), ptr(idx), ptr(stats), int(stats_per_cf), b, cin, pc.cout, f, t, int(pool), stream(),
             tag=_conv_tag(b, cin, pc, f, t) + ' s16x3', flops=_conv_flops(b, cin, pc, f, t))
        return y, idx, stats
    if precision != 'f32':
        if residual is not None:
            assert residual.shape == y.shape and residual.is_contiguous(), (residual.shape, y.shape)
            call('pbsed_conv_fwd_bf16_res', ptr(x), ptr(wp), ptr(bias), ptr(scale), ptr(shift), int(relu), ptr(seq_len),
                 ptr(y), ptr(idx), ptr(stats), int(stats_per_cf), b, cin, pc.cout, f, t, pc.kh, pc.kw, int(pool),
                 NSPLIT[precision], ptr(residual), stream(), tag=_conv_tag(b, cin, pc, f, t) + ' +res ' + precision,
                 flops=_conv_flops(b, cin, pc, f, t))
            return y, idx, stats
        call('pbsed_conv_fwd_bf16', ptr(x), ptr(wp), ptr(bias), ptr(scale), ptr(shift), int(relu), ptr(seq_len),
             ptr(y), ptr(idx), ptr(stats), int(stats_per_cf), b, cin, pc.cout, f, t, pc.kh, pc.kw, int(pool),
             NSPLIT[precision], stream(), tag=_conv_tag(b, cin, pc, f, t) + ' ' + precision,
             flops=_conv_flops(b, cin, pc, f, t))
        return y, idx, stats
    if residual is not None:
        assert residual.shape == y.shape and residual.is_contiguous(), (residual.shape, y.shape)
        call('pbsed_conv_fwd_res', ptr(x), ptr(wp), ptr(bias), ptr(scale), ptr(shift), int(relu), ptr(seq_len),
             ptr(y), ptr(idx), ptr(stats), int(stats_per_cf), b, cin, pc.cout, f, t, pc.kh, pc.kw,
             int(pool), ptr(residual), stream(), tag=_conv_tag(b, cin, pc, f, t) + ' +res', flops=_conv_flops(b, cin, pc, f, t))
        return y, idx, stats
    call('pbsed_conv_fwd', ptr(x), ptr(wp), ptr(bias), ptr(scale), ptr(shift), int(relu), ptr(seq_len),
         ptr(y), ptr(idx), ptr(stats), int(stats_per_cf), b, cin, pc.cout, f, t, pc.kh, pc.kw,
         int(pool), stream(), tag=_conv_tag(b, cin, pc, f, t), flops=_conv_flops(b, cin, pc, f, t))
    return y, idx, stats


def conv_bwd_data(g, pc, wd, x_shape, unpool_idx=None, seq_len=None, bn=None, relu=True, precision='f32'):
    """Gradient wrt the conv input.  ``bn=(x, mean, invstd, scale, shift)`` additionally pushes it
    back through mask/ReLU/BN-apply (returns dz and the (sum dz, sum dz*xhat) statistics)."""
    b, cin, f, t = _dims4(torch.empty(x_shape, device='meta'))
    dz = torch.empty(x_shape, device=g.device, dtype=torch.float32)
    stats = None
    bx = bmean = binv = bsc = bsh = None
    if bn is not None:
        bx, bmean, binv, bsc, bsh = bn
        stats = _zero_stats(cin, g.device)
    if precision == 'winox3' and t % 4:
        precision, wd = 'wino', pc.dgrad('wino')
    if precision == 's16x3' and t % 4:
        precision, wd = 'f32', pc.dgrad()
    if precision == 'c1x3':
        assert f == 1 and pc.kh == 1 and unpool_idx is None
        call('pbsed_conv1d_bwd_data_x3', ptr(g), ptr(wd), ptr(seq_len), ptr(dz), ptr(bx), ptr(bmean), ptr(binv), ptr(bsc),
             ptr(bsh), int(relu), ptr(stats), b, cin, pc.cout, t, pc.kw, stream(),
             tag=_conv_tag(b, cin, pc, f, t) + ' c1x3', flops=_conv_flops(b, cin, pc, f, t))
        return dz, stats
    if precision in ('wino', 'winox3'):
        call('pbsed_conv_bwd_data_' + precision, ptr(g), ptr(wd), ptr(unpool_idx), ptr(seq_len), ptr(dz), ptr(bx),
             ptr(bmean), ptr(binv), ptr(bsc), ptr(bsh), int(relu), ptr(stats), b, cin, pc.cout, f, t, stream(),
             tag=_conv_tag(b, cin, pc, f, t) + ' ' + precision, flops=_conv_flops(b, cin, pc, f, t))
        return dz, stats
    if precision == 's16x3':
        call('pbsed_conv_bwd_data_s16', ptr(g), ptr(wd), ptr(unpool_idx), ptr(seq_len), ptr(dz), ptr(bx),
             ptr(bmean), ptr(binv), ptr(bsc), ptr(bsh), int(relu), ptr(stats), b, cin, pc.cout, f, t, stream(),
             tag=_conv_tag(b, cin, pc, f, t) + ' s16x3', flops=_conv_flops(b, cin, pc, f, t))
        return dz, stats
    if precision != 'f32':
        call('pbsed_conv_bwd_data_bf16', ptr(g), ptr(wd), ptr(unpool_idx), ptr(seq_len), ptr(dz), ptr(bx),
             ptr(bmean), ptr(binv), ptr(bsc), ptr(bsh), int(relu), ptr(stats), b, cin, pc.cout, f, t,
             pc.kh, pc.kw, NSPLIT[precision], stream(), tag=_conv_tag(b, cin, pc, f, t) + ' ' + precision,
             flops=_conv_flops(b, cin, pc, f, t))
        return dz, stats
    call('pbsed_conv_bwd_data', ptr(g), ptr(wd), ptr(unpool_idx), ptr(seq_len), ptr(dz), ptr(bx),
         ptr(bmean), ptr(binv), ptr(bsc), ptr(bsh), int(relu), ptr(stats), b, cin, pc.cout, f, t,
         pc.kh, pc.kw, stream(), tag=_conv_tag(b, cin, pc, f, t), flops=_conv_flops(b, cin, pc, f, t))
    return dz, stats


_SCRATCH = {}           # (device index, stream handle) -> the partial-sum scratch this side owns and registered with the library


SCRATCH_MAX_STREAMS = 8      # registered (device, stream) scratch buffers kept alive per device (least recently used go first)


def ensure_scratch(device):
    """The weight-gradient kernels put their partial-sum slots into a buffer of the CALLER per (device, stream)
    (include/pbsed.h: pbsed_set_scratch); registered on first use and kept alive here - at most SCRATCH_MAX_STREAMS per
    device: a program that cycles through fresh streams would otherwise pin 160 MB per stream handle for ever and run into
    the library's 64-entry registration table; the least recently used registration is withdrawn (pbsed_set_scratch(NULL))
    and its buffer released.  The stream is the CURRENT stream of ``device``."""
    dev = torch.device(device)
    dev_i = dev.index if dev.index is not None else torch.cuda.current_device()
    with torch.cuda.device(dev_i):
        key = (dev_i, stream())
        buf = _SCRATCH.pop(key, None)
        if buf is None:
            mine = [k for k in _SCRATCH if k[0] == dev_i]
            while len(mine) >= SCRATCH_MAX_STREAMS:
                old = mine.pop(0)                        # dict order = least recently used first (entries are re-inserted on use)
                call('pbsed_set_scratch', None, 0, old[1])
                _SCRATCH.pop(old)
            buf = torch.empty(_lib.lib().pbsed_scratch_bytes(), dtype=torch.uint8, device=f'cuda:{dev_i}')
            call('pbsed_set_scratch', ptr(buf), buf.numel(), key[1])
        _SCRATCH[key] = buf                              # (re-)insert at the most-recently-used end
    return buf


def conv_bwd_weight(x, g, pc, dw, db=None, scale=None, shift=None, relu=True, seq_len=None,
                    unpool_idx=None, precision='f32', bng=None, per_cf=False):
    """dw (+=), db (+=): gradients of the conv parameters (buffers must be pre-zeroed/accumulating).  ``precision``
    'bf16': bf16-MFMA operands, fp32 accumulation (layers with >= 32 input and output channels).
    ``bng``: a LazyBNGrad instead of ``g`` (conv_bwd_weight_bng_supported must hold): the kernel forms dY from (dz, x, coef)
    in its loader and the formed gradient is RETURNED (same shape as dz) for the layer's data gradient."""
    ensure_scratch(x.device)
    b, cin, f, t = _dims4(x)
    if bng is not None:
        coef = bng.coefficients()
        gout = torch.empty_like(bng.dz)
        call('pbsed_conv_bwd_weight_bng', ptr(x), ptr(scale), ptr(shift), int(relu), ptr(seq_len), ptr(bng.dz), ptr(bng.x), ptr(coef),
             int(per_cf), ptr(bng.seq_len), ptr(gout), ptr(unpool_idx), ptr(dw), ptr(db), b, cin, pc.cout, f, t, pc.kh, pc.kw, stream(),
             tag=_conv_tag(b, cin, pc, f, t) + ' x3pc', flops=_conv_flops(b, cin, pc, f, t))
        return gout
    if precision == 'bf16' and cin >= 32 and pc.cout >= 32:
        call('pbsed_conv_bwd_weight_bf16', ptr(x), ptr(scale), ptr(shift), int(relu), ptr(seq_len), ptr(g),
             ptr(unpool_idx), ptr(dw), ptr(db), b, cin, pc.cout, f, t, pc.kh, pc.kw, stream(),
             tag=_conv_tag(b, cin, pc, f, t) + ' bf16', flops=_conv_flops(b, cin, pc, f, t))
        return
    # which kernel the library picks for these shapes (conv_wgrad.hip dispatch), for the bench's tags: the producer / consumer
    # bf16x3 kernel from 64 channels on, the Winograd-F(4,3) fp32 kernel for the other 3x3 layers with >= 64 output channels
    k33 = pc.kh == 3 and pc.kw == 3
    x3pc = k33 and t % 4 == 0 and ((cin >= 64 and pc.cout >= 64) or (cin == 32 and pc.cout == 32))    # conv_wgrad_launch's rule
    s16 = k33 and t % 4 == 0 and 2 <= cin <= 16 and pc.cout in (16, 32)
    wino = k33 and not x3pc and pc.cout >= 64 and cin >= 16
    k11 = pc.kh == 1 and pc.kw == 1 and f > 1                     # 1x1 conv2d: conv1d_wgrad_pc_kernel<1> over rows / conv_wgrad_bf16_kernel<1,1,2,3>
    c1pc = k11 and unpool_idx is None and cin >= 128 and pc.cout >= 128 and t % 4 == 0
    b16x3 = k11 and not c1pc and 32 <= cin < 1024 and pc.cout >= 32
    if f == 1 and pc.kh == 1 and unpool_idx is None:              # Conv1d layers: conv1d_wgrad_pc_kernel / conv_wgrad_bf16_kernel<1,KW,2,3>
        c1pc = cin >= 64 and pc.cout >= 64 and (pc.kw == 3 or (pc.kw == 1 and cin >= 512))
        b16x3 = not c1pc and pc.kw in (1, 3) and 32 <= cin < 1024 and pc.cout >= 32
    call('pbsed_conv_bwd_weight', ptr(x), ptr(scale), ptr(shift), int(relu), ptr(seq_len), ptr(g),
         ptr(unpool_idx), ptr(dw), ptr(db), b, cin, pc.cout, f, t, pc.kh, pc.kw, stream(),
         tag=_conv_tag(b, cin, pc, f, t) + (' x3pc' if x3pc else ' s16x3' if s16 else ' wino' if wino else ' c1x3' if c1pc else ' bf16x3' if b16x3 else ''), flops=_conv_flops(b, cin, pc, f, t))


def pool21_fwd(x):
    """(2,1) max-pool of [B,C,F,T] over frequency-row pairs -> (y [B,C,F/2,T], argmax bytes)."""
    b, c, f, t = x.shape
    assert f % 2 == 0 and x.is_contiguous()
    y = torch.empty((b, c, f // 2, t), device=x.device, dtype=torch.float32)
    idx = torch.empty(y.shape, device=x.device, dtype=torch.uint8)
    call('pbsed_pool21_fwd', ptr(x), ptr(y), ptr(idx), y.numel(), t, stream())
    return y, idx


def pool21_bwd_add(g, idx, dx):
    """dx[argmax row] += g (dx [B,C,F,T], g / idx [B,C,F/2,T])."""
    call('pbsed_pool21_bwd_add', ptr(g.contiguous()), ptr(idx), ptr(dx), g.numel(), g.shape[-1], stream())
    return dx


def add_inplace(a, b):
    assert a.shape == b.shape and a.is_contiguous() and b.is_contiguous()
    call('pbsed_add_inplace', ptr(a), ptr(b), a.numel(), stream())
    return a


class BNState:
    """Per-layer batch statistics / fused scale+shift produced by bn_finalize (or eval params)."""

    def __init__(self, c, device):
        buf = torch.empty((6, c), device=device, dtype=torch.float32)
        self.mean, self.invstd, self.scale, self.shift, self.m1, self.m2 = buf.unbind(0)
        self.c = c


BN_EPOCH = [0]        # bumped whenever running statistics are written behind torch's back (bn_finalize in training mode)


def bn_finalize(stats, count, norm, training_update=True):
    if training_update:
        BN_EPOCH[0] += 1
    st = BNState(stats.shape[1], stats.device)
    call('pbsed_bn_finalize', ptr(stats), float(count), ptr(norm.gamma.detach()), ptr(norm.beta.detach()),
         float(norm.eps), float(norm.momentum), ptr(norm.running_mean if training_update else None),
         ptr(norm.running_power if training_update else None), ptr(st.mean), ptr(st.invstd),
         ptr(st.scale), ptr(st.shift), st.c, stream())
    return st


def channel_stats(x, seq_len):
    """Masked per-channel (sum, sum of squares) of a stack input [B,C,(S,)T] -> stats for bn_finalize."""
    b, c, s, t = _dims4(x)
    stats = _zero_stats(c, x.device)
    call('pbsed_channel_stats', ptr(x), ptr(seq_len), ptr(stats), b, c, s, t, stream())
    return stats


def bn_relu_fwd(x, st, seq_len, relu=True):
    """mask * relu(x * scale + shift): a norm + activation that closes a stack (no consumer conv to fuse into)."""
    b, c, s_, t = _dims4(x)
    y = torch.empty_like(x)
    call('pbsed_bn_relu_fwd', ptr(x), ptr(st.scale), ptr(st.shift), ptr(seq_len), ptr(y), int(relu), b, c, s_, t, stream())
    return y


def bn_relu_bwd(dy, x, st, seq_len, relu=True):
    """Backward of bn_relu_fwd up to the norm's input statistics: returns (dz, partial sums for bn_backward)."""
    b, c, s_, t = _dims4(x)
    dz = torch.empty_like(x)
    stats = _zero_stats(c, x.device)
    call('pbsed_bn_relu_bwd', ptr(dy.contiguous()), ptr(x), ptr(st.scale), ptr(st.shift), ptr(st.mean), ptr(st.invstd), ptr(seq_len),
         ptr(dz), ptr(stats), int(relu), b, c, s_, t, stream())
    return dz, stats


def bn_eval_params(norm):
    """Per-channel (mean, invstd, scale, shift) of a norm layer applied with its running statistics.  Cached on the module
    until a parameter / buffer changes (torch versions for in-place updates, PACK_EPOCH / BN_EPOCH for the library's own
    writes): an inference loop derives them once, not once per batch and layer."""
    tensors = (norm.gamma, norm.beta, norm.running_mean, norm.running_power)
    key = (PACK_EPOCH[0], BN_EPOCH[0]) + tuple((t._version, t.data_ptr()) for t in tensors)
    cached = getattr(norm, '_pbsed_eval', None)
    if cached is not None and cached[0] == key:
        return cached[1]
    st = BNState(norm.gamma.numel(), norm.gamma.device)
    call('pbsed_bn_eval_params', ptr(norm.gamma.detach()), ptr(norm.beta.detach()), float(norm.eps),
         ptr(norm.running_mean), ptr(norm.running_power), ptr(st.mean), ptr(st.invstd), ptr(st.scale),
         ptr(st.shift), st.c, stream())
    try:
        norm._pbsed_eval = (key, st)
    except AttributeError:
        pass
    return st


def bn_backward(dz, x, st, stats, count, dgamma, dbeta, seq_len):
    """In place dz -> dx; accumulates dgamma/dbeta."""
    b, c, s, t = _dims4(x)
    call('pbsed_bn_bwd', ptr(dz), ptr(x), ptr(stats), float(count), ptr(st.mean), ptr(st.invstd), ptr(st.scale),
         ptr(dgamma), ptr(dbeta), ptr(seq_len), b, c, s, t, stream())
    return dz


class LazyBNGrad:
    """The gradient wrt a norm's input, not yet formed: dx = k1[c] dz + k2[c] x + k3[c] inside the sequences.  The weight-gradient
    kernel of the conv that PRODUCED x forms it while it stages dY (conv_bwd_weight(bng=...), which also writes it out for the
    data gradient); ``materialize`` is the stand-alone pass (pbsed_bn_bwd) for consumers without that loader."""

    def __init__(self, dz, x, st, stats, count, dgamma, dbeta, seq_len):
        self.dz, self.x, self.st, self.stats, self.count = dz, x, st, stats, count
        self.dgamma, self.dbeta, self.seq_len = dgamma, dbeta, seq_len
        self.shape = dz.shape

    def coefficients(self):
        """dgamma / dbeta (+=) and the coefficient table [3, C] (pbsed_bn_bwd_coef); once."""
        c = self.st.c
        coef = torch.empty((3, c), device=self.dz.device, dtype=torch.float32)
        call('pbsed_bn_bwd_coef', ptr(self.stats), float(self.count), ptr(self.st.mean), ptr(self.st.invstd), ptr(self.st.scale),
             ptr(self.dgamma), ptr(self.dbeta), ptr(coef), c, stream())
        return coef

    def materialize(self):
        return bn_backward(self.dz, self.x, self.st, self.stats, self.count, self.dgamma, self.dbeta, self.seq_len)


def conv_bwd_weight_bng_supported(pc, cin, f, t, per_cf, precision='f32'):
    """Whether the weight-gradient kernel of this layer has the BN-backward dY loader (pbsed_conv_bwd_weight_bng_supported)."""
    if precision == 'bf16' and cin >= 32 and pc.cout >= 32:
        return False
    return bool(_lib.lib().pbsed_conv_bwd_weight_bng_supported(pc.kh, pc.kw, cin, pc.cout, f, t, int(per_cf)))


def augment_logmel(x, masks, seq_len, noise=None, noise_scale=None, mean=None, inv_std=None, clamp=None):
    """In place on x [B,1,F,T] or [B,F,T]: optional normalisation clamp((x - mean[f]) * inv_std[f]) (second half of a
    statistics-tracking front-end pass), += noise_scale[b]*noise, per-clip time / frequency masks (int32 [B,4] = t_on,
    t_off, f_on, f_off; None: no masks) and the sequence mask."""
    _lib.require_gpu(x)
    b, f, t = x.shape[0], x.shape[-2], x.shape[-1]
    assert x.is_contiguous() and (masks is None or (masks.shape == (b, 4) and masks.dtype == torch.int32))
    call('pbsed_augment_logmel', ptr(x), ptr(noise), ptr(noise_scale), ptr(masks), ptr(seq_len), ptr(mean), ptr(inv_std),
         float(clamp if clamp is not None else 3e38), b, f, t, stream())
    return x


def feature_norm_stats(n_filters, device):
    """Zeroed [PBSED_STAT_SLOTS][F][2] accumulators for the statistics pass of the front-end kernels."""
    return _zero_stats(n_filters, device)


def feature_norm_update(stats, count, fe):
    """Advance the cumulative per-mel statistics of ``fe`` (modules.NormalizedLogMelExtractor) by one batch and refresh
    its ``mean`` / ``inv_std`` buffers (all on the device, no host sync)."""
    call('pbsed_feature_norm_update', ptr(stats), float(count), ptr(fe.running_mean), ptr(fe.running_power),
         ptr(fe.num_tracked_values), float(fe.norm_eps), ptr(fe.mean), ptr(fe.inv_std), fe.mean.numel(), stream())


def bct_to_tbc(x):
    b, c, t = x.shape
    y = torch.empty((t, b, c), device=x.device, dtype=torch.float32)
    call('pbsed_bct_to_tbc', ptr(x), ptr(y), b, c, t, stream())
    return y


def tbc_to_bct(x, shift=0):
    t, b, c = x.shape
    y = torch.empty((b, c, t), device=x.device, dtype=torch.float32)
    call('pbsed_tbc_to_bct', ptr(x), ptr(y), b, c, t, int(shift), stream())
    return y


def transposed(param):
    """W^T of a 2-D parameter, cached ON the parameter per version and refreshed with the packed conv weights in the one
    launch after every optimiser step (refresh_packs, mode 12) instead of one transpose launch per matrix and step."""
    key = (param._version, PACK_EPOCH[0], param.data_ptr())
    cache = getattr(param, '_pbsed_pack', None)
    if cache is None or cache.get('key') != key:
        cache = {'key': key}
        try:
            param._pbsed_pack = cache
        except AttributeError:
            pass
    if 'T' in cache:
        return cache['T']
    w = param.detach()
    y = transpose2d(w)
    cache['T'] = y
    r, c = w.shape
    _register_pack(param, w, y, (r, c, 1, 1, c, r), 12, 'T')
    return y


def transpose2d(x):
    r, c = x.shape
    y = torch.empty((c, r), device=x.device, dtype=torch.float32)
    call('pbsed_transpose2d', ptr(x.contiguous()), ptr(y), r, c, stream())
    return y


# Verification tap (tests/test_gpu_configs.py::test_c3_rnn_launches_in_situ): when a list is assigned, tm_gemm and gru_wgrad
# append their operands and results - what a launch-by-launch check of the recurrent part of a train step needs (the conv
# launches have engine.DECISION_TAP).
LAUNCH_TAP = None


def tm_gemm(xs, ws, bias=None, precision='f32', role='fwd'):
    """Time-major projection: sum_i xs[i] [T,B,K_i] @ ws[i] [N,K_i]^T (+ bias [N]) -> [T,B,N].  'f32': fp32-class products
    (exact bf16x3 operand splits on the bf16 MFMA); 'bf16': plain bf16 operands.  ``role`` ('fwd' | 'bwd') only labels the
    launch for bench.py's forward / backward accounting."""
    t, b = xs[0].shape[:2]
    n = ws[0].shape[0]
    ks = [x.shape[2] for x in xs]
    assert all(x.shape[:2] == (t, b) and x.is_contiguous() for x in xs)
    assert all(w.shape == (n, k) and w.is_contiguous() for w, k in zip(ws, ks))
    y = torch.empty((t, b, n), device=xs[0].device, dtype=torch.float32)
    call('pbsed_tm_gemm', len(xs), _lib.ptr_array(xs), _lib.ptr_array(ws), _lib.int_array(ks), ptr(bias), ptr(y), t * b, n,
         int(precision == 'bf16'), stream(), tag=f'{"+".join(map(str, ks))}->{n} R{t * b}' + (' bf16' if precision == 'bf16' else '') + ' ' + role,
         flops=2. * t * b * n * sum(ks))
    if LAUNCH_TAP is not None:
        LAUNCH_TAP.append(('tm_gemm', list(xs), list(ws), bias, precision, role, y))
    return y


def gru_scan_fwd(gi, w_hh, b_hh, reverse, seq_len, save=True):
    """gi: list of [T,B,3H] per chain.  Returns (hs list [T,B,H], save list [T,B,4,H]|None)."""
    n = len(gi)
    t, b, g = gi[0].shape
    h = g // 3
    dev = gi[0].device
    hs = [torch.empty((t, b, h), device=dev, dtype=torch.float32) for _ in range(n)]
    sv = [torch.empty((t, b, 4, h), device=dev, dtype=torch.float32) for _ in range(n)] if save else None
    call('pbsed_gru_scan_fwd', n, _lib.ptr_array(gi), _lib.ptr_array([w.contiguous() for w in w_hh]),
         _lib.ptr_array(b_hh), _lib.ptr_array(hs), _lib.ptr_array(sv) if save else None,
         _lib.int_array(reverse), ptr(seq_len), b, h, t, stream())
    return hs, sv


def gru_scan_bwd(w_hh_t, hs, save, dy, reverse, seq_len):
    n = len(hs)
    t, b, h = hs[0].shape
    dev = hs[0].device
    dgi = [torch.empty((t, b, 3 * h), device=dev, dtype=torch.float32) for _ in range(n)]
    dgh = [torch.empty((t, b, 3 * h), device=dev, dtype=torch.float32) for _ in range(n)]
    dhz = [torch.empty((t, b, h), device=dev, dtype=torch.float32) for _ in range(n)]
    call('pbsed_gru_scan_bwd', n, _lib.ptr_array(w_hh_t), _lib.ptr_array(hs), _lib.ptr_array(save),
         _lib.ptr_array(dy), _lib.ptr_array(dgi), _lib.ptr_array(dgh), _lib.ptr_array(dhz),
         _lib.int_array(reverse), ptr(seq_len), b, h, t, stream())
    return dgi, dgh


last_scan_blocks = 0    # workgroups of the persistent BPTT scan enqueued last (one per CU for the scan's whole duration)


class launch_cus:
    """``with ops.launch_cus(n):`` - the weight-gradient launches enqueued inside size their persistent grids for ``n`` compute
    units (pbsed_set_launch_cus) instead of the whole device; for launches that run beside a persistent scan on a side stream."""

    def __init__(self, cus):
        self.cus = int(cus)

    def __enter__(self):
        self.old = _lib.lib().pbsed_set_launch_cus(self.cus)
        return self

    def __exit__(self, *exc):
        _lib.lib().pbsed_set_launch_cus(self.old)
        return False


def cus_beside_last_scan(device):
    """Compute units the last enqueued persistent BPTT scan leaves free (at least an eighth of the device: the scans keep within 7/8)."""
    n = max(_granule_capacity(_dev_index(torch.device(device)), 64, True, False, 1), 8)      # = the device's CU count (one scan workgroup per CU)
    return max(n - last_scan_blocks, n // 8) // 8 * 8


_GRANULE_WS = {}        # (device, shape) -> [granule workspace, epoch counter] of the persistent GRU scan
_GRU_FLAGS = {}         # device -> [int32 flag words, words handed out since the last check]
GRU_FLAG_WORDS = 64


def gru_flags(device):
    """The device's error words of the persistent scans ([GRU_FLAG_WORDS] int32; a scan that hits its bounded-spin
    time-out sets its word).  Trainer.step hands them to the fused Adam (which then skips the update) and copies them to
    the host with the step's summary; inference checks them at the end of every batch."""
    key = str(torch.device(device))
    st = _GRU_FLAGS.get(key)
    if st is None:
        st = _GRU_FLAGS[key] = [torch.zeros(GRU_FLAG_WORDS, dtype=torch.int32, device=device), 0]
    return st


def _gru_err_flag(device):
    st = gru_flags(device)
    if st[1] >= GRU_FLAG_WORDS:
        check_gru_sync()                             # nobody looked for a long time: look now (host sync)
    word = st[0][st[1]:st[1] + 1]
    st[1] += 1
    return word


def gru_flags_raise(host_words):
    """``host_words``: a host copy of gru_flags()[0].  Drops the scan workspaces (they hold words of mixed parity after
    a time-out) and raises if any word is set."""
    words = np.asarray(host_words)
    if np.any(words):
        _GRANULE_WS.clear()
        for st in _GRU_FLAGS.values():
            st[0].zero_()
            st[1] = 0
        if np.any(words & 2):
            # a workgroup of an XCD-local ring ran on another XCD than the library's placement probe saw (include/pbsed.h,
            # pbsed_gru_set_xcd_local): that exchange is off from here on, the scans use the placement-independent one
            _lib.lib().pbsed_gru_set_xcd_local(0)
            raise RuntimeError('persistent GRU scan: a workgroup of an XCD-local ring was dispatched to another XCD than the '
                               'placement probe observed (partition mode / CU mask changed?); this step\'s results were '
                               'discarded (the optimiser update was skipped on the device) and the XCD-local exchange is '
                               'now off for this process (PBSED_GRU_XCD_LOCAL=0 does the same from the start).')
        raise RuntimeError('persistent GRU scan: inter-workgroup hand-off timed out; this step\'s results were discarded '
                           '(the optimiser update was skipped on the device).  PBSED_GRU_PERSIST=0 selects the '
                           'launch-per-step scans.')


def check_gru_sync():
    """Raise if any persistent GRU scan since the last check hit its bounded-spin timeout (host sync)."""
    for st in list(_GRU_FLAGS.values()):
        if st[1]:
            st[1] = 0
            gru_flags_raise(st[0].cpu().numpy())


_PINNED = {}            # byte size class (power of two) -> list of [view, event of its last use, held, backing uint8 buffer]
_PINNED_MAX_BYTES = 1 << 30      # page-locked memory the pool may keep; beyond it idle buffers of the largest classes go first


def _pinned_trim():
    total = sum(sl[3].numel() for pool in _PINNED.values() for sl in pool)
    for cls in sorted(_PINNED, reverse=True):
        pool = _PINNED[cls]
        for sl in list(pool):
            if total <= _PINNED_MAX_BYTES:
                return
            if not sl[2] and (sl[1] is None or sl[1].query()):
                pool.remove(sl)
                total -= sl[3].numel()


def pinned_buffer(shape, dtype, hold=False):
    """A pinned host buffer out of a pool keyed by BYTE SIZE CLASS (next power of two, >= 256 B): one that no pending batch
    holds and whose last asynchronous use has completed, else a new one (allocating pinned memory costs a driver call; the
    pool settles at the few buffers the pipeline depth needs and variable-length batches share classes instead of pinning a
    buffer per distinct shape; nothing here waits for the device).  slot[0] is a view of the requested shape / dtype.  The
    caller records an event behind its copy with ``pinned_buffer_used``; ``hold`` keeps the buffer out of circulation until
    its reader clears slot[2]."""
    shape = tuple(int(d) for d in shape)
    nbytes = int(np.prod(shape, dtype=np.int64)) * torch.empty((), dtype=dtype).element_size()
    cls = max(256, 1 << max(nbytes - 1, 0).bit_length())
    pool = _PINNED.setdefault(cls, [])
    slot = next((sl for sl in pool if not sl[2] and (sl[1] is None or sl[1].query())), None)
    if slot is None:
        slot = [None, None, False, torch.empty(cls, dtype=torch.uint8, pin_memory=True)]
        pool.append(slot)
        _pinned_trim()
    slot[0] = slot[3][:nbytes].view(dtype).view(shape)
    slot[2] = hold
    return slot


def pinned_buffer_used(slot):
    slot[1] = torch.cuda.Event()
    slot[1].record()
    return slot[1]


def host_to_device(a, device, dtype=None):
    """Small host array -> new device tensor WITHOUT making the host wait for the stream.  A copy out of pageable memory
    (``torch.as_tensor(a).to(device)``, ``torch.tensor(list, device=...)``) returns only when everything queued before it
    has run (9 ms behind ten 1 ms kernels, tools/micro/h2d_block_probe.py): one such copy per batch puts the host in
    lock-step with the device.  Staged through a pooled pinned buffer instead (host 7 - 18 us)."""
    t = torch.as_tensor(np.ascontiguousarray(a))
    if dtype is not None:
        t = t.to(dtype)
    slot = pinned_buffer(t.shape, t.dtype)
    slot[0].copy_(t)
    out = slot[0].to(device, non_blocking=True)
    pinned_buffer_used(slot)
    return out


def gru_flags_snapshot():
    """Asynchronous host copies (pinned) of the error words of the scans launched since the last look, queued behind
    them on the current stream; ``gru_flags_check`` reads them once the caller knows the stream got that far."""
    snap = []
    for st in _GRU_FLAGS.values():
        if st[1]:
            st[1] = 0
            slot = pinned_buffer((GRU_FLAG_WORDS,), torch.int32, hold=True)
            slot[0].copy_(st[0], non_blocking=True)
            pinned_buffer_used(slot)
            snap.append(slot)
    return snap


def gru_flags_check(snapshot):
    for slot in snapshot:
        slot[1].synchronize()              # the copy's own event: the caller may only have waited for an earlier one
        words = slot[0].numpy().copy()
        slot[2] = False
        gru_flags_raise(words)


_CU_COUNT = {}


class ScanWatch:
    """Slow-down guard of the persistent scans.  They assume one workgroup per CU on <= 7/8 of the device; if something
    else holds CUs while a scan runs (a collective on another stream, a second process on the device) the scan does not
    fail - its hand-offs just get slower.  Every ``every``-th scan call is bracketed by events; ``check()`` (called by
    Trainer.step, no host sync: only completed events are read) compares the newest duration with the running median of
    its kind and warns when it is more than ``factor`` times slower."""

    def __init__(self, every=16, factor=1.6):
        self.every, self.factor = every, factor
        self.calls, self.pending, self.history, self.warned = 0, [], {}, 0

    def bracket(self, kind):
        self.calls += 1
        if self.calls % self.every:
            return None
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        self.pending.append((kind, e0, e1))
        return e1

    def check(self):
        import warnings
        still = []
        for kind, e0, e1 in self.pending:
            if not e1.query():
                still.append((kind, e0, e1))
                continue
            ms = e0.elapsed_time(e1)
            hist = self.history.setdefault(kind, [])
            if len(hist) >= 4:
                med = sorted(hist)[len(hist) // 2]
                if ms > self.factor * med:
                    self.warned += 1
                    warnings.warn(f'persistent GRU scan {kind}: {ms:.2f} ms against a median of {med:.2f} ms - another stream or '
                                  f'process is holding compute units (PBSED_GRU_PERSIST=0 selects the launch-per-step scans)')
            hist.append(ms)
            del hist[:-32]
        self.pending = still


scan_watch = ScanWatch()


def _dev_index(dev):
    dev = dev.index if isinstance(dev, torch.device) else dev
    return torch.cuda.current_device() if dev is None else dev


_POLL_TUNED = {}        # (device, kind, slot, shape) -> chosen delay: a scan shape is measured once per process and device
_POLL_DEFAULT = {}      # (device, kind) -> the library's built-in delays [fwd, fwd_gate, bwd, bwd_gate], read before any change
_POLL_INSTALLED = {}    # (device, kind) -> the delays the library holds right now
POLL_TUNE_MAX_SHAPES = 8        # shapes measured per (device, kind, slot); further shapes take the built-in delay


def _install_poll_delay(dev_i, kind, slot, delay):
    """The library keeps ONE delay set per (device, kind); a tuned value belongs to ONE shape.  So the value of the shape
    that is about to be launched is installed in front of every launch (a dictionary look-up; a C-ABI call only when the
    value differs from what the library holds) - a validation pass at another T or B neither inherits the training
    shape's delay nor leaves its own behind.  ``delay`` None = the built-in default."""
    key = (dev_i, kind)
    if key not in _POLL_DEFAULT:
        cur = (C.c_int * 4)()
        call('pbsed_gru_get_poll_delays', kind, cur)
        _POLL_DEFAULT[key] = list(cur)
        _POLL_INSTALLED[key] = list(cur)
    want = _POLL_DEFAULT[key][slot] if delay is None else delay
    have = _POLL_INSTALLED[key]
    if have[slot] != want:
        have[slot] = want
        call('pbsed_gru_set_poll_delays', kind, *have)


def _tune_poll_delay(dev, kind, slot, shape, launch, allow_tune=True):
    """In-place measurement of a persistent scan's first-poll delay (slot 0: forward, 2: BPTT; units of 64 clocks) on THIS
    device, with THIS shape and these operands: the built-in defaults were measured on one box in one DVFS state, the forward
    optimum is sharp (one unit early costs 30 %) and moves with clocks and with what shares the device.  At the first scan
    of a shape (T >= 64: short scans are not worth it) the scan is run a few times per candidate around the default - it is
    idempotent: outputs are rewritten, the workspace parity flips per launch - and ties go to the LATER delay (the safe side
    of the cliff).  The result is remembered PER SHAPE and installed in front of every launch of that shape
    (_install_poll_delay).  At most POLL_TUNE_MAX_SHAPES shapes per (device, kind, slot) are measured (inference over
    variable-length batches would otherwise pay ~30 extra scans and a host sync per distinct length; ``allow_tune=False``
    skips the measurement for a call); everything else runs the built-in delay.  PBSED_GRU_AUTOTUNE=0 keeps the defaults /
    PBSED_GRU_POLL_DELAYS.  One host sync per measured shape, never again."""
    dev_i = _dev_index(dev)
    key = (dev_i, kind, slot, shape)
    with torch.cuda.device(dev_i):
        if key in _POLL_TUNED:
            _install_poll_delay(dev_i, kind, slot, _POLL_TUNED[key])
            return
        n_tuned = sum(1 for k in _POLL_TUNED if k[:3] == key[:3])
        if shape[4] < 64 or not allow_tune or n_tuned >= POLL_TUNE_MAX_SHAPES or os.environ.get('PBSED_GRU_AUTOTUNE', '1') == '0' \
                or os.environ.get('PBSED_GRU_POLL_DELAYS'):
            _install_poll_delay(dev_i, kind, slot, None)
            return
        _install_poll_delay(dev_i, kind, slot, None)
        base = _POLL_DEFAULT[(dev_i, kind)][slot]
        timing_was = _lib.timing
        _lib.timing = None                          # the bench's event brackets must not see the tuning runs
        try:
            def measure(d):
                _install_poll_delay(dev_i, kind, slot, d)
                launch(tag='tune')
                e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                e0.record()
                for _ in range(3):
                    launch(tag='tune')
                e1.record()
                e1.synchronize()
                return e0.elapsed_time(e1) / 3

            def pick(cands, best):
                for d in cands:
                    ms = measure(d)
                    if best is None or ms < best[0] * .995:           # a later delay wins ties within 0.5 %
                        best = (ms, d)
                    elif ms <= best[0] * 1.005 and d > best[1]:
                        best = (min(ms, best[0]), d)
                return best
            # coarse pass over everything between "no wait" and a little past the built-in value (the XCD-local exchange moved
            # the optimum far below the write-through protocol's 20 .. 25 units), then single units around the best
            best = pick(range(0, base + 7, 3), None)
            best = pick([d for d in range(max(best[1] - 2, 0), best[1] + 3) if d != best[1]], best)
            _install_poll_delay(dev_i, kind, slot, best[1])
        finally:
            _lib.timing = timing_was
    check_gru_sync()
    _POLL_TUNED[key] = best[1]


def _granule_capacity(dev, h, bwd, bf16, tiles):
    """Blocks of the persistent scan kernel of this shape that can be co-resident on ``dev`` (pbsed_gru_granule_capacity: the
    occupancy API of that very kernel instantiation x CUs, one block per CU at most)."""
    key = (dev, h, bwd, bf16, tiles)
    if key not in _CU_COUNT:
        with torch.cuda.device(dev):
            _CU_COUNT[key] = int(_lib.lib().pbsed_gru_granule_capacity(h, int(bwd), int(bf16), tiles))
    return _CU_COUNT[key]


def _granule_scan(nch, nlayers, b, h, t, device=None, fwd=False, precision='f32'):
    """Persistent granule-exchange scans need every workgroup co-resident: a ring per (chain, layer) plus a projection
    group per layer boundary, each H/16 x ceil(B/16) blocks, on at most 7/8 of what the library reports as resident capacity
    for that kernel on this device (one block per CU where the kernel fits at all: 256 on an MI355X in SPX mode; a
    partitioned device or a kernel that does not fit falls back to the launch-per-step scans instead of raising).  The
    forward scan switches to two batch tiles per block (H/16 x ceil(B/32)) when one tile per block does not fit (the library
    applies the same rule).  The forward decision also covers the BPTT scan that will read its saved factors."""
    if h not in (64, 128, 256, 512) or os.environ.get('PBSED_GRU_PERSIST', '2') != '2':
        return False
    dev = torch.cuda.current_device() if device is None else device
    dev = dev.index if isinstance(dev, torch.device) else dev
    if dev is None:
        dev = torch.cuda.current_device()
    bf16 = precision == 'bf16'
    blocks = nch * (2 * nlayers - 1) * ((b + 15) // 16) * (h // 16)
    cap = min(_granule_capacity(dev, h, False, bf16, 1), _granule_capacity(dev, h, True, bf16, 1))
    if fwd and blocks > cap * 7 // 8:
        blocks = nch * (2 * nlayers - 1) * ((b + 31) // 32) * (h // 16)
        cap = _granule_capacity(dev, h, False, bf16, 2)
    return blocks <= cap * 7 // 8 and nch * nlayers * t * ((b + 15) // 16 * 16) * h * 4 < 2 ** 32


def _per_chain(nch, nlayers, b, h, t, dev, fwd=False):
    """More than 32 clips per GPU: all chains together need more co-resident workgroups than the device has CUs, one
    chain at a time fits (batch 64: 192 of 256) - run the persistent scans chain by chain instead of falling back to
    one launch per time step."""
    return nch > 1 and not _granule_scan(nch, nlayers, b, h, t, dev, fwd) and _granule_scan(1, nlayers, b, h, t, dev, fwd)


def gru_stack_fwd(gi0, w_ih, b_ih, w_hh, b_hh, reverse, seq_len, nlayers, save=True, precision='f32'):
    """Layer-wavefront scan of unidirectional stacks.  gi0: per chain [T,B,3H]; weight lists are indexed
    [chain*nlayers + layer] (w_ih/b_ih entries of layer 0 may be None).  Returns (hs, save) lists."""
    nch = len(gi0)
    t, b, g = gi0[0].shape
    h = g // 3
    dev = gi0[0].device
    # two batch tiles per block (more than 32 clips in one launch) unless the saved factors are for a BPTT that has to take
    # the launch-per-step kernels (other save format)
    nb2 = (not save) or _granule_scan(nch, nlayers, b, h, t, dev) or _per_chain(nch, nlayers, b, h, t, dev)
    if _per_chain(nch, nlayers, b, h, t, dev, fwd=nb2):
        hs, sv = [], []
        for c in range(nch):
            sl = slice(c * nlayers, (c + 1) * nlayers)
            hc, sc = gru_stack_fwd(gi0[c:c + 1], w_ih[sl], b_ih[sl], w_hh[sl], b_hh[sl], reverse[c:c + 1], seq_len, nlayers, save, precision)
            hs += hc
            sv += sc if save else []
        return hs, (sv if save else None)
    n = nch * nlayers
    hs = [torch.empty((t, b, h), device=dev, dtype=torch.float32) for _ in range(n)]
    gran = _granule_scan(nch, nlayers, b, h, t, dev, fwd=nb2)
    # saved per step: (r, z, n, gh_n), or in granule mode the five factors BPTT multiplies dh_t with
    sv = [torch.empty((t, b, 5 if gran else 4, h), device=dev, dtype=torch.float32) for _ in range(n)] if save else None
    if gran:
        key = (str(dev), n, t, b, h)
        gw = _GRANULE_WS.get(key)
        if gw is None:
            gw = _GRANULE_WS[key] = [torch.zeros(nch * t * ((b + 15) // 16 * 16) * h * (nlayers + 3 * (nlayers - 1)), dtype=torch.int32, device=dev), 0]
        ws = _gru_err_flag(dev)

        def launch(tag=f'{nch}x{nlayers} B{b} H{h} T{t}'):
            call('pbsed_gru_stack_fwd_granule_bf16' if precision == 'bf16' else 'pbsed_gru_stack_fwd_granule', nch, nlayers, _lib.ptr_array(gi0), _lib.ptr_array(w_ih),
                 _lib.ptr_array(b_ih), _lib.ptr_array(w_hh), _lib.ptr_array(b_hh), _lib.ptr_array(hs),
                 _lib.ptr_array(sv) if save else None, _lib.int_array(reverse), ptr(seq_len), b, h, t, ptr(gw[0]),
                 gw[1] + 1, ptr(ws), stream(), tag=tag,
                 flops=2. * nch * (2 * nlayers - 1) * t * b * 3 * h * h)        # recurrent + layer-boundary projection products
            gw[1] += 1                               # parity flips per launched call: the previous call's words never match

        two_tiles = nch * (2 * nlayers - 1) * ((b + 15) // 16) * (h // 16) > _granule_capacity(_dev_index(dev), h, False, precision == 'bf16', 1) * 7 // 8
        _tune_poll_delay(dev, 2 if two_tiles else 1 if nlayers == 1 else 0, 0, (nch, nlayers, b, h, t, precision), launch)
        watch_end = scan_watch.bracket(('fwd', nch, nlayers, b, h, t)) if scan_watch is not None else None
        launch()
        if watch_end is not None:
            watch_end.record()
        return hs, sv
    call('pbsed_gru_stack_fwd', nch, nlayers, _lib.ptr_array(gi0), _lib.ptr_array(w_ih), _lib.ptr_array(b_ih),
         _lib.ptr_array(w_hh), _lib.ptr_array(b_hh), _lib.ptr_array(hs), _lib.ptr_array(sv) if save else None,
         _lib.int_array(reverse), ptr(seq_len), b, h, t, stream())
    return hs, sv


def gru_stack_bwd(w_hh_t, w_ih_up_t, hs, save, dy_top, reverse, seq_len, nlayers, precision='f32'):
    nch = len(dy_top)
    t, b, h = hs[0].shape
    dev = hs[0].device
    if _per_chain(nch, nlayers, b, h, t, dev):
        dgi, dgh = [], []
        for c in range(nch):
            sl = slice(c * nlayers, (c + 1) * nlayers)
            a, g_ = gru_stack_bwd(w_hh_t[sl], w_ih_up_t[sl], hs[sl], save[sl], dy_top[c:c + 1], reverse[c:c + 1], seq_len, nlayers, precision)
            dgi += a
            dgh += g_
        return dgi, dgh
    n = nch * nlayers
    dgi = [torch.empty((t, b, 3 * h), device=dev, dtype=torch.float32) for _ in range(n)]
    dgh = [torch.empty((t, b, 3 * h), device=dev, dtype=torch.float32) for _ in range(n)]
    if _granule_scan(nch, nlayers, b, h, t, dev):
        assert save[0].shape[2] == 5, 'granule BPTT needs the granule forward scan\'s save format'
        key = (str(dev), 'bwd', n, t, b, h)
        gw = _GRANULE_WS.get(key)
        if gw is None:
            gw = _GRANULE_WS[key] = [torch.zeros(nch * t * ((b + 15) // 16 * 16) * h * (2 * nlayers - 1), dtype=torch.int32, device=dev), 0]
        ws = _gru_err_flag(dev)

        def launch(tag=f'{nch}x{nlayers} B{b} H{h} T{t}'):
            call('pbsed_gru_stack_bwd_granule_bf16' if precision == 'bf16' else 'pbsed_gru_stack_bwd_granule', nch, nlayers, _lib.ptr_array(w_hh_t), _lib.ptr_array(w_ih_up_t),
                 _lib.ptr_array(hs), _lib.ptr_array(save), _lib.ptr_array(dy_top), _lib.ptr_array(dgi), _lib.ptr_array(dgh),
                 _lib.int_array(reverse), ptr(seq_len), b, h, t, ptr(gw[0]), gw[1] + 1, ptr(ws), stream(),
                 tag=tag, flops=2. * nch * (2 * nlayers - 1) * t * b * 3 * h * h)
            gw[1] += 1

        _tune_poll_delay(dev, 1 if nlayers == 1 else 0, 2, (nch, nlayers, b, h, t, precision), launch)
        watch_end = scan_watch.bracket(('bwd', nch, nlayers, b, h, t)) if scan_watch is not None else None
        global last_scan_blocks
        last_scan_blocks = nch * (2 * nlayers - 1) * (h // 16) * ((b + 15) // 16)      # workgroups (= CUs) the scan holds while it runs
        launch()
        if watch_end is not None:
            watch_end.record()
        return dgi, dgh
    dhz = [torch.empty((t, b, h), device=dev, dtype=torch.float32) for _ in range(n)]
    call('pbsed_gru_stack_bwd', nch, nlayers, _lib.ptr_array(w_hh_t), _lib.ptr_array(w_ih_up_t), _lib.ptr_array(hs),
         _lib.ptr_array(save), _lib.ptr_array(dy_top), _lib.ptr_array(dgi), _lib.ptr_array(dgh), _lib.ptr_array(dhz),
         _lib.int_array(reverse), ptr(seq_len), b, h, t, stream())
    return dgi, dgh


def gru_wgrad(dg, x, shift, dw, db, precision='f32'):
    """dw[i] += dg[i]^T x[i] (time shift shift[i] on x), db[i] += column sums of dg[i]; all time-major [T,B,*]; the x[i] may
    differ in width (all weight gradients of a GRU backward pass go in one launch, at most 16 per launch).
    'f32': fp32-class products (exact bf16x3 operand splits on the bf16 MFMA); 'bf16': plain bf16 operands."""
    ensure_scratch(dg[0].device)
    t, b, g = dg[0].shape
    ks = [v.shape[2] for v in x]
    assert all(d.shape == (t, b, g) and d.is_contiguous() for d in dg) and all(v.shape[:2] == (t, b) and v.is_contiguous() for v in x)
    assert all(w.shape == (g, k) and w.is_contiguous() for w, k in zip(dw, ks))
    if LAUNCH_TAP is not None:
        LAUNCH_TAP.append(('gru_wgrad', list(dg), list(x), list(shift), list(dw), list(db), precision))
    for a in range(0, len(dg), 16):
        sl = slice(a, a + 16)
        call('pbsed_gru_wgrad_multi', len(dg[sl]), _lib.ptr_array(dg[sl]), _lib.ptr_array(x[sl]), _lib.int_array(shift[sl]),
             _lib.ptr_array(dw[sl]), _lib.ptr_array(db[sl]), t, b, g, _lib.int_array(ks[sl]), int(precision == 'bf16'), stream(),
             flops=2. * t * b * g * sum(ks[sl]))


def squash_fwd(x, eps):
    y = torch.empty_like(x)
    call('pbsed_squash_fwd', ptr(x), ptr(y), x.numel(), float(eps), stream())
    return y


def squash_bwd(y, dy, eps):
    dx = torch.empty_like(y)
    call('pbsed_squash_bwd', ptr(y), ptr(dy.contiguous()), ptr(dx), y.numel(), float(eps), stream())
    return dx


def fbcrnn_loss(logit_fwd, logit_bwd, weak_targets, boundary_targets, seq_len, *, minimum_score=1e-5,
                strong_weight=1., slat=False, label_smoothing=0., class_weights=None, want_grad=True,
                inputs_are_scores=False, summary=None):
    """``summary``: optional float32 [3*B*K + 1] device buffer the same launch fills with what CRNN.review reports to
    the host (weak-label mask, masked weak targets, clip-level scores, boundary label rate)."""
    b, k, t = logit_fwd.shape
    dev = logit_fwd.device
    y_f = torch.empty_like(logit_fwd)
    y_b = torch.empty_like(logit_fwd) if logit_bwd is not None else None
    d_f = torch.empty_like(logit_fwd) if want_grad else None
    d_b = torch.empty_like(logit_fwd) if (want_grad and logit_bwd is not None) else None
    loss = torch.empty((), device=dev, dtype=torch.float32)
    weak_c = weak_targets.contiguous()                  # named so two temporaries can never share storage
    bnd_c = None if boundary_targets is None else boundary_targets.contiguous()
    call('pbsed_fbcrnn_loss', ptr(logit_fwd), ptr(logit_bwd), ptr(weak_c), ptr(bnd_c), ptr(class_weights),
         ptr(seq_len), ptr(y_f), ptr(y_b), ptr(d_f), ptr(d_b), ptr(loss), b, k, t, float(minimum_score),
         float(strong_weight), int(slat), float(label_smoothing), int(inputs_are_scores), ptr(summary), stream())
    return loss, y_f, y_b, d_f, d_b


def bicrnn_loss(logit, strong_targets, seq_len, want_grad=True, inputs_are_scores=False):
    b, k, t = logit.shape
    y = torch.empty_like(logit)
    d = torch.empty_like(logit) if want_grad else None
    loss = torch.empty((), device=logit.device, dtype=torch.float32)
    scratch = torch.empty((), device=logit.device, dtype=torch.float64)
    call('pbsed_bicrnn_loss', ptr(logit), ptr(strong_targets.contiguous()), ptr(seq_len), ptr(y), ptr(d),
         ptr(loss), ptr(scratch), b, k, t, int(inputs_are_scores), stream())
    return loss, y, d


def bicrnn_review_summary(y, strong_targets, seq_len, segment_length, packed):
    """Fills ``packed`` (float32 [2*B*S*K + 2*B*K], S = T // segment_length) with y_seg | t_seg | mask_mean | mask_cnt."""
    b, k, t = y.shape
    s = t // segment_length
    n = b * s * k
    assert packed.numel() == 2 * n + 2 * b * k
    call('pbsed_bicrnn_review_summary', ptr(y.contiguous()), ptr(strong_targets.contiguous()), ptr(seq_len), ptr(packed[:n]),
         ptr(packed[n:2 * n]), ptr(packed[2 * n:2 * n + b * k]), ptr(packed[2 * n + b * k:]), b, k, t, int(segment_length), stream())


class LogMelTables:
    """Device tables for the fused front-end (window, twiddles, sparse mel filters)."""

    def __init__(self, fbanks, device):
        fb = np.asarray(fbanks, dtype=np.float32)
        nz = fb > 0
        start = np.array([int(np.argmax(r)) if r.any() else 0 for r in nz], dtype=np.int32)
        end = np.array([len(r) - int(np.argmax(r[::-1])) if r.any() else 0 for r in nz], dtype=np.int32)
        length = (end - start).astype(np.int32)
        off = np.concatenate([[0], np.cumsum(length)[:-1]]).astype(np.int32)
        w = np.concatenate([fb[m, start[m]:end[m]] for m in range(fb.shape[0])]).astype(np.float32)
        k = np.arange(960, dtype=np.float64)
        win = 0.42 - 0.5 * np.cos(2 * np.pi * k / 960) + 0.08 * np.cos(4 * np.pi * k / 960)
        q = np.arange(1024, dtype=np.float64)
        tw = np.stack([np.cos(-2 * np.pi * q / 1024), np.sin(-2 * np.pi * q / 1024)], -1)
        as_dev = lambda a, dt: torch.as_tensor(np.ascontiguousarray(a), dtype=dt).to(device)
        self.window = as_dev(win, torch.float32)
        self.twiddle = as_dev(tw, torch.float32)
        self.start, self.len, self.off = (as_dev(a, torch.int32) for a in (start, length, off))
        self.w = as_dev(w, torch.float32)
        self.n_filters = fb.shape[0]
        self.zero = torch.zeros(self.n_filters, device=device)
        self.one = torch.ones(self.n_filters, device=device)


def logmel_fwd(wav, tables, mean, inv_std, n_frames, seq_len=None, eps=1e-18, clamp=6.0, stats=None, pad_front=320,
               mel_points=None, frame_pos=None):
    """wav [B, N] f32 (device) -> normalised, clamped, masked log-mel [B, 1, F, T].  ``stats``: see
    feature_norm_stats (then pass mean = inv_std = None, clamp = None to get the raw log-mel).  ``frame_pos`` [B, T] int32
    (device): first sample of every frame's window (time-warped framing, pb_sed_amd/data.py::TimeWarp) instead of
    320 t - pad_front."""
    _lib.require_gpu(wav)
    b, n = wav.shape
    out = torch.empty((b, 1, tables.n_filters, n_frames), device=wav.device, dtype=torch.float32)
    if frame_pos is not None:
        assert frame_pos.shape == (b, n_frames) and frame_pos.dtype == torch.int32 and frame_pos.is_contiguous(), frame_pos.shape
        call('pbsed_logmel_fwd_frames', ptr(wav.contiguous()), b, n, n_frames, ptr(seq_len), ptr(frame_pos), ptr(tables.window),
             ptr(tables.twiddle), ptr(tables.start), ptr(tables.len), ptr(tables.off), ptr(tables.w), tables.w.numel(),
             tables.n_filters, ptr(tables.zero if mean is None else mean), ptr(tables.one if inv_std is None else inv_std),
             float(eps), float(clamp if clamp is not None else 3e38), ptr(out), ptr(stats), ptr(mel_points), stream(),
             nbytes=b * (n * 4 + tables.n_filters * n_frames * 4))
        return out
    call('pbsed_logmel_fwd', ptr(wav.contiguous()), b, n, n_frames, ptr(seq_len), ptr(tables.window),
         ptr(tables.twiddle), ptr(tables.start), ptr(tables.len), ptr(tables.off), ptr(tables.w), tables.w.numel(),
         tables.n_filters, ptr(tables.zero if mean is None else mean), ptr(tables.one if inv_std is None else inv_std),
         float(eps), float(clamp if clamp is not None else 3e38), ptr(out), ptr(stats), int(pad_front), ptr(mel_points), stream(),
         nbytes=b * (n * 4 + tables.n_filters * n_frames * 4))
    return out


def logmel_from_stft(stft, tables, mean, inv_std, seq_len=None, eps=1e-18, clamp=6.0, stats=None, mel_points=None):
    """stft [B, 1, T, bins, 2] f32 (the reference's ``inputs['stft']``) -> normalised, clamped, masked log-mel [B, 1, F, T]."""
    _lib.require_gpu(stft)
    b, c, t, bins, two = stft.shape
    assert c == 1 and two == 2, stft.shape
    x = stft.to(torch.float32).contiguous()
    out = torch.empty((b, 1, tables.n_filters, t), device=stft.device, dtype=torch.float32)
    call('pbsed_logmel_from_stft', ptr(x), b, t, bins, ptr(seq_len), ptr(tables.start), ptr(tables.len), ptr(tables.off),
         ptr(tables.w), tables.w.numel(), tables.n_filters, ptr(tables.zero if mean is None else mean),
         ptr(tables.one if inv_std is None else inv_std), float(eps), float(clamp if clamp is not None else 3e38),
         ptr(out), ptr(stats), ptr(mel_points), stream(), nbytes=b * t * (bins * 8 + tables.n_filters * 4))
    return out


def grad_sumsq(g, out):
    call('pbsed_grad_sumsq', ptr(g), g.numel(), ptr(out), stream())


def adam_step(p, g, m, v, *, lr, beta1=.9, beta2=.999, eps=1e-8, step, grad_scale=1., max_norm=1e10,
              sumsq=None, norm_out=None, skip_flags=None):
    call('pbsed_adam_step', ptr(p), ptr(g), ptr(m), ptr(v), p.numel(), float(lr), float(beta1),
         float(beta2), float(eps), int(step), float(grad_scale), float(max_norm), ptr(sumsq),
         ptr(norm_out), ptr(skip_flags), 0 if skip_flags is None else skip_flags.numel(), stream())


# ------------------------------------------------------------------------------------- post-processing
def ensemble_mean_mask(scores, seq_len):
    """scores: list of [B,(n,)K,T] device tensors (one per model) -> mean over models * sequence mask."""
    s0 = scores[0].contiguous()
    b, t = s0.shape[0], s0.shape[-1]
    r = s0.numel() // t
    out = torch.empty_like(s0)
    keep = [s.contiguous() for s in scores]
    call('pbsed_ensemble_mean_mask', _lib.ptr_array(keep), len(keep), ptr(out), ptr(seq_len), r // b, r, t, stream())
    return out


def _rows(x, per_row):
    t = x.shape[-1]
    r = x.numel() // t
    pr = np.broadcast_to(np.asarray(per_row), x.shape[:-1]).reshape(-1)
    return r, t, pr


def medfilt(scores, lengths):
    """Zero-padded median filter along the last axis; ``lengths`` broadcasts against scores.shape[:-1]."""
    x = scores.contiguous()
    r, t, n = _rows(x, lengths)
    out = torch.empty_like(x)
    n_dev = host_to_device(n, x.device, torch.int32)          # keep alive until the launch is enqueued
    call('pbsed_medfilt', ptr(x), ptr(out), ptr(n_dev), r, t, stream())
    return out


def boundariesfilt(scores, lengths, want_f64=False):
    x = scores.contiguous()
    r, t, n = _rows(x, lengths)
    out = torch.empty_like(x)
    out64 = torch.empty(x.shape, device=x.device, dtype=torch.float64) if want_f64 else None
    n_dev = host_to_device(n, x.device, torch.int32)
    call('pbsed_boundariesfilt', ptr(x), ptr(out), ptr(out64), ptr(n_dev), r, t, stream())
    return out64 if want_f64 else out


def event_frames(scores, thresholds, lengths, max_events=None):
    """scores [..., T] (class rows), per-row thresholds / valid lengths -> (events [R,max,2] int32, counts [R])."""
    x = scores.contiguous()
    r, t, th = _rows(x, thresholds)
    _, _, ln = _rows(x, lengths)
    max_events = max_events or (t // 2 + 1)
    ev = torch.zeros((r, max_events, 2), dtype=torch.int32, device=x.device)
    cnt = torch.zeros((r,), dtype=torch.int32, device=x.device)
    th_dev, ln_dev = host_to_device(th, x.device, torch.float32), host_to_device(ln, x.device, torch.int32)   # distinct live buffers
    call('pbsed_event_frames', ptr(x), ptr(th_dev), ptr(ln_dev), ptr(ev), ptr(cnt), r, t, max_events, stream())
    return ev, cnt
