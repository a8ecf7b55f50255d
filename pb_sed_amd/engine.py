"""Execution engine: drives the HIP kernels through the layer tables of ``pb_sed_amd.modules``.

Forward and backward are explicit chains of C-ABI calls (no per-op autograd graph); the model
classes wrap one whole network pass in a single ``torch.autograd.Function`` so that callers keep
the reference contract ``loss = model.review(batch, model(batch))['loss']; loss.backward()``
(reference pb_sed/models/weak_label/crnn.py:69-178, trainer loop SURVEY.md A.8).

Parameter gradients are accumulated by the kernels straight into ``p.grad`` (aliases of one flat
gradient buffer, see ``flatten_parameters``), which is what the fused Adam and the data-parallel
all-reduce operate on.
"""
import os

import numpy as np
import torch

from . import ops
from .ops import PackedConv


# ------------------------------------------------------------------------------------- seq_len cache
_SEQ_CACHE = {}


def seq_to_device(seq_host, device):
    """int32 device copy of a host seq_len array; cached by value (a pageable H2D copy stalls the host until
    the stream drains, and training loops present the same few length vectors over and over)."""
    key = (str(device), tuple(int(v) for v in np.asarray(seq_host).reshape(-1)))
    t = _SEQ_CACHE.get(key)
    if t is None:
        if len(_SEQ_CACHE) > 4096:
            _SEQ_CACHE.clear()
        t = ops.host_to_device(np.asarray(seq_host), device, torch.int32)
        _SEQ_CACHE[key] = t
    return t


# ------------------------------------------------------------------------------------- parameters
def flatten_parameters(module):
    """Re-home every parameter (and its .grad) of ``module`` in one flat fp32 buffer each.

    Returns (flat_param, flat_grad).  Idempotent; must be called after ``.to(device)``.
    """
    params = [p for p in module.parameters()]
    flat = getattr(module, '_pbsed_flat', None)
    if flat is not None and all(getattr(p, '_pbsed_flat_id', None) == id(flat[0]) and
                                p.data_ptr() == flat[0].data_ptr() + 4 * p._pbsed_off
                                for p in params):
        for p in params:
            if p.grad is None or p.grad.data_ptr() != flat[1].data_ptr() + 4 * p._pbsed_off:
                view = flat[1][p._pbsed_off:p._pbsed_off + p.numel()].view(p.shape)
                if p.grad is None:
                    view.zero_()          # optimizer.zero_grad(set_to_none=True): start from a clean slice
                else:
                    view.copy_(p.grad)    # keep gradients accumulated outside the flat buffer
                p.grad = view
        return flat
    device = params[0].device
    n = sum(p.numel() for p in params)
    fp = torch.empty(n, device=device, dtype=torch.float32)
    fg = torch.zeros(n, device=device, dtype=torch.float32)
    off = 0
    for p in params:
        k = p.numel()
        fp[off:off + k].copy_(p.detach().reshape(-1))
        p.data = fp[off:off + k].view(p.shape)
        p.grad = fg[off:off + k].view(p.shape)
        p._pbsed_off, p._pbsed_flat_id = off, id(fp)
        off += k
    module._pbsed_flat = (fp, fg)
    return module._pbsed_flat


def _grad(p):
    """Accumulation target for a parameter's gradient (None if frozen)."""
    if not p.requires_grad:
        return None
    if p.grad is None:
        p.grad = torch.zeros_like(p)
    return p.grad


# ------------------------------------------------------------------------------------- conv stacks
class LayerDesc:
    out_norm = None                 # last layer of a stack that closes with its own norm + ReLU (SURVEY.md A.4 reading (iii))

    def __init__(self, conv, in_norm):
        self.conv, self.in_norm = conv, in_norm
        self.skips_in = []          # residual connections ending at this layer's INPUT: (source layer index, skip conv | None)


def describe_stack(cnns):
    """Flatten one or more _CNN modules into [conv + the norm applied (with ReLU) to its input] (+ the residual
    connections of each stack, re-indexed to the flattened layer list)."""
    convs = [c for cnn in cnns for c in cnn.convs]
    layers = []
    for j, c in enumerate(convs):
        in_norm = c.norm if c.pre else (convs[j - 1].norm if j > 0 and convs[j - 1].post else None)
        layers.append(LayerDesc(c, in_norm))
    base = 0
    for cnn in cnns:
        for src, dst in enumerate(getattr(cnn, 'residual_connections', None) or []):
            if dst is not None:
                key = f'{src}_{dst}'
                layers[base + dst].skips_in.append((base + src, cnn.skip_convs[key] if key in cnn.skip_convs else None))
        base += len(cnn.convs)
    if convs[-1].post:
        raise NotImplementedError('a trailing post-activation norm (output_layer=False without '
                                  'pre_activation) has no consumer conv to fuse into')
    for k, cnn in enumerate(cnns):
        if getattr(cnn, 'out_norm', None) is not None:
            if k != len(cnns) - 1:
                raise NotImplementedError('a closing norm + ReLU is built for the LAST stack of a chain (the next stack\'s '
                                          'first layer would normalise again)')
            layers[-1].out_norm = cnn.out_norm            # its own launch: pbsed_bn_relu_fwd / pbsed_bn_relu_bwd
    return layers


# Verification tap: when a list is assigned, every conv layer of stack_forward appends what decided its ReLU and its
# (2,1) pool - ('layer', the layer's conv module, norm module whose ReLU feeds it | None, raw layer input, BN scale, BN shift,
# pool argmax bytes | None, BN state, operand format of the launch) - every skip-path pool ('skip', source conv, destination
# conv, crossed conv, argmax bytes), a stack's raw output ('out', last conv, y), and stack_backward the gradient arriving at
# every conv's output ('grad', conv, g) and leaving the stack ('grad_in', first conv, g): what a layer-by-layer check of the
# launches needs (tests/test_gpu_configs.py::test_c3_bf16_launches_in_situ).  The
# kernels decide a ReLU as fmaf(x, scale, shift) > 0 everywhere (forward prologues, data- and weight-gradient kernels), so
# the sign of the exact x * scale + shift reproduces it.  tests/hip_decisions.py turns the entries into the masks the float64
# oracle is run with (oracle/decisions.py).
DECISION_TAP = None

# PBSED_FUSE_BN_BWD=1 (OFF by default: measured 0.00 ms net on the C2 step): BN backward formed in the dY
# loader of the weight-gradient kernels that have one (ops.LazyBNGrad) instead of a stand-alone elementwise pass per norm.  The
# default is the stand-alone passes everywhere; tests flip the module attribute and restore what they found.
FUSE_BN_BWD = os.environ.get('PBSED_FUSE_BN_BWD', '0') != '0'
# PBSED_SIDE_WGRAD=1 (off by default: prepared while the GPU pool was closed to the build, NOT measured yet - DESIGN.md section 8):
# the weight gradients of the output heads run on a second stream NEXT TO the persistent BPTT scan instead of in front of it.  The
# scan occupies 192 of the 256 CUs and is latency-bound (7 % MFMA-pipe busy); the heads' weight gradients are leaves of the
# backward graph (nothing downstream reads them before the gradient norm), 0.12 - 0.2 ms per FBCRNN step.  The scan is enqueued
# first, the deferred launches behind an event recorded in front of it; the main stream joins the side stream at the end of
# the recurrent backward (before anything can read the gradients).  The library's slot scratch is per (device, stream).
SIDE_WGRAD = os.environ.get('PBSED_SIDE_WGRAD', '0') == '1'
_SIDE_STREAMS = {}


def _side_stream(device):
    dev = torch.device(device)
    idx = dev.index if dev.index is not None else torch.cuda.current_device()
    if idx not in _SIDE_STREAMS:
        _SIDE_STREAMS[idx] = torch.cuda.Stream(device=idx)
    return _SIDE_STREAMS[idx]


def _has_streams(t):
    return t.is_cuda


class _DeferredLaunches(list):
    """Weight-gradient launches collected during a stack_backward (closures over live tensors: they keep their operands alive
    until ``run`` has enqueued them and the main stream has joined the side stream)."""

    def run_beside(self, ready_event, device):
        """Enqueue the collected launches on the side stream behind ``ready_event`` (recorded on the main stream when their
        operands were final); returns the stream to join."""
        side = _side_stream(device)
        if self:
            # the launches' grids are cut for the CUs the scan leaves free (ops.cus_beside_last_scan): cut for the whole device,
            # three quarters of their workgroups would wait for the scan to end and the launch would finish no earlier than
            # it does behind the scan
            with torch.cuda.stream(side), ops.launch_cus(ops.cus_beside_last_scan(device)):
                side.wait_event(ready_event)
                for job in self:
                    job()
        return side


def _count(seq_host, t, rows):
    return float(np.minimum(np.asarray(seq_host), t).sum() * rows)


def _prec(precision, cin, pc=None, dgrad=False, unpool=False):
    """Operand format / algorithm of one conv launch.  bf16 modes need >= 32 input channels (below that the conv
    is HBM-bound and keeps fp32-class arithmetic).  In fp32 mode the MFMA-bound 3x3 layers run the Winograd-F(4,3)
    kernels (csrc/conv_wino.hip: same arithmetic type and results, half the multiplications): contraction over
    >= 32 channels into >= 64 output channels of the launch (a data gradient produces the layer's cin)."""
    if precision != 'f32' and cin >= 32:
        return precision                 # (layers below 32 input channels run fp32-class in every mode: the rules below)
    # Conv1d layers of the fp32 path: exact three-way bf16 operand splits on the bf16 MFMA (fp32-class results) - from 96 output
    # channels of the launch on (blocks of 128) the producer / consumer kernel of csrc/conv1d_pc.hip (weights streamed from L2 in
    # fragment order, four producer waves staging x), below that the pipelined kernel of csrc/conv_bf16.hip (NS = 3)
    if pc is not None and pc.weight.dim() == 3 and pc.cin >= 32 and pc.cout >= 32:
        n_out = pc.cin if dgrad else pc.cout
        return 'c1x3' if n_out >= 96 and pc.kw in (1, 3) else 'bf16x3'
    if pc is not None and pc.kh == 3 and pc.kw == 3:
        k_in, n_out = (pc.cout, pc.cin) if dgrad else (pc.cin, pc.cout)
        # bf16x3 Winograd (csrc/conv_winox3.hip: the transform-domain products from exact three-way bf16 splits on the bf16 MFMA)
        # from 32 channels on either side (32->32 forward 0.152 vs 0.216 ms direct, data gradient 0.138 vs 0.215; with 16 input
        # channels half of every K = 32 MFMA would be padding and the producers stage 32 channels per chunk whatever the layer
        # has - measured: 16->32 forward 0.144 vs 0.116 ms direct, data gradient 0.164 vs 0.130, 16->16 twice as slow); with 32-cout
        # blocks the data gradient through a pool into 32 channels is ahead too (0.193 vs 0.232 ms direct at 32->32).  Rows that
        # are not 16-byte aligned (T % 4) take the fp32-MFMA Winograd kernel of csrc/conv_wino.hip inside ops.conv_fwd.
        if k_in >= 32 and n_out >= 32:
            return 'winox3'
        # few-channel layers (contraction over <= 16 channels into <= 32: 16->16, 16->32 forward, the data gradients that
        # contract 16 channels, the tag-conditioned 11->16 first layer): csrc/conv_s16.hip - two taps x 16 channels per K = 32
        # MFMA, bf16x3 operands; the 1-channel first layer of the FBCRNN (9 products per output) stays on the direct kernel
        if 2 <= k_in <= 16 and n_out <= 32:
            return 's16x3'
    if pc is not None and pc.weight.dim() == 4 and pc.kh == 1 and pc.kw == 1:
        # 1x1 conv2d layers (every second layer of net_config 'deep'): the pipelined bf16-MFMA kernel with exact three-way
        # splits (csrc/conv_bf16.hip, NS = 3: one row per tile, two under a (2,1) pool; residual sums in its epilogue) from 32
        # channels on either side instead of the fp32-MFMA direct kernel
        k_in, n_out = (pc.cout, pc.cin) if dgrad else (pc.cin, pc.cout)
        if k_in >= 32 and n_out >= 32:
            return 'bf16x3'
    return 'f32'


class _StackCtx(list):
    """ctx of stack_forward (one entry per layer)."""
    final = None                # (raw output of the last conv, BN state, frozen) of a stack that closes with norm + ReLU


def stack_forward(layers, x, seq_dev, seq_host, training, precision='f32', x_tbc=None):
    """Run a conv stack.  Returns (y, ctx) - ctx is what stack_backward needs.  ``precision``: 'f32' |
    'bf16' | 'bf16x3' operand format of the forward / data-gradient MFMAs (weight gradients are always fp32).
    ``x_tbc``: the stack input in the scans' time-major layout [T,B,C] instead of ``x`` (the output nets behind the GRUs)."""
    ctx = _StackCtx()
    st_in, st_frozen = None, False
    if x is None:
        x = ops.tbc_to_bct(x_tbc)
    if layers[0].in_norm is not None:
        # a first layer with its own pre-activation norm (padertorch input_layer=False, SURVEY.md A.4 variant (i)): the batch
        # statistics of the stack input come from a reduction of their own, every other norm gets them from a conv epilogue
        n0 = layers[0].in_norm
        st_frozen = bool(training and n0.freeze_stats)
        if training and not n0.freeze_stats:
            rows = x.shape[2] if x.dim() == 4 else 1
            st_in = ops.bn_finalize(ops.channel_stats(x, seq_dev), _count(seq_host, x.shape[-1], rows), n0)
        else:
            st_in = ops.bn_eval_params(n0)
    for j, L in enumerate(layers):
        c = L.conv
        nxt = layers[j + 1] if j + 1 < len(layers) else None
        next_norm = nxt.in_norm if nxt is not None else L.out_norm
        # frozen statistics (cnn_2d.freeze(n, freeze_norm_stats=True), pb_sed/experiments/weak_label_crnn/training.py:343-350):
        # the layer normalises with its running statistics in training too and they are not updated
        batch_stats = training and next_norm is not None and not next_norm.freeze_stats
        per_cf = bool(batch_stats and c.ndim == 2 and nxt.conv.ndim == 1)
        if c.ndim == 1 and x.dim() == 4:
            x = x.flatten(1, 2)                       # 'b c f t -> b (c f) t' is a view in this layout
        pc = PackedConv(c.conv.weight)
        pr = _prec(precision, pc.cin, pc)
        # residual connections ending at the NEXT layer's input = this conv's output: their sum rides in this launch's
        # epilogue (added before the store and before the statistics of the next layer's norm)
        res, skip_ctx = None, []
        for src, skip_conv in (nxt.skips_in if nxt is not None else []):
            r, sctx = _skip_forward(layers, ctx, src, j + 1, skip_conv)
            res = r if res is None else ops.add_inplace(res, r)
            skip_ctx.append((src, skip_conv, sctx))
        if res is not None and pr != 'bf16x3':
            pr = 'f32'                       # the residual add lives in the epilogue of the direct fp32 / bf16-MFMA kernels
        y, idx, stats = ops.conv_fwd(
            x, pc, pc.fwd(pr), bias=c.conv.bias.detach(),
            scale=None if st_in is None else st_in.scale, shift=None if st_in is None else st_in.shift,
            relu=True, seq_len=seq_dev, pool=c.pool_f, want_stats=batch_stats, stats_per_cf=per_cf, precision=pr,
            residual=res)
        ctx.append((x, st_in, pc, idx, pr, st_frozen, skip_ctx))
        if DECISION_TAP is not None:
            DECISION_TAP.append(('layer', c, L.in_norm if st_in is not None else None, x,
                                 None if st_in is None else st_in.scale, None if st_in is None else st_in.shift, idx, st_in, pr))
        st_frozen = bool(training and next_norm is not None and next_norm.freeze_stats)
        if next_norm is None:
            st_in = None
        elif batch_stats:
            rows = 1 if (per_cf or y.dim() == 3) else y.shape[2]
            st_in = ops.bn_finalize(stats, _count(seq_host, y.shape[-1], rows), next_norm)
        else:
            st_in = ops.bn_eval_params(next_norm)
        x = y
    if DECISION_TAP is not None:
        DECISION_TAP.append(('out', layers[-1].conv, x))          # raw output of the stack's last conv
    if layers[-1].out_norm is not None:
        # the stack's closing norm + ReLU (st_in / st_frozen are the last conv's "next norm" state): a launch of its own
        ctx.final = (x, st_in, st_frozen)
        if DECISION_TAP is not None:
            DECISION_TAP.append(('final', layers[-1].out_norm, x, st_in.scale, st_in.shift))
        x = ops.bn_relu_fwd(x, st_in, seq_dev)
    return x, ctx


def _skip_forward(layers, ctx, src, dst, skip_conv):
    """Skip path x_src -> shape of x_dst: the (2,1) pools of the layers in between, then the 1x1 skip conv (if any)."""
    r = ctx[src][0]
    pools = []
    for k in range(src, dst):
        if layers[k].conv.pool_f:
            r, pidx = ops.pool21_fwd(r)
            pools.append(pidx)
            if DECISION_TAP is not None:
                DECISION_TAP.append(('skip', layers[src].conv, layers[dst].conv, layers[k].conv, pidx))
    pcs = None
    r_in = r
    if skip_conv is not None:
        pcs = PackedConv(skip_conv.weight)
        r, _, _ = ops.conv_fwd(r_in, pcs, pcs.fwd('f32'), bias=skip_conv.bias.detach(), seq_len=None)
    elif r is ctx[src][0]:
        r = r.clone()                    # summed in place with other residuals / never alias a saved activation
    return r, (pools, pcs, r_in)


def _skip_backward(ctx, src, skip_conv, sctx, g):
    """Gradient of the skip path: returns dL/dx_src contributed by the residual whose sum has gradient ``g``."""
    pools, pcs, r_in = sctx
    if skip_conv is not None:
        dw, db = _grad(skip_conv.weight), _grad(skip_conv.bias)
        if dw is not None:
            ops.conv_bwd_weight(r_in, g, pcs, dw, db, relu=False)
        g, _ = ops.conv_bwd_data(g, pcs, pcs.dgrad('f32'), r_in.shape)
    for pidx in reversed(pools):
        full = torch.zeros((*pidx.shape[:2], pidx.shape[2] * 2, pidx.shape[3]), device=g.device, dtype=torch.float32)
        g = ops.pool21_bwd_add(g, pidx, full)
    return g


def stack_backward(layers, ctx, g, seq_dev, seq_host, need_input_grad, on_layer_done=None, g_tbc=None, defer=None):
    """Backward of stack_forward; accumulates parameter grads, returns grad wrt the stack input.
    ``on_layer_done(j)`` fires once every gradient owned by layers >= j is final.  ``g_tbc``: the output gradient in the
    time-major layout [T,B,C] (instead of ``g``).  ``defer``: a _DeferredLaunches that takes the weight-gradient launches instead
    of running them (SIDE_WGRAD; the caller runs them on a side stream and joins it)."""
    def trainable(j):
        mods = [layers[j].conv.conv] + ([layers[j].in_norm] if layers[j].in_norm is not None else [])
        return any(p.requires_grad for m in mods for p in m.parameters())

    lowest = min((j for j in range(len(layers)) if trainable(j)), default=len(layers))
    if g is None:
        g = ops.tbc_to_bct(g_tbc)
    if ctx.final is not None:                    # backward through the closing ReLU + norm first
        x_raw, st_f, frozen_f = ctx.final
        norm_f = layers[-1].out_norm
        dz, stats_f = ops.bn_relu_bwd(g, x_raw, st_f, seq_dev)
        rows = 1 if x_raw.dim() == 3 else x_raw.shape[2]
        count = float('inf') if frozen_f else _count(seq_host, x_raw.shape[-1], rows)
        g = ops.bn_backward(dz, x_raw, st_f, stats_f, count, _grad(norm_f.gamma), _grad(norm_f.beta), seq_dev)
    pending = {}                                 # source layer -> gradient arriving over residual connections
    deferred_done = None                         # layer whose on_layer_done waits for its norm gradients (LazyBNGrad)
    for j in reversed(range(len(layers))):
        L = layers[j]
        c = L.conv
        x, st_in, pc, idx, pr, frozen, skip_ctx = ctx[j]
        dw, db = _grad(c.conv.weight), _grad(c.conv.bias)
        wprec = 'bf16' if pr == 'bf16' else 'f32'
        lazy = g if isinstance(g, ops.LazyBNGrad) else None
        if lazy is not None:
            # g = the BN backward of layer j + 1's input norm, not formed yet: this conv's weight-gradient kernel forms it in its
            # dY loader (and writes it for the data gradient) where it has that loader; everything else takes the stand-alone pass
            nxt_1d = j + 1 < len(layers) and layers[j + 1].conv.ndim == 1
            per_cf = bool(c.ndim == 2 and nxt_1d)
            fuse = (FUSE_BN_BWD and dw is not None and not skip_ctx and DECISION_TAP is None and not (j < lowest and not need_input_grad)
                    and ops.conv_bwd_weight_bng_supported(pc, x.shape[1], x.shape[2] if x.dim() == 4 else 1, x.shape[-1], per_cf, wprec))
            if not fuse:
                g, lazy = lazy.materialize(), None
                if deferred_done is not None:
                    on_layer_done(deferred_done)
                    deferred_done = None
        # g = dL/d(output of conv j) = dL/d(input of layer j+1): the residuals summed into it take the same gradient
        for src, skip_conv, sctx in skip_ctx:
            gs = _skip_backward(ctx, src, skip_conv, sctx, g.contiguous())
            pending[src] = gs if src not in pending else ops.add_inplace(pending[src], gs)
        if j < lowest and not need_input_grad:                # nothing below needs a gradient (frozen front layers)
            if on_layer_done is not None:
                on_layer_done(0)
            return None
        if lazy is None:
            g = g.contiguous()
        if DECISION_TAP is not None:
            DECISION_TAP.append(('grad', c, g))                   # gradient wrt this conv's (pooled) output
        if lazy is not None:
            g = ops.conv_bwd_weight(x, None, pc, dw, db,
                                    scale=None if st_in is None else st_in.scale,
                                    shift=None if st_in is None else st_in.shift,
                                    relu=True, seq_len=seq_dev, unpool_idx=idx, precision=wprec, bng=lazy, per_cf=per_cf)
            if deferred_done is not None:        # the norm gradients of layer j + 1 were written by this launch's coefficient pass
                on_layer_done(deferred_done)
                deferred_done = None
        elif dw is not None:
            def wgrad(x=x, g=g, pc=pc, dw=dw, db=db, st_in=st_in, idx=idx, wprec=wprec):
                ops.conv_bwd_weight(x, g, pc, dw, db,
                                    scale=None if st_in is None else st_in.scale,
                                    shift=None if st_in is None else st_in.shift,
                                    relu=True, seq_len=seq_dev, unpool_idx=idx, precision=wprec)
            if defer is not None:
                defer.append(wgrad)
            else:
                wgrad()
        norm0 = L.in_norm if (j == 0 and st_in is not None) else None
        norm0_grads = norm0 is not None and any(p.requires_grad for p in norm0.parameters())
        if j == 0 and not need_input_grad and not norm0_grads:
            if on_layer_done is not None:
                on_layer_done(0)
            return None
        pr = _prec('f32' if pr in ('wino', 'winox3', 'c1x3', 's16x3') else pr, pc.cin, pc, dgrad=True, unpool=idx is not None)
        wd = pc.dgrad(pr)
        if st_in is not None:
            dz, stats = ops.conv_bwd_data(g, pc, wd, x.shape, idx, seq_dev,
                                          bn=(x, st_in.mean, st_in.invstd, st_in.scale, st_in.shift), precision=pr)
            rows = 1 if x.dim() == 3 else x.shape[2]
            norm = L.in_norm
            # frozen statistics are constants: no mean / variance terms in the input gradient (count = inf drops them)
            count = float('inf') if frozen else _count(seq_host, x.shape[-1], rows)
            g = ops.LazyBNGrad(dz, x, st_in, stats, count, _grad(norm.gamma), _grad(norm.beta), seq_dev)
            if j == 0 or j in pending:
                g = g.materialize()              # no conv below to form it / a residual joins
        else:
            g, _ = ops.conv_bwd_data(g, pc, wd, x.shape, idx, None, precision=pr)
        if j in pending:                         # x_j also feeds a residual connection
            g = ops.add_inplace(g, pending.pop(j))
        if on_layer_done is not None and not isinstance(g, ops.LazyBNGrad):
            on_layer_done(j)        # layer j's in_norm belongs to it or to j-1's tail: both done now
        elif on_layer_done is not None:
            deferred_done = j       # dgamma / dbeta of layer j's norm are written by the next layer's launch
    if DECISION_TAP is not None:
        DECISION_TAP.append(('grad_in', layers[0].conv, g))      # gradient wrt the stack input
    return g


# ------------------------------------------------------------------------------------- recurrent part
class _Chain:
    """One GRU direction of one wrapper."""

    def __init__(self, wrapper, widx, direction):
        self.wrapper, self.widx, self.direction = wrapper, widx, direction
        self.reverse = bool(wrapper.reverse) ^ bool(direction)
        self.suffix = '_reverse' if direction else ''

    def p(self, name, layer):
        return getattr(self.wrapper.rnn, f'{name}_l{layer}{self.suffix}')


def _chains(wrappers):
    chains = []
    for wi, w in enumerate(wrappers):
        for d in range(2 if w.bidirectional else 1):
            chains.append(_Chain(w, wi, d))
    if len(chains) > 6:
        raise NotImplementedError('at most six concurrent GRU chains (pbsed.h: both directions of three networks)')
    if len(chains) > 2 and not _scan_as_stack(wrappers):
        raise NotImplementedError('more than two concurrent GRU chains need the stack scans (hidden size 64 / 128 / 256 / 512)')
    return chains


def _heads_forward(wrappers, x_w, seq_dev, seq_host, training, precision='f32', x_tbc=None):
    """``x_tbc``: per wrapper the head input in the scans' layout [T,B,C] (then ``x_w[wi]`` may be None)."""
    logits, head_ctx = [], []
    for wi, w in enumerate(wrappers):
        layers = describe_stack([w.output_net])
        y, c = stack_forward(layers, x_w[wi], seq_dev, seq_host, training, precision,
                             x_tbc=None if x_tbc is None else x_tbc[wi])
        logits.append(y)
        head_ctx.append((layers, c))
    return logits, head_ctx


def _gemm_prec(precision, k):
    """Operand format of a time-major projection (ops.tm_gemm): plain bf16 in the bf16 training mode, else fp32-class."""
    return 'bf16' if _prec(precision, k) == 'bf16' else 'f32'


def _stack_rnn_forward(wrappers, chains, h, seq_dev, seq_host, training, precision='f32', h_tbc=None):
    """Unidirectional multi-layer stacks (FBCRNN): layer-wavefront scan, T + L - 1 launches.  The input projections of the
    first layer run time-major (ops.tm_gemm on h transposed once) when the input width allows 16-byte rows."""
    nl = wrappers[0].num_layers
    gi0, pcs0 = [], []
    if h_tbc is None and h.shape[1] % 4 == 0:
        h_tbc = ops.bct_to_tbc(h)
    for ch in chains:
        w_ih = ch.p('weight_ih', 0)
        if h_tbc is not None:
            gi0.append(ops.tm_gemm([h_tbc], [w_ih.detach()], ch.p('bias_ih', 0).detach(), _gemm_prec(precision, w_ih.shape[1])))
            pcs0.append(None)
            continue
        pc = PackedConv(w_ih.unsqueeze(-1), owner=w_ih)
        pr = _prec(precision, pc.cin)
        y, _, _ = ops.conv_fwd(h, pc, pc.fwd(pr), bias=ch.p('bias_ih', 0).detach(), seq_len=None, precision=pr)
        gi0.append(ops.bct_to_tbc(y))
        pcs0.append(pc)
    idx = [(ch, l) for ch in chains for l in range(nl)]
    hs, save = ops.gru_stack_fwd(
        gi0, [ch.p('weight_ih', l).detach() if l else None for ch, l in idx],
        [ch.p('bias_ih', l).detach() if l else None for ch, l in idx],
        [ch.p('weight_hh', l).detach() for ch, l in idx], [ch.p('bias_hh', l).detach() for ch, l in idx],
        [ch.reverse for ch in chains], seq_dev, nl, save=training, precision='bf16' if precision == 'bf16' else 'f32')
    # one chain per wrapper: the heads take the top layer's states in the scans' layout
    top = [hs[ci * nl + nl - 1] for ci in range(len(chains))]
    logits, head_ctx = _heads_forward(wrappers, [None] * len(chains), seq_dev, seq_host, training, precision, x_tbc=top)
    return logits, ('stack', chains, (h, pcs0, hs, save, precision, h_tbc if training else None), head_ctx)


def _stack_rnn_backward(wrappers, ctx, dlogits, seq_dev, seq_host):
    _, chains, (h, pcs0, hs, save, precision, h_tbc), head_ctx = ctx
    nl = wrappers[0].num_layers
    dy_top = []
    defer = _DeferredLaunches() if (SIDE_WGRAD and DECISION_TAP is None and _has_streams(dlogits[0])) else None
    for wi, w in enumerate(wrappers):
        layers, c = head_ctx[wi]
        d = stack_backward(layers, c, dlogits[wi], seq_dev, seq_host, True, defer=defer)
        dy_top.append(ops.bct_to_tbc(d))
    idx = [(ch, l) for ch in chains for l in range(nl)]
    w_hh_t = [ops.transposed(ch.p('weight_hh', l)) for ch, l in idx]
    w_ih_up_t = [ops.transposed(ch.p('weight_ih', l + 1)) if l + 1 < nl else None for ch, l in idx]
    if defer is not None:
        heads_done = torch.cuda.Event()
        heads_done.record()                      # the heads' gradients wrt their outputs are final on the main stream here
    dgi, dgh = ops.gru_stack_bwd(w_hh_t, w_ih_up_t, hs, save, dy_top, [ch.reverse for ch in chains], seq_dev, nl,
                                 precision='bf16' if precision == 'bf16' else 'f32')
    side = defer.run_beside(heads_done, dlogits[0].device) if defer is not None else None     # the scan is enqueued: now its neighbours
    dh = None
    jobs = ([], [], [], [], [])                  # all weight gradients of the stacks: one launch
    for ci, ch in enumerate(chains):
        for l in range(nl):
            i = ci * nl + l
            w_hh, w_ih = ch.p('weight_hh', l), ch.p('weight_ih', l)
            if w_hh.requires_grad:
                _wgrad_job(jobs, dgh[i], hs[i], 1 if ch.reverse else -1, _grad(w_hh), _grad(ch.p('bias_hh', l)))
            if w_ih.requires_grad:
                if l == 0 and h_tbc is None:
                    h_tbc = ops.bct_to_tbc(h)
                _wgrad_job(jobs, dgi[i], h_tbc if l == 0 else hs[i - 1], 0, _grad(w_ih), _grad(ch.p('bias_ih', l)))
            if l == 0 and pcs0[ci] is not None:
                pr = _prec(precision, pcs0[ci].cin)
                dx, _ = ops.conv_bwd_data(ops.tbc_to_bct(dgi[i]), pcs0[ci], pcs0[ci].dgrad(pr), h.shape, precision=pr)
                dh = dx if dh is None else dh.add_(dx)
    if pcs0[0] is None:
        # data gradient of the first layers' input projections, all chains in one time-major product: dh = sum_c dgi_c W_ih,c
        w_t = [ops.transposed(ch.p('weight_ih', 0)) for ch in chains]
        dh = ops.tbc_to_bct(ops.tm_gemm([dgi[ci * nl] for ci in range(len(chains))], w_t, None, _gemm_prec(precision, w_t[0].shape[1]), role='bwd'))
    if jobs[0]:
        ops.gru_wgrad(*jobs, precision='bf16' if precision == 'bf16' else 'f32')
    if side is not None:
        torch.cuda.current_stream().wait_stream(side)      # nothing behind this point may run before the heads' gradients are final
        defer.clear()
    return dh


def _wgrad_job(jobs, dg, x, shift, dw, db):
    """``jobs``: (dg, x, shift, dw, db) lists of ONE batched weight-gradient launch (ops.gru_wgrad) per backward pass."""
    for lst, v in zip(jobs, (dg, x, shift, dw, db)):
        lst.append(v)


def _scan_as_stack(wrappers):
    return wrappers[0].hidden_size in (64, 128, 256, 512)


def rnn_forward(wrappers, h, seq_dev, seq_host, training, precision='f32', h_tbc=None):
    """wrappers: list of modules.GRU sharing the input h [B,C,T] (``h_tbc``: the same in the scans' layout [T,B,C], if the
    caller has it; a LIST of such tensors, one per wrapper and h = None: independent networks whose layers share the scan
    launches - inference only, no ctx for a backward pass).  Returns (logits per wrapper, ctx).

    Layer-by-layer path (bidirectional GRUs): every layer's input projection is a time-major product (ops.tm_gemm) - of the
    transposed CNN output for the first layer, of the previous layer's scan outputs (one source per direction, nothing
    concatenated) above it; input widths that are no multiple of 4 (tag-conditioned first layers) take the k = 1 convolution
    on the CNN layout."""
    chains = _chains(wrappers)
    num_layers = wrappers[0].num_layers
    assert all(w.num_layers == num_layers for w in wrappers)
    if h is not None and not any(w.bidirectional for w in wrappers) and wrappers[0].hidden_size in (64, 128, 256, 512) \
            and all(w.rnn.input_size == h.shape[1] for w in wrappers):
        return _stack_rnn_forward(wrappers, chains, h, seq_dev, seq_host, training, precision, h_tbc)
    of_w = [[i for i, ch in enumerate(chains) if ch.widx == wi] for wi in range(len(wrappers))]
    # ``h_tbc`` may carry zero channels behind the input_size real ones (widths like 266 = 256 + 10 tags padded to whole
    # float4s): the first layer's weights are padded with zero columns to match
    if h_tbc is None and h.shape[1] % 4 == 0:
        h_tbc = ops.bct_to_tbc(h)
    if isinstance(h_tbc, (list, tuple)):          # one input per wrapper: independent networks sharing the scan launches
        assert len(h_tbc) == len(wrappers) and h is None
        src = [[x] for x in h_tbc]
    else:
        src = [[h_tbc] if h_tbc is not None else None for _ in wrappers]      # per wrapper: time-major sources of the layer input
    layer_ctx, hs = [], None
    for l in range(num_layers):
        gi, pcs = [], []
        for ch in chains:
            w_ih = ch.p('weight_ih', l)
            xs = src[ch.widx]
            if xs is not None:
                if len(xs) == 1:
                    ws = [w_ih.detach()]
                    if xs[0].shape[2] != w_ih.shape[1]:          # zero-padded input channels
                        ws = [torch.cat([ws[0], ws[0].new_zeros((w_ih.shape[0], xs[0].shape[2] - w_ih.shape[1]))], dim=1)]
                else:                                  # one column block of W_ih per source
                    edges = np.cumsum([0] + [x.shape[2] for x in xs])
                    ws = [w_ih.detach()[:, edges[j]:edges[j + 1]].contiguous() for j in range(len(xs))]
                gi.append(ops.tm_gemm(xs, ws, ch.p('bias_ih', l).detach(), _gemm_prec(precision, w_ih.shape[1])))
                pcs.append(None)
                continue
            pc = PackedConv(w_ih.unsqueeze(-1), owner=w_ih)
            pr = _prec(precision, pc.cin)
            y, _, _ = ops.conv_fwd(h, pc, pc.fwd(pr), bias=ch.p('bias_ih', l).detach(), seq_len=None, precision=pr)
            gi.append(ops.bct_to_tbc(y))
            pcs.append(pc)
        w_hh_l = [ch.p('weight_hh', l).detach() for ch in chains]
        b_hh_l = [ch.p('bias_hh', l).detach() for ch in chains]
        if _scan_as_stack(wrappers):
            # one layer of a (bi)directional GRU = len(chains) independent one-layer stacks: persistent scan
            none = [None] * len(chains)
            hs, save = ops.gru_stack_fwd(gi, none, none, w_hh_l, b_hh_l, [ch.reverse for ch in chains], seq_dev, 1,
                                         save=training, precision='bf16' if precision == 'bf16' else 'f32')
        else:
            hs, save = ops.gru_scan_fwd(gi, w_hh_l, b_hh_l, [ch.reverse for ch in chains], seq_dev, save=training)
        layer_ctx.append((src, pcs, hs, save))
        src = [[hs[i] for i in of_w[wi]] for wi in range(len(wrappers))]
    # the heads take the top layer's states in the scans' layout (both directions side by side)
    top = []
    for wi, w in enumerate(wrappers):
        outs = [hs[i] for i in of_w[wi]]
        top.append(outs[0] if len(outs) == 1 else torch.cat(outs, dim=2))
    logits, head_ctx = _heads_forward(wrappers, [None] * len(wrappers), seq_dev, seq_host, training, precision, x_tbc=top)
    return logits, (chains, layer_ctx, head_ctx, precision, h)


def rnn_backward(wrappers, ctx, dlogits, seq_dev, seq_host):
    """Returns grad wrt the shared input h."""
    if ctx[0] == 'stack':
        return _stack_rnn_backward(wrappers, ctx, dlogits, seq_dev, seq_host)
    chains, layer_ctx, head_ctx, precision, h = ctx
    num_layers = wrappers[0].num_layers
    hid = wrappers[0].hidden_size
    of_w = [[i for i, ch in enumerate(chains) if ch.widx == wi] for wi in range(len(wrappers))]
    dy = [None] * len(chains)                    # per chain: grad wrt its top-layer output, time-major [T,B,H]
    # SIDE_WGRAD: leaves of the backward graph run on a second stream beside the NEXT persistent scan (a one-layer BiGRU scan
    # occupies 64 of the 256 CUs): the heads' weight gradients beside the top layer's scan, a layer's GRU weight gradients
    # beside the scan of the layer below; the bottom layer's follow on the main stream as before
    beside = SIDE_WGRAD and DECISION_TAP is None and _has_streams(dlogits[0]) and _scan_as_stack(wrappers)
    defer = _DeferredLaunches() if beside else None
    side, held = None, []
    for wi, w in enumerate(wrappers):
        layers, c = head_ctx[wi]
        d_out = stack_backward(layers, c, dlogits[wi], seq_dev, seq_host, True, defer=defer)       # [B, H*dirs, T]
        for k, i in enumerate(of_w[wi]):
            dy[i] = ops.bct_to_tbc(d_out[:, k * hid:(k + 1) * hid].contiguous())
    dh_in = None
    jobs = ([], [], [], [], [])                  # weight gradients of ALL layers: one launch after the last scan
    padded = []                                  # (gradient of a zero-padded W_ih, the parameter's gradient)
    for l in reversed(range(num_layers)):
        if beside:
            ready = torch.cuda.Event()
            ready.record()                       # what the deferred launches read is final on the main stream here
        src, pcs, hs, save = layer_ctx[l]
        w_hh_t = [ops.transposed(ch.p('weight_hh', l)) for ch in chains]
        if _scan_as_stack(wrappers):
            dgi, dgh = ops.gru_stack_bwd(w_hh_t, [None] * len(chains), hs, save, dy, [ch.reverse for ch in chains],
                                         seq_dev, 1, precision='bf16' if precision == 'bf16' else 'f32')
            if beside:                           # this layer's scan is enqueued: the launches collected so far run beside it
                if jobs[0]:                      # (the GRU weight gradients of the layer above)
                    defer.append(lambda j=jobs: ops.gru_wgrad(*j, precision='bf16' if precision == 'bf16' else 'f32'))
                    jobs = ([], [], [], [], [])
                side = defer.run_beside(ready, dlogits[0].device)
                held.extend(defer)               # the closures keep their operands alive until the join
                del defer[:]
        else:
            dgi, dgh = ops.gru_scan_bwd(w_hh_t, hs, save, dy, [ch.reverse for ch in chains], seq_dev)
        x_cat = {}                                 # the layer input once per wrapper, time-major, for the weight gradients
        dgi_bct = {}                               # conv-path first layers: dgi on the CNN layout, once per chain

        def dgi_b(i):
            if i not in dgi_bct:
                dgi_bct[i] = ops.tbc_to_bct(dgi[i])
            return dgi_bct[i]
        for i, ch in enumerate(chains):
            w_hh, w_ih = ch.p('weight_hh', l), ch.p('weight_ih', l)
            if w_hh.requires_grad:
                _wgrad_job(jobs, dgh[i], hs[i], 1 if ch.reverse else -1, _grad(w_hh), _grad(ch.p('bias_hh', l)))
            if w_ih.requires_grad:
                if pcs[i] is None:
                    if ch.widx not in x_cat:
                        xs = src[ch.widx]
                        x_cat[ch.widx] = xs[0] if len(xs) == 1 else torch.cat(xs, dim=2)
                    dw = _grad(w_ih)
                    if x_cat[ch.widx].shape[2] != w_ih.shape[1]:     # zero-padded input channels: gradient of the padded matrix
                        dw_pad = dw.new_zeros((w_ih.shape[0], x_cat[ch.widx].shape[2]))
                        padded.append((dw_pad, dw))
                        dw = dw_pad
                    _wgrad_job(jobs, dgi[i], x_cat[ch.widx], 0, dw, _grad(ch.p('bias_ih', l)))
                else:                                # e.g. 266 = 256 + 10 tag-conditioned inputs: not a float4 multiple
                    ops.conv_bwd_weight(h, dgi_b(i), pcs[i], _grad(w_ih), _grad(ch.p('bias_ih', l)),
                                        precision='bf16' if _prec(precision, pcs[i].cin) == 'bf16' else 'f32')
        # data gradient of the input projections: per wrapper, the sum over its chains of dgi W_ih
        new_dy = [None] * len(chains)
        for wi in range(len(wrappers)):
            mine = of_w[wi]
            if pcs[mine[0]] is None:
                w_t = [ops.transposed(chains[i].p('weight_ih', l)) for i in mine]       # [In, 3H]
                k_in = src[wi][0].shape[2] if len(src[wi]) == 1 else w_t[0].shape[0]
                if k_in != w_t[0].shape[0]:               # zero-padded input channels: zero rows
                    w_t = [torch.cat([w, w.new_zeros((k_in - w.shape[0], w.shape[1]))]) for w in w_t]
                gp = _gemm_prec(precision, w_t[0].shape[1])
                if l > 0:                            # straight into the per-chain gradients of the layer below
                    for k, i in enumerate(mine):
                        new_dy[i] = ops.tm_gemm([dgi[j] for j in mine], [w[k * hid:(k + 1) * hid] for w in w_t], None, gp, role='bwd')
                else:
                    d = ops.tbc_to_bct(ops.tm_gemm([dgi[j] for j in mine], w_t, None, gp, role='bwd'))
                    dh_in = d if dh_in is None else dh_in.add_(d)
            else:                                    # first layer on the CNN layout
                for i in mine:
                    pr = _prec(precision, pcs[i].cin)
                    d, _ = ops.conv_bwd_data(dgi_b(i), pcs[i], pcs[i].dgrad(pr), h.shape, precision=pr)
                    dh_in = d if dh_in is None else dh_in.add_(d)
        dy = new_dy
    if jobs[0]:
        ops.gru_wgrad(*jobs, precision='bf16' if precision == 'bf16' else 'f32')
    if side is not None:
        torch.cuda.current_stream().wait_stream(side)      # nothing behind this point may run before the deferred gradients are final
        del held[:]
    for dw_pad, dw in padded:
        dw.add_(dw_pad[:, :dw.shape[1]])
    return dh_in


# ------------------------------------------------------------------------------------- front-end
def _tables(fe, device):
    if fe._tables is None or fe._tables.window.device != device:
        fe._tables = ops.LogMelTables(fe.fbanks.detach().cpu().numpy(), device)
    return fe._tables


def _features(fe, raw_fn, seq_dev, seq_host, n_frames):
    """Shared driver of both front-end entry points.  Eval / frozen statistics: ONE fused launch (normalised, clamped,
    masked).  Training with statistics tracking (the reference's NormalizedLogMelExtractor default): raw
    log-mel + per-mel sums in the same launch -> cumulative statistics update -> normalise / clamp (+ augmentation) in
    place.  ``raw_fn(mean, inv_std, clamp, stats, mel_points)`` launches the front-end kernel."""
    track = fe.training and not fe.freeze_stats
    pts = None
    if fe.training and fe.frequency_warping_fn is not None:       # per-clip mel warping: filters built inside the kernel
        fe.last_mel_points = fe.sample_mel_points(len(seq_host))
        pts = torch.from_numpy(fe.last_mel_points).to(seq_dev.device)
    if not track:
        x = raw_fn(fe.mean, fe.inv_std, fe.clamp, None, pts)
        return augment_features(fe, x, seq_dev, seq_host) if (fe.training and fe.augments) else x
    stats = ops.feature_norm_stats(fe.number_of_filters, seq_dev.device)
    x = raw_fn(None, None, None, stats, pts)
    count = float(np.minimum(np.asarray(seq_host), n_frames).sum())
    ops.feature_norm_update(stats, count, fe)
    return augment_features(fe, x, seq_dev, seq_host, normalise=True)


def features_from_audio(fe, audio, seq_dev, n_frames, seq_host=None, pad_front=320, frame_pos=None):
    """``frame_pos`` [B, n_frames] int32: time-warped framing (pb_sed_amd/data.py::TimeWarp)."""
    if not getattr(fe, 'fused_waveform_frontend', True):
        raise NotImplementedError(f'the fused waveform front-end is built for STFT 1024 / 960 / 320, not {fe.stft_size} / '
                                  f"{fe.window_length} / {fe.shift}: hand the model inputs['stft'] (pbsed_logmel_from_stft)")
    tables = _tables(fe, audio.device)
    if seq_host is None:
        seq_host = seq_dev.cpu().numpy()
    return _features(fe, lambda mean, inv_std, clamp, stats, pts: ops.logmel_fwd(
        audio, tables, mean, inv_std, n_frames, seq_dev, eps=fe.eps, clamp=clamp, stats=stats, pad_front=pad_front,
        mel_points=pts, frame_pos=frame_pos), seq_dev, seq_host, n_frames)


def augment_features(fe, x, seq_dev, seq_host=None, normalise=False):
    """Training-only second pass over the features: optional normalisation with the just-updated statistics, then the
    augmentation of the reference's training config (noise, time mask, frequency mask; draws on the host) - one launch."""
    masks = scales = noise = None
    if fe.augments:
        if seq_host is None:
            seq_host = seq_dev.cpu().numpy()
        masks, scales = fe.sample_augmentation(seq_host)
        masks = torch.from_numpy(masks).to(x.device)
        if scales is not None:
            noise, scales = torch.randn_like(x), torch.from_numpy(scales).to(x.device)
    return ops.augment_logmel(x, masks, seq_dev, noise, scales, fe.mean if normalise else None,
                              fe.inv_std if normalise else None, fe.clamp if normalise else None)


def features_from_stft(fe, stft, seq_host, seq_dev=None):
    """The reference's own input contract: a CPU-computed complex STFT ``inputs['stft']`` [B,1,T,bins,2]
    (pb_sed/models/weak_label/crnn.py:79-90) through ``pbsed_logmel_from_stft`` (same epilogue as the audio path)."""
    tables = _tables(fe, stft.device)
    if seq_dev is None:
        seq_dev = seq_to_device(seq_host, stft.device)
    return _features(fe, lambda mean, inv_std, clamp, stats, pts: ops.logmel_from_stft(
        stft, tables, mean, inv_std, seq_dev, eps=fe.eps, clamp=clamp, stats=stats, mel_points=pts),
        seq_dev, seq_host, stft.shape[2])
