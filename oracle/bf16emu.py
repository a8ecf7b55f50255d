"""Oracle: the bf16 training mode (BASELINE.json configs[2]: "bf16 on 1xMI355X") restated on the CPU.

TEST INFRASTRUCTURE ONLY (see oracle/__init__.py).  Parity unpinned in the sense of oracle/nn.py: the reference has no
bf16 path of its own (pb_sed/experiments/strong_label_crnn/training.py runs fp32 on whatever device padertorch is given);
"bf16" is BASELINE.json's configuration of this build, and what it means is defined HERE and in DESIGN.md section 2:

    every matrix product of the network takes both operands ROUNDED TO NEAREST-EVEN bf16 at the point where the HIP
    kernels stage them (v_cvt_pk_bf16_f32 / the packers' f2bf_rne) and accumulates in fp32; everything else - batch-norm
    statistics and apply, ReLU, pooling, gate maths, GRU state, losses, weight / gradient storage - is fp32.

The products and their operands, forward and backward (the backward products are NOT the derivative of the rounded
forward: each rounds its own operands, which is what the kernels do and what this module mirrors with autograd Functions):

* convolutions with >= 32 input channels (pb_sed_amd/engine.py::_prec; the 11- / 16-channel layers stay fp32):
  y = conv(bf16(relu(norm(x)) * mask), bf16(w)) + b;   dx = conv^T(bf16(dy), bf16(w));
  dw = corr(bf16(dy), bf16(relu(norm(x)) * mask));   db = sum bf16(dy)   (the kernels' ones-vector MFMA; both from the
  unrounded operands when the layer has fewer than 32 OUTPUT channels: the heads' class layers keep the fp32 kernels)
* time-major projections of the GRUs (W_ih x + b_ih, all layers / directions): y = bf16(x) bf16(W)^T + b,
  dx = bf16(dy) bf16(W), dW = bf16(dy)^T bf16(x), db = sum dy
* the recurrence: gh_t = bf16(h_{t-1}) bf16(W_hh)^T + b_hh;   BPTT: dh_{t-1} += bf16(d gh_t) bf16(W_hh)
  (d gh_t = the three gate-gradient operands dh_t * factor the backward scan forms);
  dW_hh = sum_t bf16(d gh_t)^T bf16(h_{t-1}), db_hh = sum_t d gh_t

Arithmetic around the roundings runs in the dtype of the model (tests use float64): a HIP run differs from this oracle by
fp32 accumulation order and by the few operands whose fp32 value sits within rounding of a bf16 tie - three orders of
magnitude below the effect of the bf16 operand rounding itself (2^-9 relative per operand), which a comparison against
the fp32 oracle cannot see through.
"""
import numpy as np
import torch
import torch.nn.functional as F
from torch import nn

from . import nn as onn

MIN_CIN = 32            # engine._prec: below this the conv is HBM-bound and runs the fp32 kernel in every mode


def rbf(x):
    """Round to the nearest bf16 (ties to even), keep the dtype."""
    return x.to(torch.bfloat16).to(x.dtype)


class _ConvBf16(torch.autograd.Function):
    @staticmethod
    def forward(ctx, x, w, b):
        xr, wr = rbf(x), rbf(w)
        # the weight gradient takes bf16 operands from 32 OUTPUT channels on as well (csrc/conv_wgrad.hip::conv_wgrad_launch:
        # narrower layers - the 10-class output layers of the heads - keep the fp32 weight-gradient kernels)
        ctx.wgrad_rounded = w.shape[0] >= MIN_CIN
        ctx.save_for_backward(xr, wr, xr if ctx.wgrad_rounded else x)
        ctx.has_bias = b is not None
        conv = F.conv2d if w.dim() == 4 else F.conv1d
        return conv(xr, wr, b)

    @staticmethod
    def backward(ctx, dy):
        xr, wr, xw = ctx.saved_tensors
        dyr = rbf(dy)
        dyw = dyr if ctx.wgrad_rounded else dy
        if wr.dim() == 4:
            dx = torch.nn.grad.conv2d_input(xr.shape, wr, dyr)
            dw = torch.nn.grad.conv2d_weight(xw, wr.shape, dyw)
        else:
            dx = torch.nn.grad.conv1d_input(xr.shape, wr, dyr)
            dw = torch.nn.grad.conv1d_weight(xw, wr.shape, dyw)
        db = dyw.sum([0] + list(range(2, dyw.dim()))) if ctx.has_bias else None
        return dx, dw, db


class _LinearBf16(torch.autograd.Function):
    """y[..., n] = sum_k bf16(x[..., k]) bf16(w[n, k]) (+ b added by the caller)."""

    @staticmethod
    def forward(ctx, x, w):
        xr, wr = rbf(x), rbf(w)
        ctx.save_for_backward(xr, wr)
        return xr @ wr.t()

    @staticmethod
    def backward(ctx, dy):
        xr, wr = ctx.saved_tensors
        dyr = rbf(dy)
        return dyr @ wr, dyr.reshape(-1, dyr.shape[-1]).t() @ xr.reshape(-1, xr.shape[-1])


def _conv_layer_forward(self, x, seq_len=None):
    """oracle/nn.py::_ConvLayer.forward with the product replaced (same decisions machinery)."""
    if self.pre:
        x = self._relu(self.norm(x, seq_len))
    p = self.k - 1
    lo, hi = p // 2, int(np.ceil(p / 2))
    x = F.pad(x, (lo, hi, lo, hi) if self.ndim == 2 else (lo, hi))
    if self.conv.in_channels >= MIN_CIN:
        y = _ConvBf16.apply(x, self.conv.weight, self.conv.bias)
    else:
        y = self.conv(x)
    if self.post:
        y = self._relu(self.norm(y, seq_len))
    if self.pool != 1:
        y = self._pool(y)
    return y


def _scan(gi, w_hh, b_hh, mask):
    """One direction of one layer.  gi [B,T,3H] = W_ih x + b_ih, mask [B,T] (t < seq_len) -> outputs [B,T,H] (zero past the
    sequence, as pad_packed_sequence leaves them).  torch.nn.GRU's gate order (r, z, n) and update rule."""
    b, t, h3 = gi.shape
    hd = h3 // 3
    h = gi.new_zeros((b, hd))
    ys = []
    for s in range(t):
        gh = _LinearBf16.apply(h, w_hh) + b_hh
        r = torch.sigmoid(gi[:, s, :hd] + gh[:, :hd])
        z = torch.sigmoid(gi[:, s, hd:2 * hd] + gh[:, hd:2 * hd])
        n = torch.tanh(gi[:, s, 2 * hd:] + r * gh[:, 2 * hd:])
        m = mask[:, s, None]
        h_new = (1 - z) * n + z * h
        ys.append(h_new * m)
        h = h_new * m + h * (1 - m)
    return torch.stack(ys, dim=1)


def _gru_forward(self, x, seq_len=None):
    """oracle/nn.py::GRU.forward with torch.nn.GRU replaced by the explicit scans above (packed-sequence semantics)."""
    x = x.transpose(1, 2)                                   # b t f
    b, t, _ = x.shape
    if self.reverse:
        x = onn.reverse_sequence(x, seq_len)
    sl = np.full(b, t) if seq_len is None else np.asarray(seq_len)
    mask = (torch.arange(t)[None] < torch.as_tensor(sl)[:, None]).to(x.dtype)
    rnn = self.rnn
    layer_in = x
    for l in range(rnn.num_layers):
        outs = []
        for d in range(2 if rnn.bidirectional else 1):
            sfx = f'_l{l}' + ('_reverse' if d else '')
            w_ih, w_hh = getattr(rnn, 'weight_ih' + sfx), getattr(rnn, 'weight_hh' + sfx)
            b_ih, b_hh = getattr(rnn, 'bias_ih' + sfx), getattr(rnn, 'bias_hh' + sfx)
            xin = onn.reverse_sequence(layer_in, sl) if d else layer_in
            y = _scan(_LinearBf16.apply(xin, w_ih) + b_ih, w_hh, b_hh, mask)
            outs.append(onn.reverse_sequence(y, sl) if d else y)
        layer_in = outs[0] if len(outs) == 1 else torch.cat(outs, dim=2)
    y = layer_in
    if self.reverse:
        y = onn.reverse_sequence(y, seq_len)
    y = y.transpose(1, 2)
    if self.output_net is not None:
        y, seq_len = self.output_net(y, seq_len)
    return y, seq_len


def enable(model, on=True):
    """Switch every conv layer and GRU wrapper of an oracle model to the bf16-operand restatement (per instance)."""
    import types
    n = 0
    for m in model.modules():
        if isinstance(m, onn._ConvLayer):
            m.forward = types.MethodType(_conv_layer_forward, m) if on else types.MethodType(onn._ConvLayer.forward, m)
            n += 1
        elif isinstance(m, onn.GRU):
            m.forward = types.MethodType(_gru_forward, m) if on else types.MethodType(onn.GRU.forward, m)
            n += 1
    return n
