"""Oracle: padertorch je-module semantics restated on stock torch (CPU).

TEST INFRASTRUCTURE ONLY (see oracle/__init__.py).  "Parity unpinned": padertorch@b7ba24a
(reference README.md:40) is not available; semantics follow SURVEY.md Appendix A.3-A.6 and the
reference's call sites:
* ``Normalization`` ('batch' norm, eps 1e-3) - pb_sed/experiments/weak_label_crnn/training.py:223-225
* ``CNN2d`` / ``CNN1d`` / hybrid ``CNN`` - pb_sed/models/weak_label/crnn.py:93,
  pb_sed/experiments/weak_label_crnn/training.py:159-169,218-242
* ``GRU`` wrapper - pb_sed/models/weak_label/crnn.py:61-67,338-340, training.py:243-260
* ``TakeLast/Mean/Sum/Max``, ``Pad`` - pb_sed/models/weak_label/crnn.py:147,158,227,288-290
"""
import math

import numpy as np
import torch
from torch import nn
import torch.nn.functional as F

from .frontend import compute_mask


# ----------------------------------------------------------------------------- reductions
class TakeLast:
    def __init__(self, axis=-1, keepdims=False):
        self.axis, self.keepdims = axis, keepdims

    def __call__(self, x, seq_len=None):
        axis = self.axis % x.dim()
        if seq_len is None:
            out = x.select(axis, x.shape[axis] - 1)
        else:
            idx = torch.as_tensor(np.asarray(seq_len) - 1, device=x.device, dtype=torch.long)
            xm = x.movedim(axis, 1)  # [B, T, ...]
            out = xm[torch.arange(x.shape[0], device=x.device), idx]
        return out.unsqueeze(axis) if self.keepdims else out


class Sum:
    def __init__(self, axis=-1, keepdims=False):
        self.axis, self.keepdims = axis, keepdims

    def __call__(self, x, seq_len=None):
        if seq_len is not None:
            x = x * compute_mask(x, seq_len, 0, self.axis)
        return x.sum(self.axis, keepdim=self.keepdims)


class Mean(Sum):
    def __call__(self, x, seq_len=None):
        if seq_len is None:
            return x.mean(self.axis, keepdim=self.keepdims)
        s = super().__call__(x, seq_len)
        n = torch.as_tensor(np.asarray(seq_len), device=x.device, dtype=x.dtype)
        shape = [1] * s.dim()
        shape[0] = -1
        return s / n.reshape(shape)


class Max:
    def __init__(self, axis=-1, keepdims=False):
        self.axis, self.keepdims = axis, keepdims

    def __call__(self, x, seq_len=None):
        if seq_len is not None:
            mask = compute_mask(x, seq_len, 0, self.axis)
            x = x * mask + (1 - mask) * torch.finfo(x.dtype).min
        return x.max(self.axis, keepdim=self.keepdims)


class Pad:
    """Pads the LAST axis with zeros: 'both' -> (size//2, ceil(size/2)), 'front', 'end'."""

    def __init__(self, side='both'):
        self.side = side

    def __call__(self, x, size):
        size = int(size)
        if self.side == 'both':
            p = (size // 2, int(math.ceil(size / 2)))
        elif self.side == 'front':
            p = (size, 0)
        elif self.side == 'end':
            p = (0, size)
        else:
            raise ValueError(self.side)
        return F.pad(x, p)


# ----------------------------------------------------------------------------- normalisation
class Normalization(nn.Module):
    """Per-channel masked batch norm over all axes but channel (data formats 'bcft', 'bct').

    Training: masked mean / biased variance over (b, [f,] t<seq_len); y = gamma*(x-mu)/sqrt(var+eps)
    + beta, re-masked.  Running statistics (mean, power) tracked with ``momentum``; eval uses them.
    For full-length batches this equals ``torch.nn.BatchNorm{1,2}d(eps)`` in training mode.
    """

    def __init__(self, num_channels, eps=1e-3, momentum=0.95):
        super().__init__()
        self.num_channels, self.eps, self.momentum = num_channels, eps, momentum
        self.gamma = nn.Parameter(torch.ones(num_channels))
        self.beta = nn.Parameter(torch.zeros(num_channels))
        self.register_buffer('running_mean', torch.zeros(num_channels))
        self.register_buffer('running_power', torch.ones(num_channels))
        self.freeze_stats = False

    def _bc(self, v, x):
        return v.reshape([1, -1] + [1] * (x.dim() - 2))

    def forward(self, x, seq_len=None):
        mask = compute_mask(x, seq_len)
        if self.training and not self.freeze_stats:
            axes = [0] + list(range(2, x.dim()))
            n = mask.sum(axes)
            xm = x * mask
            mean = xm.sum(axes) / n
            d = (x - self._bc(mean, x)) * mask
            var = (d * d).sum(axes) / n
            with torch.no_grad():
                m = self.momentum
                self.running_mean.mul_(m).add_((1 - m) * mean.detach())
                self.running_power.mul_(m).add_((1 - m) * (var + mean * mean).detach())
        else:
            mean = self.running_mean
            var = self.running_power - self.running_mean ** 2
        y = (x - self._bc(mean, x)) * self._bc(torch.rsqrt(var + self.eps), x)
        y = y * self._bc(self.gamma, x) + self._bc(self.beta, x)
        return y * mask


# ----------------------------------------------------------------------------- conv stacks
class _ConvLayer(nn.Module):
    """One CNN layer: [norm -> relu ->] pad('both') -> conv(+bias) [-> norm -> relu] [-> pool].

    ``pre`` selects pre-activation (norm/relu on the INPUT channels); ``post`` post-activation.
    """

    def __init__(self, ndim, cin, cout, k, pool=1, pre=False, post=False, eps=1e-3):
        super().__init__()
        self.ndim, self.k, self.pool, self.pre, self.post = ndim, k, pool, pre, post
        conv_cls = nn.Conv2d if ndim == 2 else nn.Conv1d
        self.conv = conv_cls(cin, cout, k)
        nn.init.xavier_uniform_(self.conv.weight)
        nn.init.zeros_(self.conv.bias)
        self.norm = Normalization(cin if pre else cout, eps=eps) if (pre or post) else None

    # Discrete decisions of a forward pass (tests only).  ``record = True`` keeps the layer's ReLU mask / (2,1)-pool row
    # selector of the last forward in ``recorded``; ``imposed = {'relu': bool mask, 'pool': 0/1 selector}`` REPLACES the
    # layer's own decisions by given ones: relu(v) -> v * mask, max-pool -> the selected row.  With the decisions of another
    # implementation's run imposed, a float64 pass differentiates the same piecewise-smooth branch of the network that run
    # took, so its gradient can be compared at rounding level (a pre-activation within rounding of zero or a pool-window tie
    # otherwise switches a whole position's contribution on or off, see tests/test_gpu_configs.py::_impose_decisions).
    record = False
    imposed = None

    def _relu(self, v):
        imp = None if self.imposed is None else self.imposed.get('relu')
        if imp is not None:
            return v * imp.reshape(v.shape).to(v.dtype)
        if self.record:
            self.recorded = dict(getattr(self, 'recorded', None) or {}, relu=(v > 0).detach())
        return F.relu(v)

    def _pool(self, y):
        imp = None if self.imposed is None else self.imposed.get('pool')
        if imp is None and not self.record:
            return F.max_pool2d(y, self.pool) if self.ndim == 2 else F.max_pool1d(y, self.pool)
        assert self.ndim == 2 and tuple(self.pool) == (2, 1), 'decisions are recorded / imposed for (2,1) pools'
        fo = y.shape[2] // 2
        lo, hi = y[:, :, 0:2 * fo:2], y[:, :, 1:2 * fo:2]
        if imp is None:
            sel = (hi > lo).detach()                      # ties: the first row, as max_pool2d
            self.recorded = dict(getattr(self, 'recorded', None) or {}, pool=sel)
        else:
            sel = imp.reshape(lo.shape).bool()
        return torch.where(sel, hi, lo)

    def forward(self, x, seq_len=None):
        if self.pre:
            x = self._relu(self.norm(x, seq_len))
        p = self.k - 1
        lo, hi = p // 2, int(math.ceil(p / 2))
        x = F.pad(x, (lo, hi, lo, hi) if self.ndim == 2 else (lo, hi))
        y = self.conv(x)
        if self.post:
            y = self._relu(self.norm(y, seq_len))
        if self.pool != 1:
            y = self._pool(y)
        return y


class _CNN(nn.Module):
    ndim = None

    def __init__(self, in_channels, out_channels, kernel_size, pool_size=1, norm='batch',
                 eps=1e-3, pre_activation=False, output_layer=True, input_layer=True, residual_connections=None,
                 final_norm=False):
        """``final_norm``: SURVEY.md A.4 reading (iii) - a pre-activation stack closes with a norm + ReLU behind its last conv.
        ``residual_connections[i] = j``: the input of layer i is added to the input of layer j (the reference's 'deep'
        net_config, pb_sed/experiments/weak_label_crnn/training.py:170-183).  padertorch's skip path restated - parity
        unpinned: in between lying max-pools are applied to the skip, then a 1x1 conv (with bias) if the channel counts
        differ; with pre-activation layers both ends are the raw tensors in front of norm + ReLU."""
        super().__init__()
        n = len(out_channels)
        ks = kernel_size if isinstance(kernel_size, (list, tuple)) else n * [kernel_size]
        ps = pool_size if isinstance(pool_size, list) and len(pool_size) == n else n * [pool_size]
        self.in_channels, self.out_channels = in_channels, list(out_channels)
        convs = []
        cin = in_channels
        for i, cout in enumerate(out_channels):
            if pre_activation:
                pre = norm is not None and not (i == 0 and input_layer)
                post = False
            else:
                pre = False
                post = norm is not None and not (i == n - 1 and output_layer)
            convs.append(_ConvLayer(self.ndim, cin, cout, ks[i], ps[i], pre, post, eps))
            cin = cout
        self.convs = nn.ModuleList(convs)
        self.out_norm = Normalization(cin, eps=eps) if final_norm else None
        res = list(residual_connections) if residual_connections is not None else n * [None]
        self.residual_connections = [r[0] if isinstance(r, (list, tuple)) else r for r in res]
        cins = [in_channels] + list(out_channels[:-1])
        self.skip_convs = nn.ModuleDict()
        for src, dst in enumerate(self.residual_connections):
            if dst is not None and cins[src] != cins[dst]:
                conv = (nn.Conv2d if self.ndim == 2 else nn.Conv1d)(cins[src], cins[dst], 1)
                nn.init.xavier_uniform_(conv.weight)
                nn.init.zeros_(conv.bias)
                self.skip_convs[f'{src}_{dst}'] = conv

    def _skip_pool(self, r, pool, key):
        """Max-pool on a skip path; decisions recorded / imposed per (src, dst, crossed layer) like _ConvLayer's."""
        imp = (getattr(self, 'imposed_skip_pool', None) or {}).get(key)
        if imp is None and not getattr(self, 'record', False):
            return F.max_pool2d(r, pool) if self.ndim == 2 else F.max_pool1d(r, pool)
        assert self.ndim == 2 and tuple(pool) == (2, 1)
        fo = r.shape[2] // 2
        lo, hi = r[:, :, 0:2 * fo:2], r[:, :, 1:2 * fo:2]
        if imp is None:
            sel = (hi > lo).detach()
            if not hasattr(self, 'recorded_skip_pool'):
                self.recorded_skip_pool = {}
            self.recorded_skip_pool[key] = sel
        else:
            sel = imp.reshape(lo.shape).bool()
        return torch.where(sel, hi, lo)

    def forward(self, x, seq_len=None):
        inputs = []
        for j, conv in enumerate(self.convs):
            for src, dst in enumerate(self.residual_connections):
                if dst == j:
                    r = inputs[src]
                    for k in range(src, dst):
                        if self.convs[k].pool != 1:
                            r = self._skip_pool(r, self.convs[k].pool, (src, dst, k))
                    key = f'{src}_{dst}'
                    if key in self.skip_convs:
                        r = self.skip_convs[key](r)
                    x = x + r
            inputs.append(x)
            x = conv(x, seq_len)
        if self.out_norm is not None:
            v = self.out_norm(x, seq_len)
            imp = getattr(self, 'imposed_out_relu', None)
            if imp is not None:                           # see _ConvLayer.imposed
                x = v * imp.reshape(v.shape).to(v.dtype)
            else:
                if getattr(self, 'record', False):
                    self.recorded_out_relu = (v > 0).detach()
                x = F.relu(v)
        return x, seq_len

    def freeze(self, num_layers=None, freeze_norm_stats=True):
        layers = self.convs if num_layers is None else self.convs[:num_layers]
        for layer in layers:
            for p in layer.parameters():
                p.requires_grad = False
            if freeze_norm_stats and layer.norm is not None:
                layer.norm.freeze_stats = True


class CNN2d(_CNN):
    ndim = 2


class CNN1d(_CNN):
    ndim = 1


class CNN(nn.Module):
    """Hybrid CNN: [cond concat ->] cnn_2d -> 'b c f t -> b (c f) t' -> cnn_1d."""

    def __init__(self, cnn_2d, cnn_1d, input_height=128, conditional_dims=0):
        super().__init__()
        self.cnn_2d, self.cnn_1d = cnn_2d, cnn_1d
        self.input_height, self.conditional_dims = input_height, conditional_dims

    def forward(self, x, seq_len=None, condition=None):
        if condition is not None:
            b, _, f, t = x.shape
            cond = condition.reshape(b, -1, 1, 1).to(x.dtype).expand(b, condition.shape[1], f, t)
            x = torch.cat([x, cond], dim=1)
        x, seq_len = self.cnn_2d(x, seq_len)
        x = x.flatten(1, 2)
        return self.cnn_1d(x, seq_len)


# ----------------------------------------------------------------------------- GRU wrapper
def reverse_sequence(x, seq_len):
    """x [B,T,F]; reverse each sequence inside its own length (padding stays at the end)."""
    if seq_len is None:
        return x.flip(1)
    b, t, _ = x.shape
    sl = torch.as_tensor(np.asarray(seq_len), device=x.device)[:, None]
    ar = torch.arange(t, device=x.device)[None]
    idx = torch.where(ar < sl, sl - 1 - ar, ar)
    return torch.gather(x, 1, idx[..., None].expand_as(x))


class GRU(nn.Module):
    def __init__(self, input_size, hidden_size, num_layers=1, bidirectional=False, reverse=False,
                 output_net=None):
        super().__init__()
        self.rnn = nn.GRU(input_size, hidden_size, num_layers, bias=True, batch_first=True,
                          dropout=0., bidirectional=bidirectional)
        self.output_net = output_net
        self.reverse = reverse

    def forward(self, x, seq_len=None):
        x = x.transpose(1, 2)  # b t f
        t = x.shape[1]
        if self.reverse:
            x = reverse_sequence(x, seq_len)
        if seq_len is not None:
            packed = nn.utils.rnn.pack_padded_sequence(
                x, torch.as_tensor(np.asarray(seq_len)).cpu(), batch_first=True)
            y, _ = self.rnn(packed)
            y, _ = nn.utils.rnn.pad_packed_sequence(y, batch_first=True, total_length=t)
        else:
            y, _ = self.rnn(x)
        if self.reverse:
            y = reverse_sequence(y, seq_len)
        y = y.transpose(1, 2)
        if self.output_net is not None:
            y, seq_len = self.output_net(y, seq_len)
        return y, seq_len
