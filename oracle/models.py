"""Oracle: FBCRNN / BiCRNN forward, losses and inference heads restated (CPU, stock torch).

TEST INFRASTRUCTURE ONLY (see oracle/__init__.py).  The maths below is reference-OWNED and is
pinned by ``tests/golden/ref_*.npz`` (reference source executed under shims, tests/golden/gen_golden.py):
* FBCRNN: pb_sed/models/weak_label/crnn.py:58-100 (forward), :107-206 (losses), :223-302 (heads)
* BiCRNN: pb_sed/models/strong_label/crnn.py:60-93 (forward), :106-112 (loss), :200-210 (heads)
Architecture constants: pb_sed/experiments/weak_label_crnn/training.py:158-260,
pb_sed/experiments/strong_label_crnn/training.py:159-263.
"""
import numpy as np
import torch
from torch import nn
import torch.nn.functional as F

from .frontend import LogMelExtractor, compute_mask
from .nn import CNN, CNN1d, CNN2d, GRU, Mean, Pad, Sum, TakeLast


def bce(p, t):
    """nn.BCELoss(reduction='none') (logs clamped at -100)."""
    return F.binary_cross_entropy(p, t, reduction='none')


# ------------------------------------------------------------------ reference-owned maths
def squash(y, minimum_score=1e-5):
    """weak_label/crnn.py:58-59"""
    return minimum_score + (1 - 2 * minimum_score) * torch.sigmoid(y)


def fbcrnn_loss(y_fwd, y_bwd, seq_len, weak_targets, boundary_targets=None, *,
                strong_fwd_bwd_loss_weight=1., slat=False, label_smoothing=0.,
                class_weights=None, imposed_sel=None):
    """weak_label/crnn.py:107-153 + :180-206.  Returns (loss, weak_mask, boundary_mask).
    ``imposed_sel`` (tests only, see oracle/nn.py::_ConvLayer.imposed): bool [B,K,T], True where max(y_fwd, y_bwd) is to
    take y_fwd - the selector of another implementation's run instead of this one's."""
    w_mask = (weak_targets < .01) | (weak_targets > .99)
    w = weak_targets * w_mask
    wt = torch.clip(w, label_smoothing, 1 - label_smoothing) if label_smoothing > 0 else w
    if y_bwd is None:
        y_last = TakeLast(axis=2)(y_fwd, seq_len)
        loss = bce(y_last, wt)[..., None].expand(y_fwd.shape)
    else:
        y_max = torch.maximum(y_fwd, y_bwd) if imposed_sel is None else torch.where(imposed_sel, y_fwd, y_bwd)
        loss = bce(y_max, wt[..., None].expand(y_fwd.shape))
    loss = loss * w_mask[..., None]
    b_mask = None
    if strong_fwd_bwd_loss_weight > 0.:
        beta = w[..., None].expand(y_fwd.shape) if slat else boundary_targets
        b_mask = (beta > .99) | (beta < .01)
        b_mask = b_mask * (b_mask.float().mean(-1, keepdim=True) > .999) * (w > .99)[..., None]
        if (b_mask == 1).any():
            bt = torch.clip(beta, label_smoothing, 1 - label_smoothing) if label_smoothing > 0 \
                else beta
            t_fwd = torch.cummax(bt, dim=-1)[0]
            t_bwd = torch.cummax(bt.flip(-1), dim=-1)[0].flip(-1)
            strong = bce(y_fwd, t_fwd)
            if y_bwd is not None:
                strong = strong / 2 + bce(y_bwd, t_bwd) / 2
            lam = b_mask * strong_fwd_bwd_loss_weight
            loss = lam * strong + (1. - lam) * loss
    loss = Mean(axis=-1)(loss, seq_len)
    weights = w_mask if class_weights is None else w_mask * torch.as_tensor(class_weights)
    loss = (loss * weights).sum() / weights.sum()
    return loss, w_mask, b_mask


def bicrnn_loss(y, seq_len, strong_targets):
    """strong_label/crnn.py:106-112 (numerator seq-masked, denominator not)."""
    mask = (strong_targets > .99) | (strong_targets < .01)
    l = bce(y, strong_targets) * mask
    return Sum(axis=-1)(l, seq_len).sum() / mask.sum()


# ------------------------------------------------------------------ model builders
def _feature_extractor(cfg):
    return LogMelExtractor(**cfg)


def build_cnn(in_channels, out_channels_2d, pool_sizes_2d, kernel_size_2d, out_channels_1d,
              kernel_size_1d, input_height, conditional_dims=0, eps=1e-3, residual_connections_2d=None,
              residual_connections_1d=None, input_layer_2d=True, input_layer_1d=False, final_norm_1d=False):
    cnn_2d = CNN2d(in_channels + conditional_dims, out_channels_2d, kernel_size_2d, pool_sizes_2d,
                   eps=eps, pre_activation=True, output_layer=False, input_layer=input_layer_2d,
                   residual_connections=residual_connections_2d)
    f = input_height
    for p in (pool_sizes_2d if isinstance(pool_sizes_2d, list) else len(out_channels_2d) * [pool_sizes_2d]):
        f //= (p[0] if isinstance(p, (tuple, list)) else p)
    cnn_1d = CNN1d(out_channels_2d[-1] * f, out_channels_1d, kernel_size_1d, 1, eps=eps,
                   pre_activation=True, output_layer=False, input_layer=input_layer_1d,
                   residual_connections=residual_connections_1d, final_norm=final_norm_1d)
    return CNN(cnn_2d, cnn_1d, input_height, conditional_dims)


def build_rnn(input_size, hidden_size, num_layers, num_events, head_hidden, bidirectional=False,
              reverse=False, eps=1e-3):
    dirs = 2 if bidirectional else 1
    output_net = CNN1d(hidden_size * dirs, [head_hidden, num_events], 1, 1, eps=eps,
                       pre_activation=False, output_layer=True)
    return GRU(input_size, hidden_size, num_layers, bidirectional, reverse, output_net)


SHALLOW = dict(
    out_channels_2d=[16, 16, 32, 32, 64, 64, 128, 128, 256],
    pool_sizes_2d=4 * [1, (2, 1)] + [1],
    kernel_size_2d=3,
    out_channels_1d=5 * [256],
    kernel_size_1d=[1, 3, 3, 3, 1],
)


class FBCRNN(nn.Module):
    """weak_label.CRNN restated.  forward(inputs) -> (y_fwd, y_bwd, seq_len_y, x, seq_len_x, targets)."""

    def __init__(self, feature_extractor, cnn, rnn_fwd, rnn_bwd, *, minimum_score=1e-5,
                 label_smoothing=0., slat=False, strong_fwd_bwd_loss_weight=1., class_weights=None):
        super().__init__()
        self.feature_extractor, self.cnn = feature_extractor, cnn
        self.rnn_fwd, self.rnn_bwd = rnn_fwd, rnn_bwd
        self.minimum_score, self.label_smoothing, self.slat = minimum_score, label_smoothing, slat
        self.strong_fwd_bwd_loss_weight, self.class_weights = strong_fwd_bwd_loss_weight, class_weights

    @classmethod
    def build(cls, num_events=10, number_of_filters=128, stft_size=1024, sample_rate=16000,
              hidden_size=256, num_layers=2, net=None, rnn_bwd=True, **kw):
        net = dict(SHALLOW if net is None else net)
        fe = LogMelExtractor(sample_rate, stft_size, number_of_filters)
        cnn = build_cnn(1, input_height=number_of_filters, **net)
        c = net['out_channels_1d'][-1]
        fwd = build_rnn(c, hidden_size, num_layers, num_events, hidden_size)
        bwd = build_rnn(c, hidden_size, num_layers, num_events, hidden_size, reverse=True) \
            if rnn_bwd else None
        return cls(fe, cnn, fwd, bwd, **kw)

    def sigmoid(self, y):
        return squash(y, self.minimum_score)

    def fwd_tagging(self, h, seq_len):
        y, sl = self.rnn_fwd(h, seq_len)
        return self.sigmoid(y), sl

    def bwd_tagging(self, h, seq_len):
        y, sl = self.rnn_bwd(h, seq_len)
        return self.sigmoid(y), sl

    def encode(self, inputs):
        x = inputs['stft']
        seq_len = np.array(inputs['seq_len'])
        x, seq_len_x = self.feature_extractor(x, seq_len=seq_len)
        h, seq_len_h = self.cnn(x, seq_len_x)
        return x, seq_len_x, h, seq_len_h

    def forward(self, inputs):
        x, seq_len_x, h, seq_len_h = self.encode(inputs)
        targets = None
        if 'weak_targets' in inputs:
            targets = (inputs['weak_targets'], inputs['boundary_targets']) \
                if 'boundary_targets' in inputs else (inputs['weak_targets'],)
        y_fwd, seq_len_y = self.fwd_tagging(h, seq_len_h)
        y_bwd = None if self.rnn_bwd is None else self.bwd_tagging(h, seq_len_h)[0]
        return y_fwd, y_bwd, seq_len_y, x, seq_len_x, targets

    def review(self, inputs, outputs):
        y_fwd, y_bwd, seq_len, x, _, targets = outputs
        loss, w_mask, b_mask = fbcrnn_loss(
            y_fwd, y_bwd, seq_len, targets[0], targets[1] if len(targets) > 1 else None,
            strong_fwd_bwd_loss_weight=self.strong_fwd_bwd_loss_weight, slat=self.slat,
            label_smoothing=self.label_smoothing, class_weights=self.class_weights,
            imposed_sel=getattr(self, 'imposed_sel', None))
        labeled = (w_mask.numpy() == 1).all(-1)
        y_weak = TakeLast(axis=2)(y_fwd, seq_len)
        if y_bwd is not None:
            y_weak = y_weak / 2 + y_bwd[..., 0] / 2
        return dict(
            loss=loss,
            scalars=dict(seq_len=np.mean(inputs['seq_len']),
                         weak_label_rate=w_mask.numpy().mean(),
                         boundary_label_rate=0. if b_mask is None else b_mask.numpy().mean()),
            images=dict(features=x[:3]),
            buffers=dict(y_weak=y_weak.detach().numpy()[labeled],
                         targets_weak=(targets[0] * w_mask).numpy()[labeled]),
        )

    # ---- inference heads (weak_label/crnn.py:223-302)
    def tagging(self, inputs):
        y_fwd, y_bwd, seq_len_y, *_ = self.forward(inputs)
        seq_len = np.ones_like(seq_len_y)
        last = TakeLast(axis=-1, keepdims=True)(y_fwd, seq_len_y)
        if y_bwd is None:
            return last, seq_len
        return (last + y_bwd[..., :1]) / 2, seq_len

    def boundaries_detection(self, inputs):
        y_fwd, y_bwd, seq_len_y, *_ = self.forward(inputs)
        m = compute_mask(y_fwd, seq_len_y)
        return torch.minimum(y_fwd * m, y_bwd * m), seq_len_y

    def sound_event_detection(self, inputs, window_length, window_shift=1):
        window_length = np.array(window_length, dtype=int)
        _, _, h, seq_len = self.encode(inputs)
        if window_length.ndim == 0:
            return self._single_window_length_sed(h, seq_len, int(window_length), window_shift)
        y = None
        for win_len in np.unique(window_length.flatten()):
            yi, seq_len_y = self._single_window_length_sed(h, seq_len, int(win_len), window_shift)
            b, k, t = yi.shape
            if window_length.ndim == 2:
                window_length = np.broadcast_to(window_length, (window_length.shape[0], k))
                yi = yi[:, None]
            if y is None:
                y = torch.zeros((b, *window_length.shape, t))
            y += (torch.from_numpy(window_length.copy()) == win_len)[..., None] * yi
        return y, seq_len_y

    def _single_window_length_sed(self, h, seq_len, window_length, window_shift):
        b, f, t = h.shape
        if window_length > window_shift:
            h = Pad('both')(h, window_length - window_shift)
        h = Pad('end')(h, window_shift - 1)
        wins = [h[..., i:i + window_length] for i in np.arange(0, t, window_shift)]
        n = len(wins)
        hw = torch.cat(wins, dim=0)
        y, _ = self.fwd_tagging(hw, None)
        y = y[..., -1].reshape(n, b, -1).permute(1, 2, 0)
        if self.rnn_bwd is not None:
            yb, _ = self.bwd_tagging(hw, None)
            y = (y + yb[..., 0].reshape(n, b, -1).permute(1, 2, 0)) / 2
        return y, 1 + (seq_len - 1) // window_shift


class BiCRNN(nn.Module):
    """strong_label.CRNN restated.  forward -> (y, seq_len_y, x, seq_len_x, targets)."""

    def __init__(self, feature_extractor, cnn, rnn, *, tag_conditioning=False):
        super().__init__()
        self.feature_extractor, self.cnn, self.rnn = feature_extractor, cnn, rnn
        self.tag_conditioning = tag_conditioning

    @classmethod
    def build(cls, num_events=10, number_of_filters=128, stft_size=1024, sample_rate=16000,
              hidden_size=256, num_layers=2, net=None, tag_conditioning=True):
        net = dict(SHALLOW if net is None else net)
        fe = LogMelExtractor(sample_rate, stft_size, number_of_filters)
        cd = num_events if tag_conditioning else 0
        cnn = build_cnn(1, input_height=number_of_filters, conditional_dims=cd, **net)
        rnn = build_rnn(net['out_channels_1d'][-1] + cd, hidden_size, num_layers, num_events,
                        hidden_size, bidirectional=True)
        return cls(fe, cnn, rnn, tag_conditioning=tag_conditioning)

    def forward(self, inputs):
        x = inputs['stft']
        seq_len = np.array(inputs['seq_len'])
        x, seq_len_x = self.feature_extractor(x, seq_len=seq_len)
        targets = (inputs['weak_targets'], inputs['strong_targets']) \
            if 'strong_targets' in inputs else None
        tag = inputs['tag_condition'].unsqueeze(-1) if self.tag_conditioning else None
        h, seq_len_h = self.cnn(x, seq_len_x, tag if self.cnn.conditional_dims else None)
        if self.tag_conditioning:
            b, f, t = h.shape
            h = torch.cat([h, tag.to(h.dtype).expand(b, tag.shape[1], t)], dim=1)
        y, seq_len_y = self.rnn(h, seq_len_h)
        return torch.sigmoid(y), seq_len_y, x, seq_len_x, targets

    def review(self, inputs, outputs):
        y, seq_len_y, x, _, targets = outputs
        return dict(loss=bicrnn_loss(y, seq_len_y, targets[1]))

    def tagging(self, inputs):
        # strong_label/crnn.py:200-202: Max(-1, keepdims=True)(y)[0] - no seq_len, plain max over T
        y, seq_len_y, *_ = self.forward(inputs)
        return y.max(-1, keepdim=True)[0], np.ones_like(seq_len_y)

    def sound_event_detection(self, inputs):
        y, seq_len_y, *_ = self.forward(inputs)
        return y * compute_mask(y, seq_len_y), seq_len_y
