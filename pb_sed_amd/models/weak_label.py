"""FBCRNN: drop-in counterpart of ``pb_sed.models.weak_label.CRNN``
(reference pb_sed/models/weak_label/crnn.py:14-340) running on the HIP kernels.

Same constructor fields, same ``forward`` / ``review`` / ``tagging`` / ``boundaries_detection`` /
``sound_event_detection`` contracts.  In addition to the reference's ``inputs['stft']`` the model
accepts ``inputs['audio_data']`` ([B, N] waveform) so that the STFT runs on the GPU inside the fused
log-mel kernel (BASELINE.json north_star).
"""
import numpy as np
import torch

from .. import engine, ops
from ..modules import (CNN, GRU, SHALLOW, NormalizedLogMelExtractor, build_cnn, build_rnn, num_frames)
from .base import SoundEventModel


class _NetFunction(torch.autograd.Function):
    """features -> CNN -> fwd/bwd GRUs -> heads -> squashed scores, as one autograd node."""

    @staticmethod
    def forward(ctx, model, x, seq_host, seq_dev, training, *params):
        layers = engine.describe_stack([model.cnn.cnn_2d, model.cnn.cnn_1d])
        h, cnn_ctx = engine.stack_forward(layers, x, seq_dev, seq_host, training, model.conv_precision)
        wrappers = [model.rnn_fwd] + ([model.rnn_bwd] if model.rnn_bwd is not None else [])
        logits, rnn_ctx = engine.rnn_forward(wrappers, h, seq_dev, seq_host, training, model.conv_precision)
        if model.keep_logits:                       # pre-squash head outputs, for parity checks at the logit level
            model.last_logits = [l.detach().clone() for l in logits]
        ys = [ops.squash_fwd(l, model.minimum_score) for l in logits]
        ctx.state = (model, layers, cnn_ctx, wrappers, rnn_ctx, ys, seq_host, seq_dev)
        ctx.mark_non_differentiable(h)
        return (h, *ys)

    @staticmethod
    def backward(ctx, _dh, *dys):
        model, layers, cnn_ctx, wrappers, rnn_ctx, ys, seq_host, seq_dev = ctx.state
        ctx.state = None
        engine.flatten_parameters(model)
        dlogits = [ops.squash_bwd(y, dy, model.minimum_score) for y, dy in zip(ys, dys)]
        hook = getattr(model, '_grad_hook', None) or (lambda name: None)
        dh = engine.rnn_backward(wrappers, rnn_ctx, dlogits, seq_dev, seq_host)
        hook('rnn')
        n2d = len(model.cnn.cnn_2d.convs)
        engine.stack_backward(layers, cnn_ctx, dh, seq_dev, seq_host, need_input_grad=False,
                              on_layer_done=lambda j: hook('cnn_1d') if j == n2d else (hook('cnn_2d') if j == 0 else None))
        return (None,) * (5 + len(model._net_params))


class _HeadsFunction(torch.autograd.Function):
    """GRUs + heads only (windowed SED re-runs them on stacked windows of h; inference only)."""

    @staticmethod
    def forward(ctx, model, h, seq_host, seq_dev):
        wrappers = [model.rnn_fwd] + ([model.rnn_bwd] if model.rnn_bwd is not None else [])
        logits, _ = engine.rnn_forward(wrappers, h, seq_dev, seq_host, False, model.conv_precision)
        return tuple(ops.squash_fwd(l, model.minimum_score) for l in logits)

    @staticmethod
    def backward(ctx, *g):
        raise NotImplementedError('windowed SED is an inference method')


class _LossFunction(torch.autograd.Function):
    """Fused FBCRNN loss (pb_sed/models/weak_label/crnn.py:107-153,180-206) on squashed scores."""

    @staticmethod
    def forward(ctx, y_fwd, y_bwd, weak, bnd, seq_dev, cfg):
        loss, _, _, d_f, d_b = ops.fbcrnn_loss(
            y_fwd.contiguous(), None if y_bwd is None else y_bwd.contiguous(), weak, bnd, seq_dev,
            minimum_score=cfg['minimum_score'], strong_weight=cfg['strong_weight'], slat=cfg['slat'],
            label_smoothing=cfg['label_smoothing'], class_weights=cfg['class_weights'],
            inputs_are_scores=True, summary=cfg.get('summary'))
        ctx.grads = (d_f, d_b)
        return loss

    @staticmethod
    def backward(ctx, g):
        d_f, d_b = ctx.grads
        return d_f * g, (None if d_b is None else d_b * g), None, None, None, None


class CRNN(SoundEventModel):
    """
    >>> crnn = CRNN.build(num_events=10)           # reference 'shallow' config
    >>> outputs = crnn({'audio_data': wav, 'seq_len': [500] * 32, 'weak_targets': w, 'boundary_targets': b})
    >>> review = crnn.review(inputs, outputs); review['loss'].backward()
    """

    def __init__(self, feature_extractor, cnn, rnn_fwd, rnn_bwd, *, minimum_score=1e-5,
                 label_smoothing=0., labelwise_metrics=(), label_mapping=None, test_labels=None,
                 slat=False, strong_fwd_bwd_loss_weight=1., class_weights=None):
        super().__init__(labelwise_metrics=labelwise_metrics, label_mapping=label_mapping,
                         test_labels=test_labels)
        self.feature_extractor = feature_extractor
        self.cnn = cnn
        self.rnn_fwd = rnn_fwd
        self.rnn_bwd = rnn_bwd
        self.minimum_score = minimum_score
        self.label_smoothing = label_smoothing
        self.slat = slat
        self.strong_fwd_bwd_loss_weight = strong_fwd_bwd_loss_weight
        self.class_weights = None if class_weights is None else torch.Tensor(class_weights)

    @classmethod
    def build(cls, num_events=10, number_of_filters=128, stft_size=1024, sample_rate=16000,
              hidden_size=256, num_layers=2, net=None, rnn_bwd=True, feature_extractor=None, **kw):
        """Reference model factory (pb_sed/experiments/weak_label_crnn/training.py:158-260).  ``feature_extractor``:
        extra NormalizedLogMelExtractor fields, e.g. the training augmentation of training.py:209-216
        (dict(n_time_masks=1, n_frequency_masks=1, max_noise_scale=.2))."""
        net = dict(SHALLOW if net is None else net)
        fe = NormalizedLogMelExtractor(sample_rate, stft_size, number_of_filters, **(feature_extractor or {}))
        cnn = build_cnn(1, input_height=number_of_filters, **net)
        c = net['out_channels_1d'][-1]
        fwd = build_rnn(c, hidden_size, num_layers, num_events, hidden_size)
        bwd = build_rnn(c, hidden_size, num_layers, num_events, hidden_size, reverse=True) if rnn_bwd else None
        return cls(fe, cnn, fwd, bwd, **kw)

    @classmethod
    def finalize_dogmatic_config(cls, config):
        """Completes a (partial) model config the way the reference does (pb_sed/models/weak_label/crnn.py:304-340):
        default factories of the three sub-modules, CNN input channels / height from the feature extractor, GRU input
        width from the 1-D stack, and ``rnn_bwd`` = a copy of ``rnn_fwd`` with ``reverse=True`` unless it is None."""
        config['feature_extractor'] = {'factory': NormalizedLogMelExtractor}
        config['cnn'] = {'factory': CNN}
        config['rnn_fwd'] = {'factory': GRU}
        config['rnn_bwd'] = {}
        fe, cnn = config['feature_extractor'], config['cnn']
        cnn['cnn_2d']['in_channels'] = 1 + fe['add_deltas'] + fe['add_delta_deltas'] + cnn['positional_encoding']
        cnn['input_height'] = fe['number_of_filters']
        CNN.finalize_dogmatic_config(cnn)                 # the 1-D stack's width depends on the height set just now
        width = cnn['cnn_1d'].get('out_channels')
        if width is not None:
            config['rnn_fwd']['rnn']['input_size'] = width[-1]
        if config.get('rnn_bwd') is not None:
            config['rnn_bwd'].update(config['rnn_fwd'].to_dict(), reverse=True)

    # ------------------------------------------------------------------ forward / review
    def _net(self, x, seq_host, seq_dev):
        self._net_params = [p for p in self.parameters()]
        engine.flatten_parameters(self)
        training = self.training and torch.is_grad_enabled()   # batch statistics + saved context
        return _NetFunction.apply(self, x, seq_host, seq_dev, training, *self._net_params)

    def sigmoid(self, y):
        return self.minimum_score + (1 - 2 * self.minimum_score) * torch.sigmoid(y)

    def forward(self, inputs):
        key = self.input_key(inputs)
        x_in = inputs.pop(key) if self.training else inputs[key]
        seq_host, seq_dev = self._seq(inputs, x_in.device)
        x = self.features(inputs, x_in, seq_host, seq_dev)
        targets = self.read_targets(inputs) if 'weak_targets' in inputs else None
        h, y_fwd, *rest = self._net(x, seq_host, seq_dev)
        y_bwd = rest[0] if rest else None
        return y_fwd, y_bwd, seq_host, x, seq_host, targets

    def read_targets(self, inputs, subsample_idx=None):
        if 'boundary_targets' in inputs:
            return inputs['weak_targets'], inputs['boundary_targets']
        return inputs['weak_targets'],

    def modify_summary(self, summary):
        """Called by the trainer before dumping a summary (reference models/weak_label/crnn.py modify_summary): metrics
        from the validation buffers, then the base class' scalar means / image grid."""
        if 'targets_weak' in summary['buffers']:
            self.add_metrics_to_summary(summary, 'weak')
        return super().modify_summary(summary)

    def review(self, inputs, outputs, defer_summary=False):
        y_fwd, y_bwd, seq_len, x, _, targets = outputs
        assert targets is not None
        weak_targets = targets[0].to(torch.float32)
        seq_dev = engine.seq_to_device(seq_len, y_fwd.device)
        bnd = None
        if self.strong_fwd_bwd_loss_weight > 0. and not self.slat:
            assert len(targets) == 2, len(targets)
            bnd = targets[1].to(torch.float32)
        cfg = dict(minimum_score=self.minimum_score, strong_weight=self.strong_fwd_bwd_loss_weight,
                   slat=self.slat, label_smoothing=self.label_smoothing,
                   class_weights=None if self.class_weights is None else self.class_weights.to(y_fwd.device))
        # summary side (host): same buffers / scalars as the reference (crnn.py:122,137,155-177), written by the loss
        # launch itself and fetched with ONE device->host transfer instead of the reference's five separate .cpu() syncs
        b, k = weak_targets.shape
        packed_dev = torch.empty(3 * b * k + 1, device=y_fwd.device, dtype=torch.float32)
        cfg['summary'] = packed_dev
        loss = _LossFunction.apply(y_fwd, y_bwd, weak_targets, bnd, seq_dev, cfg)
        with torch.no_grad():
            host = torch.empty(packed_dev.shape, dtype=packed_dev.dtype, pin_memory=True)
            host.copy_(packed_dev, non_blocking=True)
            copied = torch.cuda.Event()
            copied.record()
        review = dict(loss=loss, scalars=dict(seq_len=np.mean(inputs['seq_len'])), images=dict(features=x[:3]), buffers={})

        def finalize():
            copied.synchronize()
            packed = host.numpy()
            wm = packed[:b * k].reshape(b, k) > .5
            w_host = packed[b * k:2 * b * k].reshape(b, k)
            y_weak_host = packed[2 * b * k:3 * b * k].reshape(b, k)
            labeled = wm.all(-1)
            review['scalars'].update(
                weak_label_rate=wm.mean(),
                boundary_label_rate=float(packed[-1]) if self.strong_fwd_bwd_loss_weight > 0. else 0.)
            review['buffers'].update(y_weak=y_weak_host[labeled], targets_weak=w_host[labeled])
            return review

        if defer_summary:
            # the caller (Trainer.step) enqueues backward + optimiser first and only then waits for the copy,
            # so the device never idles on the host between forward and backward
            review['_finalize'] = finalize
            return review
        return finalize()

    # ------------------------------------------------------------------ inference heads
    def tagging(self, inputs):
        y_fwd, y_bwd, seq_len_y, *_ = self.forward(inputs)
        seq_len = np.ones_like(seq_len_y)
        idx = engine.seq_to_device(seq_len_y, y_fwd.device).long() - 1
        last = y_fwd[torch.arange(y_fwd.shape[0], device=y_fwd.device), :, idx][..., None]
        if y_bwd is None:
            return last, seq_len
        return (last + y_bwd[..., :1]) / 2, seq_len

    def boundaries_detection(self, inputs):
        y_fwd, y_bwd, seq_len_y, *_ = self.forward(inputs)
        t = y_fwd.shape[-1]
        m = (torch.arange(t, device=y_fwd.device)[None] <
             engine.seq_to_device(seq_len_y, y_fwd.device)[:, None])[:, None, :]
        return torch.minimum(y_fwd * m, y_bwd * m), seq_len_y

    def sound_event_detection(self, inputs, window_length, window_shift=1):
        """Windowed SED (pb_sed/models/weak_label/crnn.py:241-302): for every window position the clip-level tagging
        score of the window (forward GRU's last + backward GRU's first output, halved).  ``window_length``: scalar, per
        class [K] or per variant and class [n, K] / [n, 1]; result [B, (n,) K, T'] with T' = ceil(T / window_shift).

        The CNN runs once on the whole clip; per distinct window length the windows of ALL positions and clips are
        gathered by one index op into a [n_windows * B, C, length] batch of short sequences and scanned together; the
        scores are then routed to the (variant, class) slots that asked for that length."""
        lengths = np.array(window_length, dtype=int)
        if lengths.ndim > 2:
            raise ValueError('window_length.ndim must not be greater than 2.')
        key = self.input_key(inputs)
        seq_host, seq_dev = self._seq(inputs, inputs[key].device)
        x = self.features(inputs, inputs[key], seq_host, seq_dev)
        with torch.no_grad():
            h = self._net(x, seq_host, seq_dev)[0]
        per_length = {int(n): self._window_scores(h, int(n), window_shift) for n in np.unique(lengths)}
        seq_len_y = 1 + (np.asarray(seq_host) - 1) // window_shift
        if lengths.ndim == 0:
            return per_length[int(lengths)], seq_len_y
        k = next(iter(per_length.values())).shape[1]
        assert lengths.shape[-1] in (1, k), lengths.shape
        slots = np.broadcast_to(lengths, lengths.shape[:-1] + (k,)) if lengths.ndim == 2 else lengths
        out = None
        for n, y in per_length.items():                       # y [B, K, T']
            chosen = torch.from_numpy(np.ascontiguousarray(slots == n)).to(y.device)[..., None]     # [(n,) K or 1, 1]
            part = chosen * (y[:, None] if lengths.ndim == 2 else y)
            out = part if out is None else out + part
        return out, seq_len_y

    def _window_scores(self, h, window_length, window_shift):
        """[B, C, T] -> [B, K, ceil(T / shift)]: tagging score of the window that the reference's padding rule centres
        on each shift position ('both' padding of length - shift, then shift - 1 at the end)."""
        b, c, t = h.shape
        lead = max(window_length - window_shift, 0) // 2
        starts = torch.arange(0, t, window_shift, device=h.device)                      # window n covers [s - lead, +length)
        idx = starts[:, None] - lead + torch.arange(window_length, device=h.device)[None]     # [n, length]
        inside = (idx >= 0) & (idx < t)
        hw = h[:, :, idx.clamp(0, t - 1)] * inside                                       # [B, C, n, length], zero padding
        n = starts.numel()
        hw = hw.permute(2, 0, 1, 3).reshape(n * b, c, window_length).contiguous()        # '(n b) c l'
        seq_host = np.full(n * b, window_length)
        seq_dev = torch.full((n * b,), window_length, dtype=torch.int32, device=h.device)
        with torch.no_grad():
            ys = _HeadsFunction.apply(self, hw, seq_host, seq_dev)
        y = ys[0][..., -1]
        if self.rnn_bwd is not None:
            y = (y + ys[1][..., 0]) / 2
        return y.reshape(n, b, -1).permute(1, 2, 0)


# ---- tuning wrappers (reference pb_sed/models/weak_label/crnn.py:343-421): one ensemble inference pass, then the leaderboard search
# of pb_sed_amd.tuning over the scores (its filters run on the device as well)
def tune_tagging(crnns, dataset, device, timestamps, event_classes, metrics, minimize=False, storage_dir=None):
    from .. import inference, tuning
    print('\nTagging Tuning')
    tagging_scores = inference.tagging(crnns, dataset, device, timestamps=timestamps, event_classes=event_classes)
    return tuning.tune_tagging(tagging_scores, medfilt_length_candidates=[1], metrics=metrics, minimize=minimize,
                               storage_dir=storage_dir, device=device)


def tune_boundary_detection(crnns, dataset, device, timestamps, event_classes, tags, metrics, stepfilt_lengths, minimize=False,
                            tag_masking='?', storage_dir=None):
    from .. import inference, tuning
    print('\nBoundaries Detection Tuning')
    boundaries_scores = inference.boundaries_detection(crnns, dataset, device, stepfilt_length=None, apply_mask=False, masks=tags,
                                                       timestamps=timestamps, event_classes=event_classes)
    return tuning.tune_boundaries_detection(boundaries_scores, medfilt_length_candidates=[1], stepfilt_length_candidates=stepfilt_lengths,
                                            tags=tags, metrics=metrics, minimize=minimize, tag_masking=tag_masking,
                                            storage_dir=storage_dir, device=device)


def tune_sound_event_detection(crnns, dataset, device, timestamps, event_classes, tags, metrics, window_lengths, window_shift,
                               medfilt_lengths, minimize=False, tag_masking='?', storage_dir=None):
    """One windowed-SED inference pass per window length, the median-filter / tag-masking search on each, and the best of all
    window lengths per class and metric (the winner's ``window_length`` / ``window_shift`` recorded with its hyper-parameters)."""
    from .. import inference, tuning
    print('\nSound Event Detection Tuning')
    leaderboard = {}
    for win_len in window_lengths:
        print(f'\n### window_length={win_len} ###')
        detection_scores = inference.sound_event_detection(
            crnns, dataset, device, model_kwargs={'window_length': win_len, 'window_shift': window_shift},
            timestamps=timestamps[::window_shift], event_classes=event_classes)
        for_winlen = tuning.tune_sound_event_detection(detection_scores, medfilt_lengths, tags, metrics=metrics, minimize=minimize,
                                                       tag_masking=tag_masking, storage_dir=storage_dir, device=device)
        for metric_name, (metric_values, hyper_params, scores) in for_winlen.items():
            for event_class in event_classes:
                hyper_params[event_class]['window_length'] = win_len
                hyper_params[event_class]['window_shift'] = window_shift
            leaderboard = tuning.update_leaderboard(leaderboard, metric_name, metric_values, hyper_params, scores, minimize=minimize)
    print('\nbest overall:')
    for metric_name in metrics:
        print(f'\n{metric_name} :\n{leaderboard[metric_name][0]}')
    return leaderboard
