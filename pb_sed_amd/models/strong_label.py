"""BiCRNN: drop-in counterpart of ``pb_sed.models.strong_label.CRNN``
(reference pb_sed/models/strong_label/crnn.py:13-210), optionally tag-conditioned."""
import numpy as np
import torch

from .. import engine, ops
from ..modules import CNN, GRU, SHALLOW, NormalizedLogMelExtractor, build_cnn, build_rnn, num_frames
from .base import SoundEventModel


def _cnn_forward(model, x, tag, seq_host, seq_dev, training):
    """Features -> CNN output, as the scans take it: (h [B,C,T] or None, h_tbc [T,B,C+K(+pad)] or None, layers, ctx, C)."""
    if tag is not None and model.cnn.conditional_dims:
        b, _, f, t = x.shape
        x = torch.cat([x, tag.reshape(b, -1, 1, 1).to(x.dtype).expand(b, tag.shape[1], f, t)], dim=1).contiguous()
    layers = engine.describe_stack([model.cnn.cnn_2d, model.cnn.cnn_1d])
    h, cnn_ctx = engine.stack_forward(layers, x, seq_dev, seq_host, training, model.conv_precision)
    n_h = h.shape[1]
    h_tbc = None
    if tag is not None:
        # the GRU input in the scans' layout [T, B, C + K (+ zero channels up to whole float4s)]: the tags are appended
        # there, so the first layer's projections run time-major like the others (the padded width is the engine's affair)
        b, _, t = h.shape
        k = tag.shape[1]
        hc = ops.bct_to_tbc(h)
        parts = [hc, tag.to(h.dtype).reshape(1, b, k).expand(t, b, k)]
        if (n_h + k) % 4:
            parts.append(h.new_zeros((t, b, 4 - (n_h + k) % 4)))
        h_tbc = torch.cat(parts, dim=2)
        h = None
    return h, h_tbc, layers, cnn_ctx, n_h


class _NetFunction(torch.autograd.Function):
    @staticmethod
    def forward(ctx, model, x, tag, seq_host, seq_dev, training, *params):
        h, h_tbc, layers, cnn_ctx, n_h = _cnn_forward(model, x, tag, seq_host, seq_dev, training)
        logits, rnn_ctx = engine.rnn_forward([model.rnn], h, seq_dev, seq_host, training, model.conv_precision, h_tbc=h_tbc)
        if model.keep_logits:
            model.last_logits = [logits[0].detach().clone()]
        y = ops.squash_fwd(logits[0], 0.)
        ctx.state = (model, layers, cnn_ctx, rnn_ctx, y, n_h, seq_host, seq_dev)
        return y

    @staticmethod
    def backward(ctx, dy):
        model, layers, cnn_ctx, rnn_ctx, y, n_h, seq_host, seq_dev = ctx.state
        ctx.state = None
        engine.flatten_parameters(model)
        hook = getattr(model, '_grad_hook', None) or (lambda name: None)
        dh = engine.rnn_backward([model.rnn], rnn_ctx, [ops.squash_bwd(y, dy, 0.)], seq_dev, seq_host)
        hook('rnn')
        n2d = len(model.cnn.cnn_2d.convs)
        engine.stack_backward(layers, cnn_ctx, dh[:, :n_h].contiguous(), seq_dev, seq_host, need_input_grad=False,
                              on_layer_done=lambda j: hook('cnn_1d') if j == n2d else (hook('cnn_2d') if j == 0 else None))
        return (None,) * (6 + len(model._net_params))


class _LossFunction(torch.autograd.Function):
    """strong_label/crnn.py:106-112 fused (numerator seq-masked, denominator over all frames)."""

    @staticmethod
    def forward(ctx, y, strong_targets, seq_dev):
        loss, _, d = ops.bicrnn_loss(y.contiguous(), strong_targets, seq_dev, inputs_are_scores=True)
        ctx.d = d
        return loss

    @staticmethod
    def backward(ctx, g):
        return ctx.d * g, None, None


class CRNN(SoundEventModel):
    def __init__(self, feature_extractor, cnn, rnn, *, tag_conditioning=False, labelwise_metrics=(),
                 label_mapping=None, eval_segment_length=1):
        super().__init__(labelwise_metrics=labelwise_metrics, label_mapping=label_mapping)
        self.feature_extractor = feature_extractor
        self.cnn = cnn
        self.rnn = rnn
        self.tag_conditioning = tag_conditioning
        self.eval_segment_length = eval_segment_length

    @classmethod
    def build(cls, num_events=10, number_of_filters=128, stft_size=1024, sample_rate=16000,
              hidden_size=256, num_layers=2, net=None, tag_conditioning=True, **kw):
        """Reference factory (pb_sed/experiments/strong_label_crnn/training.py:159-263,
        pb_sed/models/strong_label/crnn.py:155-197)."""
        net = dict(SHALLOW if net is None else net)
        fe = NormalizedLogMelExtractor(sample_rate, stft_size, number_of_filters)
        cd = num_events if tag_conditioning else 0
        cnn = build_cnn(1, input_height=number_of_filters, conditional_dims=cd, **net)
        rnn = build_rnn(net['out_channels_1d'][-1] + cd, hidden_size, num_layers, num_events, hidden_size,
                        bidirectional=True)
        return cls(fe, cnn, rnn, tag_conditioning=tag_conditioning, **kw)

    @classmethod
    def finalize_dogmatic_config(cls, config):
        """pb_sed/models/strong_label/crnn.py:155-198: sub-module factories, tag conditioning adds ``num_events``
        constant input planes to the CNN and as many inputs to the GRU, which defaults to ONE bidirectional layer
        (the experiment config asks for two; the caller's entry wins)."""
        config['feature_extractor'] = {'factory': NormalizedLogMelExtractor}
        config['cnn'] = {'factory': CNN}
        config['rnn'] = {'factory': GRU}
        fe, cnn, rnn = config['feature_extractor'], config['cnn'], config['rnn']
        out_channels = rnn['output_net'].get('out_channels')
        num_events = out_channels[-1] if out_channels else None
        in_channels = 1 + fe['add_deltas'] + fe['add_delta_deltas'] + cnn['positional_encoding']
        if config['tag_conditioning'] and num_events is not None:
            cnn['conditional_dims'] = num_events
            in_channels += num_events
        cnn['cnn_2d']['in_channels'] = in_channels
        cnn['input_height'] = fe['number_of_filters']
        CNN.finalize_dogmatic_config(cnn)
        width = cnn['cnn_1d'].get('out_channels')
        rnn['rnn'].update({'num_layers': 1, 'bias': True, 'dropout': 0., 'bidirectional': True})
        if width is not None:
            rnn['rnn']['input_size'] = width[-1] + (num_events if config['tag_conditioning'] and num_events else 0)
        GRU.finalize_dogmatic_config(rnn)                 # head width follows bidirectional=True

    def forward(self, inputs):
        key = self.input_key(inputs)
        x_in = inputs.pop(key) if self.training else inputs[key]
        seq_host, seq_dev = self._seq(inputs, x_in.device)
        x = self.features(inputs, x_in, seq_host, seq_dev)
        targets = (inputs['weak_targets'], inputs['strong_targets']) if 'strong_targets' in inputs else None
        tag = inputs['tag_condition'].to(torch.float32) if self.tag_conditioning else None
        self._net_params = [p for p in self.parameters()]
        engine.flatten_parameters(self)
        training = self.training and torch.is_grad_enabled()
        y = _NetFunction.apply(self, x, tag, seq_host, seq_dev, training, *self._net_params)
        return y, seq_host, x, seq_host, targets

    MAX_JOINT = 3           # networks per joint scan launch (two directions each: pbsed.h's limit of six chains)

    @classmethod
    def can_run_jointly(cls, models):
        """Inference of several networks of one shape on one batch can share its recurrent scans (``forward_jointly``)."""
        m0 = models[0]
        return len(models) > 1 and all(
            type(m) is cls and not m.training and m.conv_precision == m0.conv_precision
            and m.tag_conditioning == m0.tag_conditioning and m.rnn.hidden_size == m0.rnn.hidden_size
            and m.rnn.hidden_size in (64, 128, 256, 512)
            and m.rnn.num_layers == m0.rnn.num_layers and m.rnn.bidirectional == m0.rnn.bidirectional
            and m.rnn.rnn.input_size == m0.rnn.rnn.input_size for m in models)

    @classmethod
    def forward_jointly(cls, models, inputs):
        """Scores [B,K,T] of every network for one batch (no gradients): each network's own features and CNN, then ONE
        persistent scan launch per GRU layer for up to MAX_JOINT networks - the scans are latency-bound chains of T
        dependent steps that leave most of the device idle, and an ensemble's networks are independent
        (reference pb_sed/models/base/inference.py:133-141 runs them one after the other).  Same arithmetic per network as
        ``forward``: the joint launch only adds chains to the scan's grid."""
        assert cls.can_run_jointly(models)
        m0 = models[0]
        key = m0.input_key(inputs)
        x_in = inputs[key]
        seq_host, seq_dev = m0._seq(inputs, x_in.device)
        tag = inputs['tag_condition'].to(torch.float32) if m0.tag_conditioning else None
        ys = []
        with torch.no_grad():
            for g0 in range(0, len(models), cls.MAX_JOINT):
                group = models[g0:g0 + cls.MAX_JOINT]
                hs, h_tbcs = [], []
                for m in group:
                    engine.flatten_parameters(m)
                    x = m.features(inputs, x_in, seq_host, seq_dev)
                    h, h_tbc, *_ = _cnn_forward(m, x, tag, seq_host, seq_dev, False)
                    if h_tbc is None:
                        h_tbc = ops.bct_to_tbc(h)
                    h_tbcs.append(h_tbc)
                logits, _ = engine.rnn_forward([m.rnn for m in group], None, seq_dev, seq_host, False, m0.conv_precision,
                                               h_tbc=h_tbcs)
                ys += [ops.squash_fwd(lg, 0.) for lg in logits]
        return ys, seq_host

    @classmethod
    def sound_event_detection_jointly(cls, models, inputs):
        ys, seq_len_y = cls.forward_jointly(models, inputs)
        t = ys[0].shape[-1]
        m = (torch.arange(t, device=ys[0].device)[None] < engine.seq_to_device(seq_len_y, ys[0].device)[:, None])[:, None, :]
        return [(y * m, seq_len_y) for y in ys]

    boundaries_detection_jointly = sound_event_detection_jointly

    def modify_summary(self, summary):
        """Called by the trainer before dumping a summary (reference models/strong_label/crnn.py modify_summary): metrics
        from the validation buffers, then the base class' scalar means / image grid."""
        if 'targets_strong' in summary['buffers']:
            self.add_metrics_to_summary(summary, 'strong')
        return super().modify_summary(summary)

    def review(self, inputs, outputs, defer_summary=False):
        """Loss + validation buffers (pb_sed/models/strong_label/crnn.py:95-138).  The host side of the summary
        (strong-label rate, strongly labelled clips, segment-wise maxima of scores and targets over
        ``eval_segment_length`` frames) is produced by one launch and fetched with ONE pinned device->host copy;
        ``defer_summary``: Trainer.step waits for it only after backward + Adam are enqueued."""
        y, seq_len_y, x, _, targets = outputs
        assert targets is not None
        strong_targets = targets[1].to(torch.float32)
        assert strong_targets.shape == y.shape, (strong_targets.shape, y.shape)
        seq_dev = engine.seq_to_device(seq_len_y, y.device)
        loss = _LossFunction.apply(y, strong_targets, seq_dev)
        b, k, t = y.shape
        seg = int(self.eval_segment_length)
        s = t // seg
        with torch.no_grad():
            packed_dev = torch.empty(2 * b * s * k + 2 * b * k, device=y.device, dtype=torch.float32)
            ops.bicrnn_review_summary(y.detach(), strong_targets, seq_dev, seg, packed_dev)
            host = torch.empty(packed_dev.shape, dtype=packed_dev.dtype, pin_memory=True)
            host.copy_(packed_dev, non_blocking=True)
            copied = torch.cuda.Event()
            copied.record()
        review = dict(loss=loss, scalars=dict(seq_len=np.mean(inputs['seq_len'])),
                      images=dict(features=x[:3], strong_targets=strong_targets[:3]), buffers={})
        seq = np.minimum(np.asarray(seq_len_y), t)

        def finalize():
            copied.synchronize()
            packed = host.numpy()
            n = b * s * k
            y_seg, t_seg = packed[:n].reshape(b, s, k), packed[n:2 * n].reshape(b, s, k)
            mask_mean, mask_cnt = packed[2 * n:2 * n + b * k].reshape(b, k), packed[2 * n + b * k:]
            labelled = np.flatnonzero((mask_mean > .999).all(-1))
            review['scalars']['strong_label_rate'] = float(mask_cnt.sum()) / (b * k * t)
            pick = lambda a: (np.concatenate([a[i, :seq[i] // seg] for i in labelled]) if len(labelled)
                              else np.zeros((0, k), np.float32))
            review['buffers'].update(y_strong=pick(y_seg), targets_strong=pick(t_seg))
            return review

        if defer_summary:
            review['_finalize'] = finalize
            return review
        return finalize()

    def tagging(self, inputs):
        y, seq_len_y, *_ = self.forward(inputs)
        return y.max(-1, keepdim=True)[0], np.ones_like(seq_len_y)

    def boundaries_detection(self, inputs):
        return self.sound_event_detection(inputs)

    def sound_event_detection(self, inputs):
        y, seq_len_y, *_ = self.forward(inputs)
        t = y.shape[-1]
        m = (torch.arange(t, device=y.device)[None] <
             engine.seq_to_device(seq_len_y, y.device)[:, None])[:, None, :]
        return y * m, seq_len_y


# ---- tuning wrappers (reference pb_sed/models/strong_label/crnn.py:213-262)
def tune_tagging(crnns, dataset, device, timestamps, event_classes, metrics, minimize=False, storage_dir=None):
    from .. import inference, tuning
    print('\nTagging Tuning')
    tagging_scores = inference.tagging(crnns, dataset, device, timestamps=timestamps, event_classes=event_classes)
    return tuning.tune_tagging(tagging_scores, medfilt_length_candidates=[1], metrics=metrics, minimize=minimize,
                               storage_dir=storage_dir, device=device)


def tune_boundary_detection(crnns, dataset, device, timestamps, event_classes, tags, metrics, stepfilt_lengths, minimize=False,
                            tag_masking=True, storage_dir=None):
    from .. import inference, tuning
    print('\nBoundaries Detection Tuning')
    boundaries_scores = inference.boundaries_detection(crnns, dataset, device, stepfilt_length=None, apply_mask=False, masks=tags,
                                                       timestamps=timestamps, event_classes=event_classes)
    return tuning.tune_boundaries_detection(boundaries_scores, medfilt_length_candidates=[1], stepfilt_length_candidates=stepfilt_lengths,
                                            tags=tags, metrics=metrics, minimize=minimize, tag_masking=tag_masking,
                                            storage_dir=storage_dir, device=device)


def tune_sound_event_detection(crnns, dataset, device, timestamps, event_classes, tags, metrics, medfilt_lengths, minimize=False,
                               tag_masking='?', storage_dir=None):
    from .. import inference, tuning
    print('\nSound Event Detection Tuning')
    detection_scores = inference.sound_event_detection(crnns, dataset, device, timestamps=timestamps, event_classes=event_classes)
    return tuning.tune_sound_event_detection(detection_scores, medfilt_lengths, tags, metrics=metrics, minimize=minimize,
                                             tag_masking=tag_masking, storage_dir=storage_dir, device=device)
