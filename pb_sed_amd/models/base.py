"""Mirror of pb_sed.models.base.model.SoundEventModel (reference pb_sed/models/base/model.py:9-88):
abstract inference API, ``example_to_device`` (called at pb_sed/models/base/inference.py:130) and the validation
summary: ``modify_summary`` averages the scalars over batches, ``add_metrics_to_summary`` turns the concatenated score /
target buffers of a validation run into macro F-score, error rate, lwlrap, mAP and mAUC (host side, numpy)."""
import abc

import numpy as np
import torch
from torch import nn

from .. import ops
from ..configurable import Configurable
from ..engine import flatten_parameters, seq_to_device


def _image_column(images, padding=2):
    """[N,1,H,W] -> [3, N*(H+pad)+pad, W+2*pad]: min-max normalised over all images, stacked in one column."""
    images = images.detach().float()
    lo, hi = images.min(), images.max()
    images = ((images - lo) / (hi - lo).clamp_min(1e-5)).clamp(0, 1)
    n, c, h, w = images.shape
    if n == 1:                                        # make_grid returns a single image unpadded
        return images[0].expand(3, h, w).clone()
    grid = images.new_zeros((3, n * (h + padding) + padding, w + 2 * padding))
    for i in range(n):
        grid[:, padding + i * (h + padding): padding + i * (h + padding) + h, padding: padding + w] = images[i]
    return grid


class SoundEventModel(nn.Module, Configurable, abc.ABC):
    def __init__(self, *, labelwise_metrics=(), label_mapping=None, test_labels=None):
        super().__init__()
        self.labelwise_metrics = labelwise_metrics
        self.label_mapping = label_mapping
        self.test_labels = test_labels
        # operand format of the conv / projection MFMAs: 'f32' (default), 'bf16' (BASELINE config 3) or
        # 'bf16x3' (exact 3-way bf16 split, fp32-class accuracy); weight gradients, BN, GRU scans stay fp32
        self.conv_precision = 'f32'
        self.keep_logits = False      # True: forward stores the pre-sigmoid head outputs in ``last_logits``
        self.last_logits = None

    @abc.abstractmethod
    def tagging(self, inputs, **params):
        pass

    @abc.abstractmethod
    def boundaries_detection(self, inputs, **params):
        pass

    @abc.abstractmethod
    def sound_event_detection(self, inputs, **params):
        pass

    def example_to_device(self, example, device=None):
        """Move every tensor / float ndarray of the (nested) example to ``device``."""
        def mv(v):
            # a copy out of pageable host memory stalls the host until the stream has drained (ops.host_to_device): pinned
            # tensors go asynchronously, small pageable ones through a pinned staging buffer, big pageable ones as they are
            if isinstance(v, np.ndarray) and v.dtype.kind == 'f':
                v = torch.from_numpy(np.ascontiguousarray(v))
            if isinstance(v, torch.Tensor):
                if v.is_cuda or torch.device(device).type != 'cuda':
                    return v.to(device)
                if v.is_pinned():
                    return v.to(device, non_blocking=True)
                return ops.host_to_device(v, device) if v.numel() * v.element_size() <= (1 << 20) else v.to(device)
            if isinstance(v, dict):
                return {k: mv(x) for k, x in v.items()}
            return v
        return {k: mv(v) for k, v in example.items()}

    def flat_parameters(self):
        """(flat_param, flat_grad): the buffers every parameter / gradient aliases."""
        return flatten_parameters(self)

    def modify_summary(self, summary):
        """Called by the trainer before a summary is dumped (reference model.py:28-42): scalars collected per batch
        become their mean; image tensors become a one-column grid normalised over the whole grid (what
        torchvision.utils.make_grid(image.flip(2), normalize=True, scale_each=False, nrow=1) returns for 3-channel
        output, with its 2-pixel padding)."""
        for key, scalar in summary.get('scalars', {}).items():
            summary['scalars'][key] = np.mean(scalar)
        for key, image in summary.get('images', {}).items():
            if image.dim() == 4 and image.shape[1] > 1:
                image = image[:, 0]
            if image.dim() == 3:
                image = image.unsqueeze(1)
            summary['images'][key] = _image_column(image.flip(2))
        return summary

    def _class_names(self):
        """(column selection or None, display name per selected column) from ``test_labels`` / ``label_mapping``."""
        cols = self.test_labels
        if cols is not None and isinstance(cols[0], str):
            assert self.label_mapping is not None
            cols = [self.label_mapping.index(name) for name in cols]
        return cols, (lambda j: (self.label_mapping[c] if self.label_mapping is not None else c)
                      if (c := (j if cols is None else cols[j])) is not None else j)

    def add_metrics_to_summary(self, summary, suffix):
        """Validation metrics from the ``y_<suffix>`` / ``targets_<suffix>`` buffers (contract: the scalar keys of the
        reference's pb_sed/models/base/model.py:44-88).  Expressed as a table: (aggregate key, label-wise key, per-class
        values, aggregate) - the ranking metrics are only defined when every class has at least two positives."""
        from ..evaluation import instance_based
        scalars, buffers = summary['scalars'], summary['buffers']
        scores = np.concatenate(buffers.pop(f'y_{suffix}'))
        targets = np.concatenate(buffers.pop(f'targets_{suffix}'))
        scalars[f'num_examples_{suffix}'] = len(scores)
        cols, name_of = self._class_names()
        if cols is not None:
            scores, targets = scores[..., cols], targets[..., cols]
        overall, per_class_lw, _ = instance_based.lwlrap(targets, scores)
        rows = [
            (f'macro_fscore_{suffix}', f'fscore_{suffix}', instance_based.get_best_fscore_thresholds(targets, scores)[1], None),
            (f'macro_error_rate_{suffix}', f'error_rate_{suffix}', instance_based.get_best_er_thresholds(targets, scores)[1], None),
            (f'lwlrap_{suffix}', f'lwlrap_{suffix}', per_class_lw, overall),
        ]
        if (targets.sum(0) > 1).all():
            from sklearn import metrics
            rows += [(f'map_{suffix}', f'ap_{suffix}', metrics.average_precision_score(targets, scores, average=None), None),
                     (f'mauc_{suffix}', f'auc_{suffix}', metrics.roc_auc_score(targets, scores, average=None), None)]
        for aggregate_key, labelwise_key, values, aggregate in rows:
            scalars[aggregate_key] = np.mean(values) if aggregate is None else aggregate
            if labelwise_key in self.labelwise_metrics:
                scalars.update({f'z/{labelwise_key}/{name_of(j)}': v for j, v in enumerate(values)})

    # ---- helpers shared by both CRNNs
    def _seq(self, inputs, device):
        seq_host = np.array(inputs['seq_len'])
        seq_dev = seq_to_device(seq_host, device)
        return seq_host, seq_dev

    def input_key(self, inputs):
        return 'audio_data' if 'audio_data' in inputs else 'stft'

    def features(self, inputs, x_in, seq_host, seq_dev):
        """Normalised log-mel [B,1,F,T] of either input contract: the reference's ``'stft'`` [B,1,T,bins,2]
        (pb_sed/models/weak_label/crnn.py:79-90) or the waveform ``'audio_data'`` [B,N] (fused STFT).  Segments cut out of
        long clips (pb_sed_amd/utils/segment.py) carry ``'stft_pad_front'`` / ``'num_frames'``; ``'frame_pos'`` [B,T] selects
        time-warped framing of the waveform (pb_sed_amd/data.py::TimeWarp)."""
        from .. import engine
        from ..modules import num_frames
        fe = self.feature_extractor
        if 'audio_data' in inputs or x_in.dim() != 5:      # a waveform: [B,N], or [B,1,N] from data.collate; stft is [B,1,T,bins,2]
            audio = x_in.reshape(x_in.shape[0], -1).to(torch.float32)
            n_frames = int(inputs.get('num_frames', 0)) or num_frames(audio.shape[1])
            frame_pos = inputs.get('frame_pos')                  # time-warped framing drawn by data.TimeWarp
            if frame_pos is not None:
                frame_pos = (frame_pos.to(device=audio.device, dtype=torch.int32) if isinstance(frame_pos, torch.Tensor)
                             and frame_pos.is_cuda else ops.host_to_device(frame_pos, audio.device, torch.int32)).contiguous()
            return engine.features_from_audio(fe, audio, seq_dev, n_frames, seq_host,
                                              pad_front=int(inputs.get('stft_pad_front', 320)), frame_pos=frame_pos)
        return engine.features_from_stft(fe, x_in, seq_host, seq_dev)
