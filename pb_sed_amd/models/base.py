"""Mirror of pb_sed.models.base.model.SoundEventModel (reference pb_sed/models/base/model.py:9-88):
abstract inference API, ``example_to_device`` (called at pb_sed/models/base/inference.py:130) and the validation
summary: ``modify_summary`` averages the scalars over batches, ``add_metrics_to_summary`` turns the concatenated score /
target buffers of a validation run into macro F-score, error rate, lwlrap, mAP and mAUC (host side, numpy)."""
import abc

import numpy as np
import torch
from torch import nn

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


class SoundEventModel(nn.Module, abc.ABC):
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
            if isinstance(v, torch.Tensor):
                return v.to(device)
            if isinstance(v, np.ndarray) and v.dtype.kind == 'f':
                return torch.from_numpy(np.ascontiguousarray(v)).to(device)
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

    def add_metrics_to_summary(self, summary, suffix):
        """Validation metrics from the ``y_<suffix>`` / ``targets_<suffix>`` buffers (reference model.py:44-88)."""
        from sklearn import metrics
        from ..evaluation import instance_based
        y = np.concatenate(summary['buffers'].pop(f'y_{suffix}'))
        summary['scalars'][f'num_examples_{suffix}'] = len(y)
        targets = np.concatenate(summary['buffers'].pop(f'targets_{suffix}'))
        test_labels = self.test_labels
        if test_labels is not None:
            if isinstance(test_labels[0], str):
                assert self.label_mapping is not None
                test_labels = [self.label_mapping.index(label) for label in test_labels]
            y, targets = y[..., test_labels], targets[..., test_labels]

        def label_wise(key, values):
            if key not in self.labelwise_metrics:
                return
            for event_class, value in enumerate(values):
                if test_labels is not None:
                    event_class = test_labels[event_class]
                if self.label_mapping is not None:
                    event_class = self.label_mapping[event_class]
                summary['scalars'][f'z/{key}/{event_class}'] = value

        _, f, _, _ = instance_based.get_best_fscore_thresholds(targets, y)
        summary['scalars'][f'macro_fscore_{suffix}'] = f.mean()
        label_wise(f'fscore_{suffix}', f)
        _, er, _, _ = instance_based.get_best_er_thresholds(targets, y)
        summary['scalars'][f'macro_error_rate_{suffix}'] = er.mean()
        label_wise(f'error_rate_{suffix}', er)
        lwlrap, per_class_lwlrap, _ = instance_based.lwlrap(targets, y)
        summary['scalars'][f'lwlrap_{suffix}'] = lwlrap
        label_wise(f'lwlrap_{suffix}', per_class_lwlrap)
        if (targets.sum(0) > 1).all():
            ap = metrics.average_precision_score(targets, y, average=None)
            summary['scalars'][f'map_{suffix}'] = np.mean(ap)
            label_wise(f'ap_{suffix}', ap)
            auc = metrics.roc_auc_score(targets, y, average=None)
            summary['scalars'][f'mauc_{suffix}'] = np.mean(auc)
            label_wise(f'auc_{suffix}', auc)

    # ---- helpers shared by both CRNNs
    def _seq(self, inputs, device):
        seq_host = np.array(inputs['seq_len'])
        seq_dev = seq_to_device(seq_host, device)
        return seq_host, seq_dev
