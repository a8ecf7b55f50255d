"""Mirror of pb_sed.models.base.model.SoundEventModel (reference pb_sed/models/base/model.py:9-26):
abstract inference API + ``example_to_device`` (called at pb_sed/models/base/inference.py:130)."""
import abc

import numpy as np
import torch
from torch import nn

from ..engine import flatten_parameters, seq_to_device


class SoundEventModel(nn.Module, abc.ABC):
    def __init__(self, *, labelwise_metrics=(), label_mapping=None, test_labels=None):
        super().__init__()
        self.labelwise_metrics = labelwise_metrics
        self.label_mapping = label_mapping
        self.test_labels = test_labels
        # operand format of the conv / projection MFMAs: 'f32' (default), 'bf16' (BASELINE config 3) or
        # 'bf16x3' (exact 3-way bf16 split, fp32-class accuracy); weight gradients, BN, GRU scans stay fp32
        self.conv_precision = 'f32'

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
        return summary

    # ---- helpers shared by both CRNNs
    def _seq(self, inputs, device):
        seq_host = np.array(inputs['seq_len'])
        seq_dev = seq_to_device(seq_host, device)
        return seq_host, seq_dev
