"""Tiny deterministic stand-ins for feature extractor / CNN / RNN used to pin the reference's
inference heads (tests/golden/gen_golden.py) and to replay them against the oracle / product classes."""
import torch


class StubFeatures(torch.nn.Module):
    def forward(self, x, seq_len=None, targets=None):
        return (x, seq_len) if targets is None else (x, seq_len, targets)


class StubCNN(torch.nn.Module):
    conditional_dims = 0

    def forward(self, x, seq_len=None, condition=None):
        return x, seq_len


class StubRNN(torch.nn.Module):
    """y[b,k,t] = sum_c A[k,c] * running-mean of h over the causal (or anti-causal) context."""

    def __init__(self, a, reverse):
        super().__init__()
        self.a = torch.as_tensor(a)
        self.reverse = reverse

    def forward(self, h, seq_len=None):
        t = h.shape[-1]
        n = torch.arange(1, t + 1, dtype=h.dtype)
        if self.reverse:
            ctx = h.flip(-1).cumsum(-1).div(n).flip(-1)
        else:
            ctx = h.cumsum(-1).div(n)
        return torch.einsum('kc,bct->bkt', self.a, ctx), seq_len


def make_tuning_metrics(targets, event_classes):
    """Two deterministic metric functions ``fn(scores) -> (metric_values, other_values)`` over a dict of score DataFrames, for pinning
    the tuning drivers (tests/golden/gen_golden.py::gen_tuning runs the REFERENCE's drivers with them, the tests run the build's):
    'hit_rate' (to maximise; per class the best of three thresholds on the clip's peak score against ``targets[audio_id][k]``, the
    threshold reported as another value) and 'leak' (to minimise; mean score of the negative clips minus that of the positive ones).
    Both depend on every filtered value, so a wrong filter length, edge or mask shows."""
    import numpy as np

    def hit_rate(scores):
        values, other = {}, {}
        ids = sorted(scores)
        for k, c in enumerate(event_classes):
            peak = np.array([scores[a][c].to_numpy().max() for a in ids])
            tgt = np.array([targets[a][k] for a in ids]) > .5
            accs = [float(np.mean((peak > thr) == tgt)) for thr in (.3, .5, .7)]
            best = int(np.argmax(accs))
            values[c], other[c] = accs[best], {'threshold': (.3, .5, .7)[best]}
        values['macro_average'] = float(np.mean([values[c] for c in event_classes]))
        return values, other

    def leak(scores):
        values = {}
        ids = sorted(scores)
        for k, c in enumerate(event_classes):
            m = np.array([scores[a][c].to_numpy().mean() for a in ids])
            tgt = np.array([targets[a][k] for a in ids]) > .5
            values[c] = float(m[~tgt].mean() - m[tgt].mean())
        values['macro_average'] = float(np.mean([values[c] for c in event_classes]))
        return values, {}

    return {'hit_rate': hit_rate, 'leak': leak}
