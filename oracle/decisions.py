"""Oracle: the DISCRETE decisions of a training pass, recorded from one run and imposed on another (tests only).

TEST INFRASTRUCTURE ONLY (see oracle/__init__.py).

The FBCRNN / BiCRNN training graph is piecewise smooth: every ReLU behind a norm (reference call sites
pb_sed/experiments/weak_label_crnn/training.py:218-242 -> padertorch CNN2d / CNN1d), every (2,1) max-pool
window (training.py:161-168) and the ``max(y_fwd, y_bwd)`` of the weak loss (pb_sed/models/weak_label/crnn.py:189) pick a
branch.  Two correct implementations whose pre-activations differ at rounding level pick different branches at a few of the
10^7..10^8 positions of a real-size step; each such position switches one contribution to a parameter gradient on or off,
which is NOT a rounding-level difference of the gradient (a per-channel beta gradient is a sum of a few thousand terms).
A gradient comparison that is meant to detect a 1e-3 defect therefore has to differentiate the SAME branch: the float64
oracle is run with the decisions of the implementation under test imposed (ReLU -> multiplication by that run's mask,
max-pool -> that run's selected row, max -> that run's selector).  What is left is rounding.

``collect`` / ``impose`` use module names (``cnn.cnn_2d.convs.3`` ...), which the oracle and the HIP build share.
"""
import torch

from .nn import _CNN, _ConvLayer


def record(model, on=True):
    """Make every conv layer / stack of ``model`` keep the decisions of its next forward pass."""
    for m in model.modules():
        if isinstance(m, (_ConvLayer, _CNN)):
            m.record = on
            for a in ('recorded', 'recorded_out_relu', 'recorded_skip_pool'):
                if hasattr(m, a):
                    delattr(m, a)


def collect(model, outputs=None):
    """Decisions of the last recorded forward: {'<layer name>': {'relu': bool, 'pool': bool}, '<stack name>':
    {'out_relu': bool, 'skip_pool': {(src, dst, k): bool}}, 'max_sel': bool [B,K,T] (FBCRNN outputs given)}."""
    dec = {}
    for name, m in model.named_modules():
        if isinstance(m, _ConvLayer) and getattr(m, 'recorded', None):
            dec[name] = dict(m.recorded)
        elif isinstance(m, _CNN):
            d = {}
            if hasattr(m, 'recorded_out_relu'):
                d['out_relu'] = m.recorded_out_relu
            if getattr(m, 'recorded_skip_pool', None):
                d['skip_pool'] = dict(m.recorded_skip_pool)
            if d:
                dec[name] = d
    if outputs is not None and len(outputs) == 6 and outputs[1] is not None:
        dec['max_sel'] = (outputs[0] >= outputs[1]).detach()       # ties: see impose()
    return dec


def impose(model, dec):
    """Impose ``dec`` (see collect) on ``model``; ``impose(model, None)`` removes every imposed decision.
    Returns the number of decision tensors put in place.  (A tie y_fwd == y_bwd splits the gradient in the reference's
    torch.maximum; the selector sends it to y_fwd.  Scores are 1e-5 + (1 - 2e-5) sigmoid(logit): ties need saturation.)"""
    n = 0
    mods = dict(model.named_modules())
    for m in mods.values():
        if isinstance(m, _ConvLayer):
            m.imposed = None
        elif isinstance(m, _CNN):
            m.imposed_out_relu, m.imposed_skip_pool = None, None
    if hasattr(model, 'imposed_sel'):
        model.imposed_sel = None
    if dec is None:
        return 0
    for name, d in dec.items():
        if name == 'max_sel':
            model.imposed_sel = d.cpu().bool()
            n += 1
            continue
        m = mods[name]
        if isinstance(m, _ConvLayer):
            m.imposed = {k: v.cpu() for k, v in d.items()}
            n += len(d)
        else:
            if 'out_relu' in d:
                m.imposed_out_relu = d['out_relu'].cpu()
                n += 1
            if 'skip_pool' in d:
                m.imposed_skip_pool = {k: v.cpu() for k, v in d['skip_pool'].items()}
                n += len(d['skip_pool'])
    return n


def disagreements(dec_a, dec_b, seq_len=None):
    """How many positions two decision sets differ at, per entry (diagnostics for the parity report).  ``seq_len``: count
    only positions t < seq_len[b] of the last axis (past the sequence the masked activations are 0 and a ReLU "decision"
    there is whatever sign the norm's shift has - without influence on anything)."""
    import numpy as np

    def count(v, w):
        d = v.cpu().bool() != w.cpu().bool().reshape(v.shape)
        if seq_len is not None and d.dim() >= 2 and d.shape[0] == len(seq_len):
            m = torch.arange(d.shape[-1])[None] < torch.as_tensor(np.asarray(seq_len))[:, None]
            d = d & m.reshape([d.shape[0]] + [1] * (d.dim() - 2) + [d.shape[-1]])
        return int(d.sum())

    out = {}
    for name, d in dec_a.items():
        if name not in dec_b:
            continue
        if torch.is_tensor(d):
            out[name] = count(d, dec_b[name])
            continue
        for k, v in d.items():
            w = dec_b[name].get(k)
            if w is None:
                continue
            if isinstance(v, dict):
                out[f'{name}.{k}'] = sum(count(v[i], w[i]) for i in v if i in w)
            else:
                out[f'{name}.{k}'] = count(v, w)
    return out


def positions(dec, seq_len=None):
    """Number of decided positions per entry of ``dec`` (same keys as ``disagreements``; inside the sequences when ``seq_len``
    is given) - the denominator of a bound on how many positions two runs may decide differently."""
    import numpy as np

    def count(v):
        if seq_len is not None and v.dim() >= 2 and v.shape[0] == len(seq_len):
            per_clip = v[0].numel() // v.shape[-1]
            return int(sum(min(int(n), v.shape[-1]) for n in np.asarray(seq_len)) * per_clip)
        return int(v.numel())

    out = {}
    for name, d in dec.items():
        if torch.is_tensor(d):
            out[name] = count(d)
            continue
        for k, v in d.items():
            out[f'{name}.{k}'] = sum(count(x) for x in v.values()) if isinstance(v, dict) else count(v)
    return out
