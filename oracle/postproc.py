"""Oracle: ensemble post-processing + event extraction restated (CPU, numpy/scipy).

TEST INFRASTRUCTURE ONLY (see oracle/__init__.py).
* ``medfilt`` / ``stepfilt``: pb_sed/filters.py:56-83, :112-135 (reference-owned, golden-pinned)
* ``boundariesfilt`` / ``filtering`` / ``postprocess``: pb_sed/models/base/inference.py:266-289,
  :225-263, :142-184 (reference-owned, golden-pinned)
* ``scores_to_event_list``: sed_scores_eval@a922e0a (absent) restated from SURVEY A.7 - parity
  unpinned; call site pb_sed/experiments/strong_label_crnn/inference.py:147-150.
"""
import numpy as np
from scipy import signal


def medfilt(x, n, axis=-1):
    """scipy.signal.medfilt along ``axis`` (zero-padded edges); identity for n == 1."""
    n = int(n)
    if n == 1:
        return x
    x = np.moveaxis(x, axis, -1)
    shape = x.shape
    y = np.stack([signal.medfilt(r, n) for r in x.reshape(-1, shape[-1])]).reshape(shape)
    return np.moveaxis(y, -1, axis)


def stepfilt(x, n, axis=-1):
    """valid correlation of pad(x, (n/2, n/2-1)) with [-1]*(n/2)+[+1]*(n/2) scaled by 1/(n/2); f64 out."""
    n = int(n)
    assert n % 2 == 0
    filt = np.concatenate((-np.ones(n // 2), np.ones(n // 2))) / (n // 2)
    x = np.moveaxis(x, axis, -1)
    shape = x.shape
    xp = np.pad(x.reshape(-1, shape[-1]), [(0, 0), (n // 2, n // 2 - 1)])
    y = np.stack([np.correlate(r, filt, mode='valid') for r in xp]).reshape(shape)
    return np.moveaxis(y, -1, axis)


def boundariesfilt(x, n, axis=-1):
    n = int(n)
    if n > 0:
        fwd, bwd = stepfilt(x, n, axis), stepfilt(np.flip(x, axis), n, axis)
    else:
        fwd, bwd = x, np.flip(x, axis)
    return np.minimum(np.maximum.accumulate(fwd, axis=axis),
                      np.flip(np.maximum.accumulate(bwd, axis=axis), axis))


def filtering(score_arr, filter_fn, filter_length):
    """Per-class (1-D lengths) / per-variant (2-D lengths -> [B,n,K,T]) filtering, results cast
    back into the score array's dtype on assignment (inference.py:225-263)."""
    filter_length = np.asarray(filter_length)
    b, *_, k, t = score_arr.shape
    if filter_length.ndim == 0:
        return filter_fn(score_arr, filter_length, axis=-1)
    if filter_length.ndim == 1:
        assert filter_length.shape[0] == k
        for c, n in enumerate(filter_length):
            score_arr[..., c, :] = filter_fn(score_arr[..., c, :], n, axis=-1)
        return score_arr
    assert filter_length.ndim == 2 and filter_length.shape[1] in (1, k)
    nv = filter_length.shape[0]
    if score_arr.ndim == 3:
        score_arr = np.broadcast_to(score_arr[:, None], (b, nv, k, t)).copy()
    for j in range(nv):
        if filter_length.shape[1] == 1:
            score_arr[:, j] = filter_fn(score_arr[:, j], filter_length[j, 0], axis=-1)
        else:
            for c in range(k):
                score_arr[:, j, c] = filter_fn(score_arr[:, j, c], filter_length[j, c], axis=-1)
    return score_arr


def postprocess(model_scores, seq_len, example_ids, medfilt_length=1, stepfilt_length=None,
                apply_mask=False, masks=None, tagging=False):
    """inference.py:142-184 for one batch: list of per-model [B,(n,)K,T] -> {id: [(n,)T,K]}."""
    s = np.mean(model_scores, axis=0)
    t = s.shape[-1]
    m = (np.arange(t)[None] < np.asarray(seq_len)[:, None]).astype(s.dtype)
    s = s * m.reshape((len(seq_len),) + (1,) * (s.ndim - 2) + (t,))
    s = filtering(s, medfilt, np.array(medfilt_length, dtype=int))
    if stepfilt_length is not None:
        s = filtering(s, boundariesfilt, np.array(stepfilt_length, dtype=int))
    out = {}
    for i, (aid, sl) in enumerate(zip(example_ids, seq_len)):
        x = s[i, ..., :sl].swapaxes(-2, -1)
        out[aid] = x.max(-2, keepdims=True) if tagging else x
    apply_mask = np.array(apply_mask, dtype=bool)
    if apply_mask.any():
        if apply_mask.ndim == 2:
            apply_mask = apply_mask[..., None, :]
        for aid in out:
            out[aid] *= np.maximum(masks[aid], 1 - apply_mask)   # in place: keeps float32
    return out


def scores_to_event_list(scores, timestamps, thresholds, event_classes):
    """scores [T,K], timestamps [T+1] -> sorted [(onset, offset, label)]; det = score > thr (strict)."""
    scores = np.asarray(scores)
    thresholds = np.broadcast_to(np.asarray(thresholds, dtype=np.float64), (scores.shape[1],))
    events = []
    for k, label in enumerate(event_classes):
        det = scores[:, k] > thresholds[k]
        z = np.concatenate(([False], det, [False])).astype(np.int8)
        d = np.diff(z)
        on, off = np.flatnonzero(d == 1), np.flatnonzero(d == -1)
        events += [(float(timestamps[a]), float(timestamps[b]), label) for a, b in zip(on, off)]
    return sorted(events)


def event_frames(scores, thresholds):
    """Index-level form: list over classes of int64 [n_events, 2] (onset_frame, offset_frame)."""
    scores = np.asarray(scores)
    thresholds = np.broadcast_to(np.asarray(thresholds, dtype=np.float64), (scores.shape[1],))
    out = []
    for k in range(scores.shape[1]):
        z = np.concatenate(([False], scores[:, k] > thresholds[k], [False])).astype(np.int8)
        d = np.diff(z)
        out.append(np.stack([np.flatnonzero(d == 1), np.flatnonzero(d == -1)], -1).astype(np.int64))
    return out
