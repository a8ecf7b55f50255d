"""Instance-based (clip-level) validation metrics: best-threshold F-score / error rate per class and lwlrap.

Host-side mirror of the reference's pb_sed/evaluation/instance_based.py (same function names, argument meaning and
return tuples: get_best_fscore_thresholds :282-309, get_best_er_thresholds :337-358, fscore_curve :246-279,
er_curve :312-334, lwlrap :186-229, fscore :33-57, error_rate :105-126), used by SoundEventModel.add_metrics_to_summary
(pb_sed/models/base/model.py:44-88) on the validation buffers.  These are [examples, classes] matrices of a few
thousand rows, evaluated once per validation run: plain numpy, no device code.

Threshold curves are computed for all classes at once from one sort per column.  For a column with the distinct
scores u_0 < ... < u_{m-1} the candidate thresholds are -inf, the midpoints (u_i + u_{i+1})/2 and +inf (a decision is
`score > threshold`); the reference evaluates the same candidates once per *instance* (ties repeat a candidate), so the
optimum and its "last best candidate" tie rule are identical.
"""
import numpy as np


def _as_columns(targets, scores):
    targets, scores = np.asarray(targets), np.asarray(scores)
    assert 0 < scores.ndim <= 2, scores.shape
    assert scores.shape == targets.shape, (scores.shape, targets.shape)
    flat = scores.ndim == 1
    if flat:
        targets, scores = targets[:, None], scores[:, None]
    return targets.astype(np.float64), scores, flat


def _candidate_counts(targets, scores):
    """Per class column: candidate thresholds (n+1, ascending, one per sorted instance + the final +inf; tied scores
    share their candidate), the number of detections above each candidate and the true positives among them."""
    n, k = scores.shape
    order = np.argsort(scores, axis=0, kind='stable')
    s = np.take_along_axis(scores, order, axis=0).astype(np.float64)
    t = np.take_along_axis(targets, order, axis=0)
    s_ext = np.concatenate([s, np.full((1, k), np.inf)], axis=0)                  # candidate i sits below instance i
    above = (n - np.arange(n + 1))[:, None] * np.ones((1, k))                      # detections if every s_j, j >= i, fires
    tp_above = np.concatenate([np.cumsum(t[::-1], axis=0)[::-1], np.zeros((1, k))], axis=0)
    # candidates of tied instances collapse onto the first of the run
    first = np.ones((n + 1, k), dtype=bool)
    first[1:] = s_ext[1:] != s_ext[:-1]
    run_start = np.maximum.accumulate(np.where(first, np.arange(n + 1)[:, None], 0), axis=0)
    above = np.take_along_axis(above, run_start, axis=0)
    tp_above = np.take_along_axis(tp_above, run_start, axis=0)
    s_run = np.take_along_axis(s_ext, run_start, axis=0)
    # threshold of a run = midpoint to the previous distinct score (-inf for the lowest run)
    prev_idx = np.maximum(run_start - 1, 0)
    s_prev = np.take_along_axis(s_ext, prev_idx, axis=0)
    with np.errstate(invalid='ignore'):
        thr = np.where(run_start == 0, -np.inf, (s_run + s_prev) / 2)
    return thr, above, tp_above


def _last_best(metric, maximise):
    """Row index of the best candidate per column; among equal optima the last (highest threshold) one."""
    m = metric[::-1]
    idx = np.argmax(m, axis=0) if maximise else np.argmin(m, axis=0)
    return metric.shape[0] - 1 - idx


def fscore_curve(targets, scores, beta=1., tp_bias=0, n_ref_bias=0, n_pos_bias=0):
    """(thresholds, f, precision, recall), one row per instance + 1, per class column (1-D inputs give 1-D outputs)."""
    targets, scores, flat = _as_columns(targets, scores)
    if not flat:
        # the reference evaluates a matrix column by column WITHOUT forwarding beta or the biases
        # (instance_based.py:268-270): matrices always get the plain F1 curve.  Kept, results must be identical.
        beta, tp_bias, n_ref_bias, n_pos_bias = 1., 0, 0, 0
    thr, n_pos, tps = _candidate_counts(targets, scores)
    n_ref = tps[0]
    p = (tps + tp_bias) / np.maximum(n_pos + n_pos_bias, 1)
    r = (tps + tp_bias) / np.maximum(n_ref + n_ref_bias, 1)
    f = (1 + beta ** 2) * p * r / (beta ** 2 * p + r + 1e-18)
    out = (thr, f, p, r)
    return tuple(o[:, 0] for o in out) if flat else out


def get_best_fscore_thresholds(targets, scores, beta=1., min_precision=0., min_recall=0., tp_bias=0, n_ref_bias=0,
                               n_pos_bias=0):
    """Per class: (threshold, f, precision, recall) of the candidate with the highest F-score."""
    assert min_precision == 0. or min_recall == 0.
    flat = np.asarray(scores).ndim == 1
    thr, f, p, r = fscore_curve(targets, scores, beta, tp_bias=tp_bias, n_ref_bias=n_ref_bias, n_pos_bias=n_pos_bias)
    if flat:
        thr, f, p, r = thr[:, None], f[:, None], p[:, None], r[:, None]
    f = np.where((p < min_precision) | (r < min_recall), 0., f)
    best = _last_best(f, True)
    cols = np.arange(f.shape[1])
    out = (thr[best, cols], f[best, cols], p[best, cols], r[best, cols])
    return tuple(o[0] for o in out) if flat else out


def er_curve(targets, scores):
    """(thresholds, error rate, insertion rate, deletion rate) per candidate and class column."""
    targets, scores, flat = _as_columns(targets, scores)
    thr, n_pos, tps = _candidate_counts(targets, scores)
    n_ref = np.maximum(tps[0], 1)
    ins, dele = n_pos - tps, tps[0] - tps
    out = (thr, (ins + dele) / n_ref, ins / n_ref, dele / n_ref)
    return tuple(o[:, 0] for o in out) if flat else out


def get_best_er_thresholds(targets, scores, max_insertion_rate=None, max_deletion_rate=None):
    """Per class: (threshold, error rate, insertion rate, deletion rate) of the candidate with the lowest error rate."""
    flat = np.asarray(scores).ndim == 1
    thr, er, ir, dr = er_curve(targets, scores)
    if flat:
        thr, er, ir, dr = thr[:, None], er[:, None], ir[:, None], dr[:, None]
    if max_insertion_rate is not None:
        er = np.where(ir > max_insertion_rate, np.inf, er)
    if max_deletion_rate is not None:
        er = np.where(dr > max_deletion_rate, np.inf, er)
    best = _last_best(er, False)
    cols = np.arange(er.shape[1])
    out = (thr[best, cols], er[best, cols], ir[best, cols], dr[best, cols])
    return tuple(o[0] for o in out) if flat else out


def _confusion(target_mat, decision_mat):
    t, d = np.asarray(target_mat, dtype=np.float64), np.asarray(decision_mat, dtype=np.float64)
    return t * d, (1. - t) * d, t * (1. - d)          # tp, fp (insertions), fn (deletions)


def fscore(target_mat, decision_mat, beta=1., event_wise=False):
    """F-beta, precision, recall of binary decisions [..., instances, classes] (event_wise: per class)."""
    axes = -2 if event_wise else (-2, -1)
    tp, fp, fn = (x.sum(axis=axes) for x in _confusion(target_mat, decision_mat))
    p, r = tp / np.maximum(tp + fp, 1), tp / np.maximum(tp + fn, 1)
    return (1 + beta ** 2) * p * r / np.maximum(beta ** 2 * p + r, 1e-15), p, r


def error_rate(target_mat, decision_mat, event_wise=False):
    """(error rate, substitution, insertion, deletion rates); substitutions pair an insertion with a deletion of the
    same instance when classes are pooled."""
    _, ins, dele = _confusion(target_mat, decision_mat)
    if event_wise:
        sub = np.zeros_like(ins)
        axes = -2
    else:
        ins, dele = ins.sum(-1, keepdims=True), dele.sum(-1, keepdims=True)
        sub = np.minimum(ins, dele)
        ins, dele = ins - sub, dele - sub
        axes = (-2, -1)
    n_ref = np.maximum(np.asarray(target_mat, dtype=np.float64).sum(axis=axes), 1)
    sub, ins, dele = sub.sum(axis=axes), ins.sum(axis=axes), dele.sum(axis=axes)
    return (sub + ins + dele) / n_ref, sub / n_ref, ins / n_ref, dele / n_ref


def lwlrap(target_mat, score_mat):
    """Label-weighted label-ranking average precision: (lwlrap, per-class lwlrap, class weights)."""
    target_mat, score_mat = np.asarray(target_mat) > 0, np.asarray(score_mat)
    if not target_mat.any():
        return 0.0, np.zeros(target_mat.shape[-1])
    assert score_mat.ndim == 2 and target_mat.shape == score_mat.shape, (target_mat.shape, score_mat.shape)
    n, k = score_mat.shape
    ranked = np.argsort(score_mat, axis=-1)[:, ::-1]                      # classes by descending score
    hit = np.take_along_axis(target_mat, ranked, axis=-1)
    prec = np.cumsum(hit, axis=-1) / np.arange(1, k + 1)                  # precision of the list cut at each rank
    per_class, count = np.zeros(k), np.zeros(k)
    np.add.at(per_class, ranked[hit], prec[hit])
    np.add.at(count, ranked[hit], 1)
    per_class /= np.maximum(count, 1)
    weight = count / count.sum()
    return float((per_class * weight).sum()), per_class, weight
