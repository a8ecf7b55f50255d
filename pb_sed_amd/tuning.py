"""Hyper-parameter tuning drivers: the leaderboard search over median-filter / step-filter lengths and tag masking that follows
every ensemble inference pass of the reference's tuning scripts.

Counterpart of ``pb_sed/models/base/tuning.py`` (reference: ``update_leaderboard`` :13-47, ``tune_tagging`` :50-100,
``boundaries_from_events`` :103-122, ``tune_boundaries_detection`` :125-207, ``tune_sound_event_detection`` :210-281; driven from
pb_sed/experiments/weak_label_crnn/tuning.py:55-65 and pb_sed/experiments/strong_label_crnn/tuning.py:64 with up to 5 window
lengths x 8 median-filter lengths x an ensemble).  Same signatures, same leaderboard
``{metric_name: (metric_values, hyper_params_and_other_values, scores)}``; what changes underneath: the reference filters every
clip's DataFrame on the host, one ``scipy.signal.medfilt`` per class column and candidate - here the clips of one length go to
the device as ONE ``[clips, classes, T]`` tensor per candidate and run through the bit-exact HIP filters of the inference path
(``pbsed_medfilt`` / ``pbsed_boundariesfilt``, csrc/postproc.hip - pinned against the reference's own filter vectors), so a
candidate costs two launches instead of clips x classes host calls.  The metric functions stay the caller's
(``metrics = {name: fn(scores) -> (values, other_values)}``): the reference's own ``f_tag`` / ``f_collar`` / ``psd_auc``
(:284-343) are thin wrappers over sed_scores_eval, which is not part of this build - pass them in where it is installed.

Pinned by ``tests/golden/ref_tuning.npz``: leaderboards the reference's own functions produced for the same inputs
(tests/golden/gen_golden.py), compared value for value, hyper-parameter for hyper-parameter, score array for score array.
"""
import copy
import json
import os

import numpy as np
import torch

from . import ops


def _columns(frame, event_classes=None):
    """(timestamps, event_classes) of a score DataFrame: 'onset', 'offset', then one column per class, frames back to back
    (what sed_scores_eval's ``validate_score_dataframe`` checks and returns)."""
    names = list(frame.columns)
    if names[:2] != ['onset', 'offset']:
        raise ValueError(f'score DataFrame: expected the columns onset, offset, <classes>; got {names[:4]} ...')
    if event_classes is not None and list(event_classes) != names[2:]:
        raise ValueError(f'score DataFrame: classes {names[2:]} differ from {list(event_classes)}')
    onset, offset = frame['onset'].to_numpy(), frame['offset'].to_numpy()
    if len(onset) > 1 and (offset[:-1] != onset[1:]).any():
        raise ValueError('score DataFrame: frames are not back to back')
    return np.concatenate((onset, offset[-1:])), names[2:]


def _on_device(scores, audio_ids, event_classes, device, fn):
    """``fn`` (a device filter over ``[clips, classes, T]``) applied to the class columns of every clip; clips of equal length
    share a launch (a zero-padded filter's edge belongs to the clip: rows are never padded to a common length).  Returns
    ``{audio_id: float64 [T, classes]}``."""
    by_len = {}
    for audio_id in audio_ids:
        by_len.setdefault(len(scores[audio_id]), []).append(audio_id)
    out = {}
    for t, ids in sorted(by_len.items()):
        host = np.stack([scores[a][event_classes].to_numpy().T for a in ids])          # [clips, classes, T] float64
        x32 = host.astype(np.float32)
        if not np.array_equal(x32.astype(np.float64), host):
            raise ValueError('tuning filters run in float32 on the device: the scores must be float32 values (as the inference '
                             'methods return them)')
        y = fn(ops.host_to_device(x32, device)).to('cpu', torch.float64).numpy()
        for i, a in enumerate(ids):
            out[a] = y[i].T
    return out


def _replaced(scores, audio_ids, event_classes, arrays):
    """deep copy of the score dict with the class columns of every clip replaced"""
    new = copy.deepcopy(scores)
    for a in audio_ids:
        new[a][event_classes] = arrays[a]
    return new


def _dump(obj, path):
    os.makedirs(os.path.dirname(str(path)) or '.', exist_ok=True)
    with open(path, 'w') as f:
        json.dump(obj, f, indent=2, sort_keys=True, default=lambda v: v.item() if hasattr(v, 'item') else str(v))


def _minimized(minimize, metric_name):
    if isinstance(minimize, dict):
        return bool(minimize[metric_name])
    if isinstance(minimize, (list, tuple)):
        return metric_name in minimize
    return bool(minimize)


def update_leaderboard(leaderboard, metric_name, metric_values, hyper_params_and_other_values, scores, minimize=False):
    """One candidate against the board.  Per class: if the candidate's metric value is at least as good as the board's
    (ties go to the LATER candidate), the class takes the candidate's value, hyper-parameters and - in every clip - its score
    column.  The first candidate of a metric founds the board with deep copies.  ``macro_average`` is recomputed over the
    candidate's classes."""
    classes = list(hyper_params_and_other_values)
    if metric_name not in leaderboard:
        leaderboard[metric_name] = ({c: metric_values[c] for c in classes}, copy.deepcopy(hyper_params_and_other_values),
                                    copy.deepcopy(scores))
    else:
        sign = -1. if _minimized(minimize, metric_name) else 1.
        best_values, best_params, best_scores = leaderboard[metric_name]
        for c in classes:
            if metric_values[c] * sign >= best_values[c] * sign:
                best_values[c] = metric_values[c]
                best_params[c].update(hyper_params_and_other_values[c])
                for audio_id in best_scores:
                    best_scores[audio_id][c] = scores[audio_id][c]
    leaderboard[metric_name][0]['macro_average'] = float(np.mean([leaderboard[metric_name][0][c] for c in classes]))
    return leaderboard


def _candidate(leaderboard, metric_name, metric_fn, scores, params, minimize, verbose, what):
    metric_values, other_values = metric_fn(scores)
    if verbose:
        print(f'\n{what}\n{metric_values}')
    entry = {c: {**params, **other_values.get(c, {})} for c in metric_values if not c.endswith('_average')}
    return update_leaderboard(leaderboard, metric_name, metric_values, entry, scores, minimize=minimize)


def _store(leaderboard, storage_dir, stem, always_annotate=False):
    """The tuned hyper-parameters per metric, each class entry carrying its metric value, as
    ``<storage_dir>/<stem>_hyper_params_<metric>.json``.  (The reference annotates the entries only when it writes them - except
    in tune_sound_event_detection, which always does: kept.)"""
    for metric_name, (values, params, _) in leaderboard.items():
        if storage_dir is not None or always_annotate:
            for c in params:
                params[c][metric_name] = values[c]
        if storage_dir is not None:
            _dump(params, os.path.join(str(storage_dir), f'{stem}_hyper_params_{metric_name}.json'))


def _tag_masking(tag_masking, metrics):
    if tag_masking in (True, False, '?'):
        tag_masking = {name: tag_masking for name in metrics}
    assert isinstance(tag_masking, dict) and tag_masking.keys() == metrics.keys(), (tag_masking, list(metrics))
    assert all(v in (True, False, '?') for v in tag_masking.values()), tag_masking
    return {name: [False, True] if v == '?' else [v] for name, v in tag_masking.items()}


def _masked(scores, audio_ids, event_classes, tags):
    new = copy.deepcopy(scores)
    for a in audio_ids:
        new[a][event_classes] *= tags[a]
    return new


def tune_tagging(tagging_scores, medfilt_length_candidates, metrics, minimize=False, storage_dir=None, device='cuda', verbose=True):
    """Median-filter length of the tagging scores per class and metric (reference :50-100)."""
    leaderboard = {}
    audio_ids = sorted(tagging_scores)
    _, event_classes = _columns(tagging_scores[audio_ids[0]])
    for medfilt_len in medfilt_length_candidates:
        if medfilt_len > 1:
            arrays = _on_device(tagging_scores, audio_ids, event_classes, device, lambda x: ops.medfilt(x, int(medfilt_len)))
            filtered = _replaced(tagging_scores, audio_ids, event_classes, arrays)
        else:
            filtered = tagging_scores
        for metric_name, metric_fn in metrics.items():
            leaderboard = _candidate(leaderboard, metric_name, metric_fn, filtered, {'medfilt_length': medfilt_len}, minimize, verbose,
                                     f'{metric_name}(medfilt_length={medfilt_len})')
    _store(leaderboard, storage_dir, 'tagging')
    if verbose:
        print('\nbest:')
        for metric_name in metrics:
            print(f'\n{metric_name} {leaderboard[metric_name][0]}')
    return leaderboard


def read_ground_truth_events(path):
    """``{audio_id: [(onset, offset, label)]}`` from a DESED-style TSV (filename, onset, offset, event_label; a clip without events
    is a row with empty fields)."""
    out = {}
    with open(path) as f:
        header = f.readline().rstrip('\n').split('\t')
        col = {name: header.index(name) for name in ('filename', 'onset', 'offset', 'event_label')}
        for line in f:
            fields = line.rstrip('\n').split('\t')
            if not fields or not fields[0]:
                continue
            audio_id = fields[col['filename']].rsplit('.', 1)[0]
            events = out.setdefault(audio_id, [])
            if len(fields) > col['event_label'] and fields[col['event_label']]:
                events.append((float(fields[col['onset']]), float(fields[col['offset']]), fields[col['event_label']]))
    return out


def boundaries_from_events(ground_truth):
    """Per clip and class ONE (first onset, last listed offset, label) - the boundary targets of the weakly supervised
    boundary detection (reference :103-122); ``ground_truth``: {audio_id: [(onset, offset, label)]} or the path of a TSV."""
    if isinstance(ground_truth, (str, os.PathLike)):
        ground_truth = read_ground_truth_events(ground_truth)
    out = {}
    for audio_id, event_list in ground_truth.items():
        spans = {}
        for onset, offset, label in event_list:
            spans[label] = (spans[label][0], offset) if label in spans else (onset, offset)
        out[audio_id] = [(onset, offset, label) for label, (onset, offset) in spans.items()]
    return out


def tune_boundaries_detection(detection_scores, medfilt_length_candidates, stepfilt_length_candidates, tags, metrics, minimize=False,
                              tag_masking=None, storage_dir=None, device='cuda', verbose=True):
    """Median-filter length x step-filter length x tag masking of the boundary scores (reference :125-207)."""
    masking = _tag_masking(tag_masking, metrics)
    leaderboard = {}
    audio_ids = sorted(detection_scores)
    _, event_classes = _columns(detection_scores[audio_ids[0]])
    for medfilt_len in medfilt_length_candidates:
        if medfilt_len > 1:
            arrays = _on_device(detection_scores, audio_ids, event_classes, device, lambda x: ops.medfilt(x, int(medfilt_len)))
            medfiltered = _replaced(detection_scores, audio_ids, event_classes, arrays)
        else:
            medfiltered = detection_scores
        for stepfilt_len in stepfilt_length_candidates:
            n = int(stepfilt_len)
            arrays = _on_device(medfiltered, audio_ids, event_classes, device,
                                lambda x: ops.boundariesfilt(x, n, want_f64=True) if n > 0 else ops.boundariesfilt(x, n))
            boundaries = _replaced(medfiltered, audio_ids, event_classes, arrays)
            masked = _masked(boundaries, audio_ids, event_classes, tags)
            for metric_name, metric_fn in metrics.items():
                for tag_masked in masking[metric_name]:
                    leaderboard = _candidate(
                        leaderboard, metric_name, metric_fn, masked if tag_masked else boundaries,
                        {'medfilt_length': medfilt_len, 'stepfilt_length': stepfilt_len, 'tag_masked': tag_masked}, minimize, verbose,
                        f'{metric_name}(medfilt_length={medfilt_len},stepfilt_length={stepfilt_len},tag_masked={tag_masked}):')
    _store(leaderboard, storage_dir, 'boundaries_detection')
    if verbose:
        print('\nbest:')
        for metric_name in metrics:
            print(f'\n{metric_name} :\n{leaderboard[metric_name][0]}')
    return leaderboard


def tune_sound_event_detection(detection_scores, medfilt_length_candidates, tags, metrics, minimize=False, tag_masking=None,
                               storage_dir=None, device='cuda', verbose=True):
    """Median-filter length x tag masking of the detection scores (reference :210-281)."""
    masking = _tag_masking(tag_masking, metrics)
    leaderboard = {}
    audio_ids = sorted(detection_scores)
    _, event_classes = _columns(detection_scores[audio_ids[0]])
    for medfilt_len in medfilt_length_candidates:
        if medfilt_len > 1:
            arrays = _on_device(detection_scores, audio_ids, event_classes, device, lambda x: ops.medfilt(x, int(medfilt_len)))
            filtered = _replaced(detection_scores, audio_ids, event_classes, arrays)
        else:
            filtered = detection_scores
        masked = _masked(filtered, audio_ids, event_classes, tags)
        for metric_name, metric_fn in metrics.items():
            for tag_masked in masking[metric_name]:
                leaderboard = _candidate(leaderboard, metric_name, metric_fn, masked if tag_masked else filtered,
                                         {'medfilt_length': medfilt_len, 'tag_masked': tag_masked}, minimize, verbose,
                                         f'{metric_name}(medfilt_length={medfilt_len},tag_masked={tag_masked}):')
    _store(leaderboard, storage_dir, 'sed', always_annotate=True)
    if verbose:
        print('\nbest:')
        for metric_name in metrics:
            print(f'\n{metric_name} :\n{leaderboard[metric_name][0]}')
    return leaderboard
