"""Ensemble inference driver: counterpart of ``pb_sed.models.base.inference``
(reference pb_sed/models/base/inference.py:12-222: ``tagging`` / ``boundaries_detection`` /
``sound_event_detection`` / ``inference``) with the whole post-processing chain on the GPU.

Per batch: every model's inference head runs on the device (HIP path), the scores stay there, and
mean over models -> sequence mask -> median filter -> (boundaries filter) are HIP kernels
that are bit-exact with the reference's numpy/scipy host code (tests/test_gpu_postproc.py replays the
reference's own golden vectors).  Only the final per-clip ``[(n,)T,K]`` arrays go to the host, keyed by
``example_id`` like the reference's result dict.  One batch is kept in flight: batch n + 1 is queued before the host
waits for batch n's scores (pinned buffer + event), so the per-clip host work overlaps the next batch's kernels.

Long clips: ``max_segment_length`` / ``segment_overlap`` / ``merge_score_segments`` / ``score_segment_overlap`` split
every batch into overlapping windows and merge the per-segment scores as the reference does
(pb_sed/models/base/inference.py:121-128,185-197 with pb_sed/utils/segment.py).  Not carried over: on-disk score storage.
"""
import numpy as np
import torch

from . import ops
from .utils.segment import is_last_segment, merge_segments, segment_batch


def _as_dev_scores(y, device):
    return y.detach().to(device=device, dtype=torch.float32).contiguous()


_COPY_STREAMS = {}      # device -> side stream of the host -> device hand-over


def _to_device_ahead(model, segment, device):
    """``model.example_to_device`` for a batch whose tensors lie in PINNED host memory: the copies go out on a side stream, so
    the hand-over of batch n + 1 (queued while batch n still computes - inference() keeps one batch in flight) runs beside
    batch n's kernels instead of in front of batch n + 1's; the compute stream waits for the copy's event.  Anything else
    (device-resident or pageable inputs) takes the plain path."""
    dev = torch.device(device)
    if dev.type != 'cuda' or not any(isinstance(v, torch.Tensor) and not v.is_cuda and v.is_pinned() for v in segment.values()):
        return model.example_to_device(segment, device)
    side = _COPY_STREAMS.get(str(dev))
    if side is None:
        side = _COPY_STREAMS[str(dev)] = torch.cuda.Stream(device=dev)
    main = torch.cuda.current_stream(dev)
    with torch.cuda.stream(side):
        out = model.example_to_device(segment, device)
        ready = torch.cuda.Event()
        ready.record(side)
    main.wait_event(ready)
    for v in out.values():
        if isinstance(v, torch.Tensor) and v.is_cuda:
            v.record_stream(main)            # allocated under the side stream, consumed on the compute stream
    return out


def _run_models(models, method, segment, kwargs):
    """[(scores, seq_len)] of every model for one segment.  Networks of one class and shape that offer ``<method>_jointly``
    (models/strong_label.py) run their recurrent scans in shared launches; anything else one model after the other, as
    the reference does (inference.py:133-141)."""
    cls = type(models[0])
    joint = getattr(cls, method + '_jointly', None)
    if joint is not None and not any(kwargs) and getattr(cls, 'can_run_jointly', lambda m: False)(models):
        return joint(models, segment)
    return [getattr(m, method)(segment, **kw) for m, kw in zip(models, kwargs)]


def filtering(scores, filter_fn, filter_length):
    """Device twin of inference.py:225-263: 0-d, per-class (1-d) or per-variant (2-d -> [B,n,K,T]) lengths."""
    filter_length = np.asarray(filter_length)
    b, *_, k, t = scores.shape
    if filter_length.ndim == 0:
        return filter_fn(scores, filter_length)
    if filter_length.ndim == 1:
        assert filter_length.shape[0] == k, filter_length.shape
        return filter_fn(scores, filter_length)                      # broadcasts over [..., K]
    assert filter_length.ndim == 2 and filter_length.shape[1] in (1, k), filter_length.shape
    n = filter_length.shape[0]
    if scores.dim() == 3:
        scores = scores[:, None].expand(b, n, k, t).contiguous()
    else:
        assert scores.shape[1] == n, (scores.shape, n)
    return filter_fn(scores, np.broadcast_to(filter_length, (n, k)))


def postprocess_enqueue(model_scores, seq_len, example_ids, *, medfilt_length=1, stepfilt_length=None,
                        apply_mask=False, masks=None, post_processing_fn=None):
    """Device half of inference.py:142-184 for one batch: ensemble mean -> sequence mask -> median filter ->
    (boundaries filter) as HIP kernels, then an asynchronous copy of the result into pinned host memory and an event
    behind it.  Nothing here waits for the device: ``postprocess_finish`` does, so a caller can enqueue the next batch
    first.  model_scores: list (one per model) of device tensors [B,(n,)K,T]; seq_len: host int array [B]."""
    dev = model_scores[0].device
    seq_len = np.asarray(seq_len)
    seq_dev = ops.host_to_device(seq_len, dev, torch.int32)
    s = ops.ensemble_mean_mask(model_scores, seq_dev)
    s = filtering(s, ops.medfilt, np.array(medfilt_length, dtype=int))
    if stepfilt_length is not None:
        sl_arr = np.array(stepfilt_length, dtype=int)
        if sl_arr.ndim == 0 and sl_arr > 0:        # the reference returns the float64 step-filter output here
            s = ops.boundariesfilt(s, sl_arr, want_f64=True)
        else:
            s = filtering(s, ops.boundariesfilt, sl_arr)
    slot = ops.pinned_buffer(s.shape, s.dtype, hold=True)
    slot[0].copy_(s, non_blocking=True)
    return {'slot': slot, 'done': ops.pinned_buffer_used(slot), 'seq_len': seq_len, 'example_ids': list(example_ids), 'apply_mask': apply_mask,
            'masks': masks, 'post_processing_fn': post_processing_fn}


def postprocess_finish(pending):
    """Host half: wait for the batch's copy, cut the per-clip ``[(n,)T,K]`` arrays (own memory: the pinned buffer is
    reused), apply the per-clip function and the tag masks.  Returns {example_id: np.ndarray}."""
    pending['done'].synchronize()
    s = np.array(pending['slot'][0].numpy())
    pending['slot'][2] = False
    post_processing_fn, masks = pending['post_processing_fn'], pending['masks']
    out = {}
    for i, (aid, sl) in enumerate(zip(pending['example_ids'], pending['seq_len'])):
        x = s[i, ..., :sl].swapaxes(-2, -1)
        out[aid] = x if post_processing_fn is None else post_processing_fn(x)
    apply_mask = np.array(pending['apply_mask'], dtype=bool)
    if apply_mask.any():
        assert masks is not None
        if apply_mask.ndim == 2:
            apply_mask = apply_mask[..., None, :]
        for aid in out:
            assert aid in masks, aid
            out[aid] = np.array(out[aid])
            out[aid] *= np.maximum(masks[aid], 1 - apply_mask)        # in place: stays float32
    return out


def postprocess_batch(model_scores, seq_len, example_ids, **kw):
    """inference.py:142-184 for one batch, both halves back to back.  Returns {example_id: np.ndarray [(n,)T,K]}."""
    return postprocess_finish(postprocess_enqueue(model_scores, seq_len, example_ids, **kw))


def create_score_dataframe(scores, timestamps, event_classes):
    """sed_scores_eval.utils.scores.create_score_dataframe restated: onset/offset columns + one per class."""
    import pandas as pd
    scores = np.asarray(scores)
    timestamps = np.asarray(timestamps)
    assert scores.ndim == 2 and len(timestamps) == scores.shape[0] + 1 and scores.shape[1] == len(event_classes)
    return pd.DataFrame(np.concatenate((timestamps[:-1, None], timestamps[1:, None], scores), axis=1),
                        columns=['onset', 'offset', *event_classes])


def scores_to_dataframes(scores, timestamps, event_classes):
    """inference.py:292-356 without the storage branch."""
    if isinstance(scores, np.ndarray):
        t, k = scores.shape
        assert len(timestamps) > t and len(event_classes) == k
        return create_score_dataframe(scores, timestamps[:t + 1], event_classes)
    ids = sorted(scores)
    if scores[ids[0]].ndim == 3:
        n = scores[ids[0]].shape[0]
        return [{a: scores_to_dataframes(scores[a][i], timestamps[a] if isinstance(timestamps, dict) else timestamps,
                                         event_classes) for a in ids} for i in range(n)]
    return {a: scores_to_dataframes(scores[a], timestamps[a] if isinstance(timestamps, dict) else timestamps,
                                    event_classes) for a in ids}


def inference(model, method, dataset, device, max_segment_length=None, segment_overlap=0,
              merge_score_segments=False, score_segment_overlap=None, model_kwargs=None, medfilt_length=1,
              stepfilt_length=None, apply_mask=False, masks=None, post_processing_fn=None, timestamps=None,
              event_classes=None, score_storage_dir=None, rank=0, world_size=1):
    """Same contract as the reference's ``inference`` (inference.py:86-222).  ``rank``/``world_size``
    shard every batch over clips (config 5: no collective needed, results are keyed by example_id)."""
    if score_storage_dir is not None:
        raise NotImplementedError('on-disk score storage is outside the MI355X hot path')
    models = list(model) if isinstance(model, (list, tuple)) else [model]
    model_kwargs = {} if model_kwargs is None else model_kwargs
    kwargs = list(model_kwargs) if isinstance(model_kwargs, (list, tuple)) else len(models) * [model_kwargs]
    assert len(kwargs) == len(models), (len(models), len(kwargs))
    for m in models:
        assert hasattr(m, method), (m, method)
        m.to(device)
        m.eval()
    scores = {}

    def enqueue(batch):
        """Everything of one batch that runs on the device, queued without waiting for it."""
        segments = [batch] if max_segment_length is None else segment_batch(batch, max_segment_length, segment_overlap)
        queued = []
        for segment in segments:
            segment = _to_device_ahead(models[0], segment, device)
            per_model, seq_len = [], None
            for y, sl in _run_models(models, method, segment, kwargs):
                per_model.append(_as_dev_scores(y, device))
                if seq_len is None:
                    seq_len = np.asarray(sl)
                else:
                    assert (np.asarray(sl) == seq_len).all(), (seq_len, sl)
            seg_masks = masks
            if masks is not None and len(segments) > 1:       # tags are keyed by clip, segments by clip + position
                seg_masks = {a: masks[a.split('_!segment!_')[0]] for a in segment['example_id']}
            queued.append(postprocess_enqueue(per_model, seq_len, segment['example_id'], medfilt_length=medfilt_length,
                                              stepfilt_length=stepfilt_length, apply_mask=apply_mask, masks=seg_masks,
                                              post_processing_fn=post_processing_fn))
        return queued, segments[-1]['example_id'][0], ops.gru_flags_snapshot()

    def finish(pending):
        queued, last_id, flags = pending
        cache = {}
        for q in queued:
            cache.update(postprocess_finish(q))
        ops.gru_flags_check(flags)      # copied behind the batch's scores: a timed-out persistent scan raises here
        if merge_score_segments and not is_last_segment(last_id):
            raise RuntimeError('a batch ended before its last segment')
        if merge_score_segments:
            cache = merge_segments(cache, segment_overlap if score_segment_overlap is None else score_segment_overlap)
        scores.update(cache)

    # one batch in flight: batch n + 1 is queued on the device before the host waits for batch n, so the per-clip host work
    # of a batch (and the copy behind it) overlaps the next batch's kernels
    pending = None
    with torch.no_grad():
        for batch in dataset:
            batch = {k: v for k, v in batch.items() if k not in ('weak_targets', 'boundary_targets', 'strong_targets')}
            if world_size > 1:
                from .trainer import shard_batch
                batch = shard_batch(batch, rank, world_size)
                if not len(batch['seq_len']):            # a ragged last batch with fewer clips than ranks: nothing for this rank
                    continue
            queued = enqueue(batch)
            if pending is not None:
                finish(pending)
            pending = queued
        if pending is not None:
            finish(pending)
    if timestamps is not None or event_classes is not None:
        assert timestamps is not None and event_classes is not None
        return scores_to_dataframes(scores, timestamps, event_classes)
    return scores


def gather_results(results, dst=None, group=None):
    """Merge the per-rank result dictionaries of a sharded ``inference(..., rank=r, world_size=n)`` pass ({example_id: scores /
    DataFrame / event list}; a list of such dictionaries - one per filter variant - is merged position by position).  Clips are
    independent, so the data path has no collective (config 5); this is the control-plane step behind it, for whatever needs all
    clips in one place - tuning, evaluation, pseudo labels: host objects through ``torch.distributed.all_gather_object`` (every rank
    gets the merged dictionary) or, with ``dst``, ``gather_object`` (only that rank; the others get None).  No process group / one
    rank: the input comes back.  An example_id reported by two ranks is an error (the shards must partition the batches)."""
    import torch.distributed as dist
    if not (dist.is_available() and dist.is_initialized()) or dist.get_world_size(group) == 1:
        return results
    world = dist.get_world_size(group)
    if dst is None:
        parts = [None] * world
        dist.all_gather_object(parts, results, group=group)
    else:
        parts = [None] * world if dist.get_rank(group) == dst else None
        dist.gather_object(results, parts, dst=dst, group=group)
        if parts is None:
            return None

    def merge(dicts):
        out = {}
        for d in dicts:
            twice = out.keys() & d.keys()
            if twice:
                raise ValueError(f'gather_results: {sorted(twice)[:3]} reported by more than one rank')
            out.update(d)
        return out
    if isinstance(results, (list, tuple)):
        assert all(len(p) == len(results) for p in parts), [len(p) for p in parts]
        return [merge([p[i] for p in parts]) for i in range(len(results))]
    return merge(parts)


def tagging(models, dataset, device, model_kwargs=None, medfilt_length=1, method='tagging', timestamps=None,
            event_classes=None, **kw):
    return inference(models, method, dataset, device, model_kwargs=model_kwargs, medfilt_length=medfilt_length,
                     post_processing_fn=lambda x: x.max(-2, keepdims=True), timestamps=timestamps,
                     event_classes=event_classes, **kw)


def boundaries_detection(models, dataset, device, model_kwargs=None, medfilt_length=1, stepfilt_length=0,
                         apply_mask=False, masks=None, method='boundaries_detection', timestamps=None,
                         event_classes=None, **kw):
    return inference(models, method, dataset, device, model_kwargs=model_kwargs, medfilt_length=medfilt_length,
                     stepfilt_length=stepfilt_length, apply_mask=apply_mask, masks=masks, timestamps=timestamps,
                     event_classes=event_classes, **kw)


def sound_event_detection(models, dataset, device, model_kwargs=None, medfilt_length=1, method='sound_event_detection',
                          apply_mask=False, masks=None, timestamps=None, event_classes=None, **kw):
    return inference(models, method, dataset, device, model_kwargs=model_kwargs, medfilt_length=medfilt_length,
                     apply_mask=apply_mask, masks=masks, timestamps=timestamps, event_classes=event_classes, **kw)


def load_hyper_params(hyper_params_dir, stage, names=('f',)):
    """The tuned hyper-parameters ``pb_sed_amd.tuning`` (or the reference's tuning scripts) wrote:
    ``<dir>/<stage>_hyper_params_<name>.json`` for stage 'tagging' / 'boundaries_detection' / 'sed'; one dict
    ``{event_class: {...}}`` per name (pb_sed/experiments/weak_label_crnn/inference.py:71,147,213-216)."""
    import json
    import os
    names = [names] if isinstance(names, str) else list(names)
    out = []
    for name in names:
        with open(os.path.join(str(hyper_params_dir), f'{stage}_hyper_params_{name}.json')) as f:
            out.append(json.load(f))
    return out


def sed_hyper_param_arrays(hyper_params, event_classes):
    """Per-variant, per-class arrays for ONE ensemble pass over several tuned parameter sets
    (pb_sed/experiments/weak_label_crnn/inference.py:239-262, pb_sed/experiments/strong_label_crnn/inference.py:117-122):
    ``medfilt_length`` / ``apply_mask`` [n, K] for ``sound_event_detection``; for FBCRNN parameter sets also
    ``model_kwargs = {'window_length': [n, K], 'window_shift': s}`` (one shift for all, as the reference demands) and the
    timestamp stride; ``thresholds``: per variant {class: threshold} or None where the metric has none (PSDS)."""
    hyper_params = [hyper_params] if isinstance(hyper_params, dict) else list(hyper_params)
    n, k = len(hyper_params), len(event_classes)
    medfilt, masked = np.zeros((n, k)), np.zeros((n, k))
    windowed = 'window_length' in hyper_params[0][event_classes[0]]
    window, shifts = np.zeros((n, k)), set()
    for i, hp in enumerate(hyper_params):
        for j, c in enumerate(event_classes):
            medfilt[i, j], masked[i, j] = hp[c]['medfilt_length'], hp[c]['tag_masked']
            if windowed:
                window[i, j] = hp[c]['window_length']
                shifts.add(hp[c]['window_shift'])
    out = {'medfilt_length': medfilt, 'apply_mask': masked, 'model_kwargs': None, 'timestamp_stride': 1,
           'thresholds': [{c: hp[c]['threshold'] for c in event_classes} if 'threshold' in hp[event_classes[0]] else None
                          for hp in hyper_params]}
    if windowed:
        if len(shifts) != 1:
            raise ValueError('Inference with multiple window shifts is not supported.')
        shift = int(shifts.pop())
        out['model_kwargs'] = {'window_length': window, 'window_shift': shift}
        out['timestamp_stride'] = shift
    return out


def tags_from_scores(tagging_scores, hyper_params, event_classes):
    """Clip-level tags of the tagging ensemble: score of each class against its tuned threshold
    (pb_sed/experiments/weak_label_crnn/inference.py:124-135).  ``tagging_scores``: {audio_id: [1, K]} as ``tagging`` returns
    them.  Returns (tags {audio_id: bool [K]}, scores {audio_id: [K]})."""
    thresholds = np.array([hyper_params[c]['threshold'] for c in event_classes])
    scores = {a: np.asarray(s)[0] for a, s in tagging_scores.items()}
    return {a: s > thresholds for a, s in scores.items()}, scores


def shift_and_widen_events(event_lists, pseudo_widening=0., onset_bias=None, offset_bias=None, decimals=None):
    """Boundary correction applied to detected events before they are written out as pseudo labels
    (pb_sed/experiments/strong_label_crnn/inference.py:177-184): onsets move earlier by ``pseudo_widening`` plus the
    class' tuned onset bias (clamped at 0 s), offsets later by ``pseudo_widening`` minus its offset bias; events that
    end up empty are dropped.  ``onset_bias`` / ``offset_bias``: {label: seconds} (missing labels: 0)."""
    onset_bias, offset_bias = onset_bias or {}, offset_bias or {}
    out = {}
    for clip_id, events in event_lists.items():
        kept = []
        for onset, offset, label in events:
            on = onset - pseudo_widening - onset_bias.get(label, 0)
            off = offset + pseudo_widening - offset_bias.get(label, 0)
            if decimals is not None:             # the boundaries path rounds to milliseconds (weak_label_crnn/inference.py:190-191)
                on, off = np.round(on, decimals), np.round(off, decimals)
            on = max(on, 0)
            if off > on:
                kept.append((on, off, label))
        out[clip_id] = kept
    return out


def scores_to_event_list(scores, thresholds, event_classes, timestamps, device='cuda'):
    """Threshold -> change points -> (onset, offset, label) per clip; the frame indices come from the
    ``pbsed_event_frames`` kernel (bit-exact index arithmetic), timestamps are looked up on the host.
    scores: {example_id: [T,K] array}; thresholds: scalar or [K]."""
    ids = sorted(scores)
    k = len(event_classes)
    thr = np.broadcast_to(np.asarray(thresholds, dtype=np.float32), (k,))
    tmax = max(scores[a].shape[0] for a in ids)
    dense = np.zeros((len(ids), k, tmax), np.float32)
    lens = np.zeros((len(ids),), np.int64)
    for i, a in enumerate(ids):
        t = scores[a].shape[0]
        dense[i, :, :t] = np.asarray(scores[a], dtype=np.float32).T
        lens[i] = t
    ev, cnt = ops.event_frames(torch.from_numpy(dense).to(device), thr[None, :], lens[:, None])      # end of the job: may wait
    ev, cnt = ev.cpu().numpy().reshape(len(ids), k, -1, 2), cnt.cpu().numpy().reshape(len(ids), k)
    out = {}
    for i, a in enumerate(ids):
        ts = timestamps[a] if isinstance(timestamps, dict) else timestamps
        events = []
        for c, label in enumerate(event_classes):
            for e in range(cnt[i, c]):
                events.append((float(ts[ev[i, c, e, 0]]), float(ts[ev[i, c, e, 1]]), label))
        out[a] = sorted(events)
    return out
