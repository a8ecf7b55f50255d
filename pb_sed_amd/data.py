"""GPU side of the data front-end in front of the model: counterparts of the reference-owned pieces of
``pb_sed.data_preparation`` that touch samples or frames.

* ``SuperposeEvents`` / ``scale_and_mix`` - scale + superposition mixing (pb_sed/data_preparation/mix.py:67-155,
  provider.py:195-215): the offsets are drawn on the host exactly as the reference draws them (same ``np.random`` calls in
  the same order, so a seeded run reproduces the reference's mixtures), the waveforms are mixed by one launch
  (``pbsed_mix_clips``) from a pool that stays resident in HBM.
* ``encode_targets`` - weak / boundary / strong target tensors from event lists (pb_sed/data_preparation/transform.py:56-124)
  in one launch (``pbsed_encode_targets``), bit-exact with the reference's encoding.

* ``TimeWarp`` - the time-warped STFT of the training transform (pb_sed/data_preparation/transform.py:36-45, samplers
  provider.py:329-338): the host draws one (anchor, anchor shift) pair per clip and turns it into the first sample of every
  frame's window; the fused front-end computes the frames there (``pbsed_logmel_fwd_frames``), event boundaries are moved
  with the same map.
* ``DynamicBucketBatcher`` / ``collate`` - batches of similar length (pb_sed/data_preparation/fetcher.py:38-51): host logic,
  no device work.

Sample -> frame alignment of event boundaries belongs to padertorch's STFT, the warp to padertorch's TimeWarpedSTFT and the
bucket rule to lazy_dataset / padertorch's DynamicExtendedTimeSeriesBucket (all absent here, parity unpinned); the rules used
are stated at ``samples_to_frames``, ``TimeWarp`` and ``DynamicBucketBatcher``.  Reading audio files and the lazy_dataset
plumbing stay host-side concerns of the caller.
"""
import struct
from math import ceil

import numpy as np
import torch

from ._lib import call, ptr, stream

LABEL_TYPES = {'weak': 0, 'boundaries': 1, 'strong': 2}


def add_label_types(example):
    """pb_sed/data_preparation/utils.py:1-30: defaults for strongly / weakly / un-labelled examples."""
    if 'events_start_samples' in example or 'events_stop_samples' in example:
        assert 'events' in example and 'events_start_samples' in example and 'events_stop_samples' in example, example.keys()
        example.setdefault('label_types', len(example['events']) * ['strong'])
        example.setdefault('unlabeled', False)
    elif 'events' in example:
        n = example['audio_data'].shape[-1]
        example['events_start_samples'] = [0 for _ in example['events']]
        example['events_stop_samples'] = [n for _ in example['events']]
        example.setdefault('label_types', len(example['events']) * ['weak'])
        example.setdefault('unlabeled', False)
    else:
        example.update(events=[], events_start_samples=[], events_stop_samples=[], label_types=[], unlabeled=True)
    return example


def samples_to_frames(start_samples, stop_samples, shift=320):
    """Event boundaries in samples -> STFT frames for the reference front-end (shift 320, window 960, 'half' fading:
    frame t is centred on sample 320 t + 160).  Restated rule (padertorch's STFT alignment is absent): an event covers
    the frames whose hop interval [320 t, 320 t + 320) it touches - onset floor(start / shift), offset ceil(stop / shift)."""
    start = [int(s) // shift for s in start_samples]
    stop = [int(ceil(int(s) / shift)) for s in stop_samples]
    return start, stop


class SuperposeEvents:
    """Same constructor and result dict as the reference's ``SuperposeEvents`` (mix.py:67-155); ``audio_data`` of the
    components are device tensors [1, N] (fp32) and so is the mixture's."""

    def __init__(self, min_overlap=1., max_length_in_samples=None, fade_length=0, label_key='events'):
        self.min_overlap, self.max_length_in_samples = min_overlap, max_length_in_samples
        self.fade_length, self.label_key = fade_length, label_key

    def place(self, lengths):
        """Start / stop sample of every component inside the mixture (first component anchors; the others are drawn
        uniformly among the offsets that keep ``min_overlap`` of the shorter signal overlapping, mix.py:95-117)."""
        starts, stops = [0], [lengths[0]]
        for n in lengths[1:]:
            min_overlap = int(np.ceil(min(n, lengths[0]) * self.min_overlap))
            lo, hi = -(n - min_overlap), lengths[0] - min_overlap
            if self.max_length_in_samples is not None:
                assert n <= self.max_length_in_samples, (n, self.max_length_in_samples)
                lo = max(lo, max(stops) - self.max_length_in_samples)
                hi = min(hi, min(starts) + self.max_length_in_samples - n)
            starts.append(int(np.floor(lo + np.random.rand() * (hi - lo + 1))))
            stops.append(starts[-1] + n)
        starts, stops = np.array(starts), np.array(stops)
        return starts - starts.min(), stops - starts.min()

    def __call__(self, components):
        assert len(components) > 0
        components = [add_label_types(c) for c in components]
        starts, stops = self.place([c['audio_data'].shape[1] for c in components])
        audio = mix_clips([[(c['audio_data'], int(s), 1.) for c, s in zip(components, starts)]], self.fade_length)[0]
        key = self.label_key
        return {
            'example_id': '+'.join(c['example_id'] for c in components),
            'dataset': '+'.join(sorted({c['dataset'] for c in components})),
            'audio_data': audio[None, :int(stops.max())],
            'seq_len': int(stops.max()),
            key: [e for c in components for e in c[key]],
            f'{key}_start_samples': [int(v + s) for c, s in zip(components, starts) for v in c[f'{key}_start_samples']],
            f'{key}_stop_samples': [int(v + s) for c, s in zip(components, starts) for v in c[f'{key}_stop_samples']],
            'label_types': [t for c in components for t in c['label_types']],
            'unlabeled': any(c['unlabeled'] for c in components),
        }


def mix_clips(mixtures, fade_length=0):
    """``mixtures``: per output clip a list of (waveform device tensor [1, N] or [N], start sample, gain).  Returns a
    device tensor [B, n_out] (n_out = longest mixture, shorter ones zero padded): ONE launch for the whole batch."""
    wavs, comps, first = [], [], [0]
    offset, n_out, device = 0, 0, None
    for clip in mixtures:
        total = max(int(s) + w.reshape(-1).shape[0] for w, s, _ in clip)
        n_out = max(n_out, total)
        for w, s, gain in clip:
            w = w.reshape(-1)
            device = w.device
            n = w.shape[0]
            comps.append(struct.pack('<qiifiii', offset, n, int(s), float(gain), int(s > 0), int(int(s) + n < total), 0))
            wavs.append(w.to(torch.float32))
            offset += n
        first.append(len(comps))
    pool = wavs[0] if len(wavs) == 1 else torch.cat(wavs)
    desc = torch.frombuffer(bytearray(b''.join(comps)), dtype=torch.uint8).to(device)
    first_dev = torch.tensor(first, dtype=torch.int32).to(device)
    out = torch.empty((len(mixtures), n_out), device=device, dtype=torch.float32)
    call('pbsed_mix_clips', ptr(pool.contiguous()), ptr(desc), ptr(first_dev), ptr(out), len(mixtures), n_out, int(fade_length), stream())
    return out


def encode_targets(examples, label_mapping, num_frames, device, seq_len=None, boundary=True, strong=True, shift=320):
    """Event lists of a batch -> (weak [B,K], boundary [B,K,T] | None, strong [B,K,T] | None) on ``device``.

    ``examples``: dicts with ``events`` (labels), ``events_start_samples`` / ``events_stop_samples`` (or ``*_frames``),
    ``label_types`` ('weak' | 'boundaries' | 'strong') and ``unlabeled`` as ``add_label_types`` leaves them;
    ``label_mapping``: {label: class index}; ``seq_len``: frames per clip (default ``num_frames`` for all)."""
    b, k = len(examples), len(label_mapping)
    rows, first, unl = [], [0], []
    for ex in examples:
        ex = add_label_types(dict(ex))
        if 'events_start_frames' in ex:
            start, stop = ex['events_start_frames'], ex['events_stop_frames']
        else:
            start, stop = samples_to_frames(ex['events_start_samples'], ex['events_stop_samples'], shift)
        for label, a, o, typ in zip(ex['events'], start, stop, ex['label_types']):
            rows.append(struct.pack('<iiii', int(label_mapping[label]), max(int(a), 0), int(o), LABEL_TYPES[typ]))
        first.append(len(rows))
        unl.append(int(bool(ex['unlabeled'])))
    ev = torch.frombuffer(bytearray(b''.join(rows) or bytes(16)), dtype=torch.uint8).to(device)
    first_dev = torch.tensor(first, dtype=torch.int32).to(device)
    unl_dev = torch.tensor(unl, dtype=torch.int32).to(device)
    seq = torch.tensor([num_frames] * b if seq_len is None else [int(v) for v in seq_len], dtype=torch.int32).to(device)
    weak = torch.empty((b, k), device=device, dtype=torch.float32)
    bnd = torch.empty((b, k, num_frames), device=device, dtype=torch.float32) if boundary else None
    strg = torch.empty((b, k, num_frames), device=device, dtype=torch.float32) if strong else None
    call('pbsed_encode_targets', ptr(ev), ptr(first_dev), ptr(unl_dev), ptr(seq), ptr(weak), ptr(bnd), ptr(strg), b, k,
         int(num_frames), stream())
    return weak, bnd, strg


class TimeWarp:
    """Piecewise-linear warp of a clip's time axis, one draw per clip: the point at ``anchor`` (fraction of the clip, drawn by
    ``anchor_sampling_fn``, reference default U(0.4, 0.6)) is moved to ``anchor + shift`` (``anchor_shift_sampling_fn``,
    U(-0.1, 0.1)); both parts are stretched / compressed linearly, the number of frames stays that of the base STFT.

    Restated rule (padertorch's TimeWarpedSTFT is absent, parity unpinned): output frame t is the base STFT's windowed FFT
    taken around the source sample ``src(c_t)``, where c_t = 320 t + 160 is the frame's centre on the regular grid and
    src(u) = u a / (a + s) for u < (a + s) N, a N + (u - (a + s) N) (1 - a) / (1 - a - s) above; event boundaries are moved by
    the inverse map, so a frame's label still describes the audio under its window."""

    def __init__(self, anchor_sampling_fn, anchor_shift_sampling_fn, shift=320, window_length=960):
        assert callable(anchor_sampling_fn) and callable(anchor_shift_sampling_fn)
        self.anchor_sampling_fn, self.anchor_shift_sampling_fn = anchor_sampling_fn, anchor_shift_sampling_fn
        self.shift, self.window_length = shift, window_length

    def sample(self, n):
        """-> (anchor [n], anchor shift [n]); the moved anchor is kept inside (0.05, 0.95)."""
        a = np.asarray(self.anchor_sampling_fn((n,)), dtype=np.float64)
        s = np.asarray(self.anchor_shift_sampling_fn((n,)), dtype=np.float64)
        return a, np.clip(a + s, .05, .95) - a

    @staticmethod
    def source_of(u, n_samples, a, s):
        """warped (output) sample position(s) -> source sample position(s)."""
        u = np.asarray(u, dtype=np.float64)
        knee = (a + s) * n_samples
        return np.where(u < knee, u * a / (a + s), a * n_samples + (u - knee) * (1. - a) / (1. - a - s))

    @staticmethod
    def warped_of(v, n_samples, a, s):
        """source sample position(s) -> warped (output) position(s): the inverse of ``source_of``."""
        v = np.asarray(v, dtype=np.float64)
        knee = a * n_samples
        return np.where(v < knee, v * (a + s) / a, (a + s) * n_samples + (v - knee) * (1. - a - s) / (1. - a))

    def frame_positions(self, n_samples, num_frames, anchors, shifts):
        """-> int32 [B, num_frames]: first sample of each frame's window in every clip (clips of ``n_samples``; an array
        gives each clip its own length)."""
        n = np.broadcast_to(np.asarray(n_samples, dtype=np.float64), np.shape(anchors))[:, None]
        centre = (np.arange(num_frames, dtype=np.float64) * self.shift + self.shift / 2)[None]
        src = self.source_of(centre, n, np.asarray(anchors)[:, None], np.asarray(shifts)[:, None])
        return np.rint(src - self.window_length / 2).astype(np.int32)

    def __call__(self, examples, num_frames):
        """Draws a warp per example -> (frame_pos int32 [B, num_frames], examples with ``events_*_samples`` moved to the
        warped time axis - feed them to ``encode_targets``)."""
        a, s = self.sample(len(examples))
        n = np.array([ex['audio_data'].shape[-1] for ex in examples], dtype=np.float64)
        out = []
        for i, ex in enumerate(examples):
            ex = add_label_types(dict(ex))
            ex['events_start_samples'] = [int(np.floor(self.warped_of(v, n[i], a[i], s[i]))) for v in ex['events_start_samples']]
            ex['events_stop_samples'] = [int(np.ceil(self.warped_of(v, n[i], a[i], s[i]))) for v in ex['events_stop_samples']]
            out.append(ex)
        return self.frame_positions(n, num_frames, a, s), out


class DynamicBucketBatcher:
    """Batches of examples with similar ``seq_len`` out of a stream (pb_sed/data_preparation/fetcher.py:38-51:
    ``batch_dynamic_bucket(DynamicExtendedTimeSeriesBucket, batch_size, max_padding_rate, len_key='seq_len',
    min_label_diversity, min_dataset_examples, expiration, max_buffered_examples, drop_incomplete, sort_key='seq_len',
    reverse_sort=True)``).

    Restated rule (lazy_dataset / padertorch buckets are absent, parity unpinned): an example joins the first open bucket
    whose length bounds [lower, upper] contain its length; a bucket opened by an example of length L starts with
    lower = L (1 - r), upper = L / (1 - r) for max_padding_rate r and tightens both with every member, so no member is
    padded by more than a fraction r of the longest; a full bucket (``batch_size``) is emitted sorted by length, longest
    first.  ``min_label_diversity``: a full bucket is only emitted once its members' ``label_key`` targets cover at least
    that many distinct classes (else it keeps waiting for a replacement and the example that does not add a class is sent to
    another bucket); ``min_dataset_examples`` {dataset: count}: same for examples per ``dataset``.  A bucket older than
    ``expiration`` examples, or the oldest one when more than ``max_buffered_examples`` wait, is closed: emitted as it is
    unless ``drop_incomplete``.  At the end of the stream the open buckets are flushed the same way."""

    def __init__(self, batch_size, max_padding_rate=.05, len_key='seq_len', min_label_diversity=0, label_key='weak_targets',
                 min_dataset_examples=None, expiration=None, max_buffered_examples=None, drop_incomplete=False,
                 sort_key='seq_len', reverse_sort=True):
        self.batch_size, self.max_padding_rate, self.len_key = batch_size, max_padding_rate, len_key
        self.min_label_diversity, self.label_key = min_label_diversity, label_key
        self.min_dataset_examples = dict(min_dataset_examples or {})
        self.expiration, self.max_buffered_examples = expiration, max_buffered_examples
        self.drop_incomplete, self.sort_key, self.reverse_sort = drop_incomplete, sort_key, reverse_sort

    class _Bucket:
        def __init__(self, born):
            self.members, self.lower, self.upper, self.born = [], 0., float('inf'), born

    def _classes(self, members):
        seen = set()
        for ex in members:
            seen.update(np.flatnonzero(np.asarray(ex[self.label_key]).reshape(-1, np.shape(ex[self.label_key])[-1]).max(0) > .5).tolist())
        return seen

    def _accepts(self, bucket, ex):
        n = ex[self.len_key]
        if not (bucket.lower <= n <= bucket.upper):
            return False
        room = self.batch_size - len(bucket.members) - 1                 # places left after this example
        if self.min_dataset_examples:
            have = {}
            for m in bucket.members + [ex]:
                have[m.get('dataset')] = have.get(m.get('dataset'), 0) + 1
            if sum(max(c - have.get(d, 0), 0) for d, c in self.min_dataset_examples.items()) > room:
                return False
        if self.min_label_diversity:
            if self.min_label_diversity - len(self._classes(bucket.members + [ex])) > room:
                return False
        return True

    def _emit(self, bucket):
        members = bucket.members
        if self.sort_key is not None:
            members = sorted(members, key=lambda ex: ex[self.sort_key], reverse=self.reverse_sort)
        return members

    def __call__(self, examples):
        """Generator over lists of examples (feed each to ``collate``)."""
        buckets, seen, r = [], 0, self.max_padding_rate
        for ex in examples:
            seen += 1
            for bucket in buckets:
                if self._accepts(bucket, ex):
                    break
            else:
                bucket = self._Bucket(seen)
                buckets.append(bucket)
            n = ex[self.len_key]
            bucket.members.append(ex)
            bucket.lower, bucket.upper = max(bucket.lower, n * (1. - r)), min(bucket.upper, n / (1. - r))
            if len(bucket.members) == self.batch_size:
                buckets.remove(bucket)
                yield self._emit(bucket)
            closing = [b for b in buckets if self.expiration is not None and seen - b.born >= self.expiration]
            if self.max_buffered_examples is not None:
                while sum(len(b.members) for b in buckets if b not in closing) > self.max_buffered_examples:
                    closing.append(next(b for b in buckets if b not in closing))
            for b in closing:
                buckets.remove(b)
                if not self.drop_incomplete:
                    yield self._emit(b)
        for b in buckets:
            if not self.drop_incomplete:
                yield self._emit(b)


def collate(batch, pad_keys=('audio_data', 'stft', 'boundary_targets', 'strong_targets'), time_axis=None):
    """List of example dicts -> dict of lists, with the array entries named in ``pad_keys`` zero padded along their time
    axis to the longest and stacked (padertorch's Collate as the reference's fetcher uses it, fetcher.py:51).  The time axis
    of an entry is its last one, except ``stft`` [1, T, bins, 2] (axis 1) - or ``time_axis[key]``."""
    out = {k: [ex[k] for ex in batch] for k in batch[0]}
    axes = {'stft': 1, **(time_axis or {})}
    for k in pad_keys:
        if k not in out or not isinstance(out[k][0], (np.ndarray, torch.Tensor)):
            continue
        ax = axes.get(k, -1)
        longest = max(v.shape[ax] for v in out[k])
        padded = []
        for v in out[k]:
            v = torch.as_tensor(v)
            pad = [0, 0] * v.dim()
            pad[2 * (v.dim() - 1 - (ax % v.dim())) + 1] = longest - v.shape[ax]
            padded.append(torch.nn.functional.pad(v, pad))
        out[k] = torch.stack(padded)
    return out


class DevicePrefetcher:
    """Iterate over host batches (dicts of tensors / arrays / lists; tensors in PINNED memory for the copies to be
    asynchronous) with the host -> device hand-over of batch n + 1 in flight on a side stream while the caller trains on
    batch n: a 10 s x 32-clip batch is 21 MB = 0.45 ms of PCIe time, 5 % of a train step if it runs in front of the step on
    the compute stream (`h2d`).  The compute stream waits for a batch's copy event when the batch is
    handed out.  On a CPU device the batches pass through unchanged.

        for batch in DevicePrefetcher(loader, 'cuda:0'):
            trainer.step(batch)
    """

    def __init__(self, iterable, device):
        self.iterable, self.device = iterable, torch.device(device)
        self._side = None

    def _stage(self, batch):
        if batch is None:
            return None
        if self.device.type != 'cuda':
            return batch, None
        if self._side is None:
            self._side = torch.cuda.Stream(device=self.device)
        with torch.cuda.stream(self._side):
            out = {k: (v.to(self.device, non_blocking=True) if isinstance(v, torch.Tensor) and not v.is_cuda else v)
                   for k, v in batch.items()}
            ready = torch.cuda.Event()
            ready.record(self._side)
        return out, ready

    def __iter__(self):
        it = iter(self.iterable)
        staged = self._stage(next(it, None))
        while staged is not None:
            batch, ready = staged
            if ready is not None:
                main = torch.cuda.current_stream(self.device)
                main.wait_event(ready)
                for v in batch.values():
                    if isinstance(v, torch.Tensor) and v.is_cuda:
                        v.record_stream(main)        # allocated under the side stream, consumed on the compute stream
            staged = self._stage(next(it, None))     # batch n + 1 starts travelling before the caller works on batch n
            yield batch
