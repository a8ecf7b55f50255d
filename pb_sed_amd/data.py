"""GPU side of the data front-end in front of the model: counterparts of the reference-owned pieces of
``pb_sed.data_preparation`` that touch samples or frames.

* ``SuperposeEvents`` / ``scale_and_mix`` - scale + superposition mixing (pb_sed/data_preparation/mix.py:67-155,
  provider.py:195-215): the offsets are drawn on the host exactly as the reference draws them (same ``np.random`` calls in
  the same order, so a seeded run reproduces the reference's mixtures), the waveforms are mixed by one launch
  (``pbsed_mix_clips``) from a pool that stays resident in HBM.
* ``encode_targets`` - weak / boundary / strong target tensors from event lists (pb_sed/data_preparation/transform.py:56-124)
  in one launch (``pbsed_encode_targets``), bit-exact with the reference's encoding.

Sample -> frame alignment of event boundaries belongs to padertorch's STFT (absent here, parity unpinned); the rule used by
``samples_to_frames`` is stated there.  Reading audio files, the lazy_dataset plumbing and bucketing stay host-side
concerns of the caller.
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
