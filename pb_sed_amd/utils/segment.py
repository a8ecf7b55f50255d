"""Long-clip segmenting for inference: counterpart of ``pb_sed.utils.segment`` (reference pb_sed/utils/segment.py:6-71,
used by pb_sed/models/base/inference.py:121-128,185-197).

``segment_batch`` cuts a batch whose longest clip exceeds ``max_length`` frames into windows of ``max_length`` frames
every ``max_length - overlap`` frames; ``merge_segments`` glues the per-segment score arrays back together, dropping
half of every overlap on either side of a seam.  Example ids carry the segment index the way the reference writes them
(``<id>_!segment!_<i>_<n>``), so score dictionaries are interchangeable with the reference's.

Two input contracts: the reference's ``'stft'`` tensor ``[B, 1, T, bins, 2]`` is cut along its frame axis; the fused
front-end's ``'audio_data'`` ``[B, N]`` is cut into the sample ranges whose STFT frames are exactly those frames (a frame
t covers samples [320 t - 320, 320 t + 640) of the clip), with ``'stft_pad_front'`` telling the front-end kernel how much
of the reference's half-window fade-in padding precedes the slice (320 samples for the first segment, none afterwards):
both give the same features as segmenting the STFT of the whole clip.
"""
from math import ceil

import numpy as np

SEGMENT_TAG = '_!segment!_'
SHIFT, WINDOW = 320, 960            # STFT hop / window of the reference front-end (provider.py:315-323)


def _frames(batch):
    if 'stft' in batch:
        return int(batch['stft'].shape[2])
    from ..modules import num_frames
    return num_frames(int(batch['audio_data'].shape[-1]))


def segment_batch(batch, max_length, overlap, keys=('stft', 'audio_data')):
    """Returns a list of batches.  A batch whose clips all fit into ``max_length`` frames is returned as is."""
    seq_len = [int(v) for v in batch['seq_len']]
    if max(seq_len) <= max_length:
        return [batch]
    shift = max_length - overlap
    assert shift > 0, (max_length, overlap)
    t_all = _frames(batch)
    n_seg = max(ceil((t_all - max_length) / shift), 0) + 1
    segments = []
    for i in range(n_seg):
        start = i * shift
        lens = [min(max_length, sl - start) for sl in seq_len]
        t_seg = max(lens)
        seg = {k: v for k, v in batch.items() if k not in keys and k not in ('example_id', 'seq_len')}
        seg['example_id'] = [f'{a}{SEGMENT_TAG}{i}_{n_seg}' for a in batch['example_id']]
        seg['seq_len'] = lens
        seg['segment_start'], seg['segment_stop'] = start, start + max_length
        if 'stft' in batch and 'stft' in keys:
            seg['stft'] = batch['stft'][:, :, start:start + t_seg]
        elif 'audio_data' in batch:
            audio = batch['audio_data']
            audio = audio.reshape(audio.shape[0], -1)
            lo = start * SHIFT - (WINDOW - SHIFT) // 2          # first sample of frame `start`
            hi = (start + t_seg - 1) * SHIFT - (WINDOW - SHIFT) // 2 + WINDOW
            seg['audio_data'] = audio[:, max(lo, 0):min(hi, audio.shape[1])]
            seg['stft_pad_front'] = max(-lo, 0)
            seg['num_frames'] = t_seg
        segments.append(seg)
    return segments


def is_last_segment(example_id):
    if SEGMENT_TAG not in example_id:
        return True
    i, n = example_id.split(SEGMENT_TAG)[-1].split('_')
    return int(i) == int(n) - 1


def merge_segments(segmental_output, segment_overlap):
    """{'<id>_!segment!_<i>_<n>': [..., T_i, K]} -> {'<id>': [..., T, K]}; entries without a segment tag pass through.
    At every seam the first segment loses ceil(overlap / 2) trailing frames and the second overlap // 2 leading ones."""
    merged = {}
    drop_tail, drop_head = ceil(segment_overlap / 2), segment_overlap // 2
    for key in sorted(segmental_output):
        if SEGMENT_TAG not in key:
            merged[key] = segmental_output[key]
            continue
        audio_id, pos = key.split(SEGMENT_TAG)
        i, n = (int(v) for v in pos.split('_'))
        if i != 0:
            continue                                   # assembled when its first segment comes up
        parts = []
        for j in range(n):
            arr = segmental_output[f'{audio_id}{SEGMENT_TAG}{j}_{n}']
            t = arr.shape[-2]
            lo = drop_head if (j > 0 and segment_overlap > 0) else 0
            hi = t - drop_tail if (j < n - 1 and segment_overlap > 0) else t
            parts.append(arr[..., lo:max(hi, lo), :] if hi >= 0 else arr[..., :0, :])
        merged[audio_id] = np.concatenate(parts, axis=-2)
    return merged
