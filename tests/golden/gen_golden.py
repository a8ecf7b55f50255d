"""Generate golden vectors by executing the REFERENCE's own source under import shims.

Runs ONLY in the build container (needs /root/reference); never at test/bench time.  Third-party
packages the reference imports (padertorch, paderbox, sed_scores_eval, torchvision, ...) are absent,
so they are replaced by stub modules; the few third-party helpers the reference-owned maths calls
(compute_mask, TakeLast/Mean/Sum/Max, Pad) are the oracle's restatements.  What is
pinned is therefore the reference-OWNED code:  pb_sed/models/weak_label/crnn.py (review, losses,
heads), pb_sed/models/strong_label/crnn.py (review), pb_sed/filters.py,
pb_sed/models/base/inference.py (inference, filtering, boundariesfilt), pb_sed/evaluation/instance_based.py.

Usage:  PYTHONDONTWRITEBYTECODE=1 python tests/golden/gen_golden.py   ->  tests/golden/ref_*.npz
No reference source is copied; only input/output arrays are stored.
"""
import importlib.abc
import importlib.machinery
import os
import sys
import types

sys.dont_write_bytecode = True
REPO = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, REPO)

import numpy as np
import scipy.signal  # noqa: F401  (import before aliasing np.bool, see SURVEY 8c gotchas)
import torch

np.int = int      # reference uses removed numpy aliases (inference.py:103-104, crnn.py:252)
np.bool = bool

from oracle import nn as onn                     # noqa: E402
from oracle.frontend import compute_mask          # noqa: E402
from tests.stubs import StubRNN, StubFeatures, StubCNN  # noqa: E402

STUB_ROOTS = ('padertorch', 'paderbox', 'sed_scores_eval', 'torchvision', 'lazy_dataset', 'sacred',
              'codecarbon', 'tensorboardX')


class _Anything:
    def __init__(self, *a, **k):
        pass

    def __call__(self, *a, **k):
        return _Anything()

    def __getattr__(self, name):
        return _Anything()


class _StubModule(types.ModuleType):
    __path__ = []

    def __getattr__(self, name):
        if name.startswith('__'):
            raise AttributeError(name)
        return _Anything


class _Finder(importlib.abc.MetaPathFinder, importlib.abc.Loader):
    def find_spec(self, fullname, path, target=None):
        if fullname.split('.')[0] in STUB_ROOTS:
            return importlib.machinery.ModuleSpec(fullname, self, is_package=True)
        return None

    def create_module(self, spec):
        return _StubModule(spec.name)

    def exec_module(self, module):
        pass


sys.meta_path.insert(0, _Finder())
import padertorch                                              # noqa: E402
import padertorch.ops.sequence.mask as pt_mask                 # noqa: E402
import padertorch.contrib.je.modules.reduce as pt_reduce       # noqa: E402
import padertorch.contrib.je.modules.conv as pt_conv           # noqa: E402
import paderbox.array as pb_array                              # noqa: E402
import paderbox.array.segment as pb_segment                    # noqa: E402


class _Model(torch.nn.Module):
    def example_to_device(self, example, device=None):
        return example


def _segment_axis(x, length, shift, axis=-1, end='cut', pad_mode=None):
    x = np.asarray(x)
    axis %= x.ndim
    xm = np.moveaxis(x, axis, 0)
    n = (xm.shape[0] - length) // shift + 1
    out = np.stack([xm[i * shift:i * shift + length] for i in range(n)])  # [n, length, ...]
    return np.moveaxis(out, (0, 1), (axis, axis + 1))


class _Segmenter:
    """padertorch.data.segment.Segmenter restated for the one way the reference uses it (utils/segment.py:26-31:
    fixed length / shift along one axis, zero padding so that the tail is covered, 'example_id' copied): a list of
    dicts with the segmented arrays + 'segment_start' / 'segment_stop'."""

    def __init__(self, length, shift, include_keys, copy_keys=(), axis=-1, mode='constant', padding=True):
        self.length, self.shift, self.include_keys, self.copy_keys, self.axis = length, shift, include_keys, copy_keys, axis

    def __call__(self, example):
        t = np.asarray(example[self.include_keys[0]]).shape[self.axis]
        n = max(-(-(t - self.length) // self.shift), 0) + 1
        out = []
        for i in range(n):
            a, b = i * self.shift, i * self.shift + self.length
            seg = {'segment_start': a, 'segment_stop': b}
            for k in self.include_keys:
                x = example[k]
                is_t = isinstance(x, torch.Tensor)
                x = np.asarray(x)
                piece = np.take(x, np.arange(a, min(b, t)), axis=self.axis)
                if b > t:
                    pad = [(0, 0)] * x.ndim
                    pad[self.axis] = (0, b - t)
                    piece = np.pad(piece, pad)
                seg[k] = torch.from_numpy(piece) if is_t else piece
            for k in self.copy_keys:
                seg[k] = example[k]
            out.append(seg)
        return out


import padertorch.data.segment as pt_segment                   # noqa: E402
pt_segment.Segmenter = _Segmenter
padertorch.Model = _Model
pt_mask.compute_mask = compute_mask
for _n in ('TakeLast', 'Mean', 'Sum', 'Max'):
    setattr(pt_reduce, _n, getattr(onn, _n))
pt_conv.Pad = onn.Pad
pb_array.segment_axis = _segment_axis
pb_segment.segment_axis = _segment_axis


def _validate_score_dataframe(scores, timestamps=None, event_classes=None):
    """sed_scores_eval.utils.scores.validate_score_dataframe restated for what pb_sed/models/base/tuning.py uses of it: the frame
    boundaries and the class columns of a score DataFrame ('onset', 'offset', then one column per class)."""
    names = list(scores.columns)
    assert names[:2] == ['onset', 'offset'], names
    if event_classes is not None:
        assert list(event_classes) == names[2:], (event_classes, names)
    return np.concatenate((scores['onset'].to_numpy(), scores['offset'].to_numpy()[-1:])), names[2:]


import sed_scores_eval.utils.scores as sse_scores               # noqa: E402
sse_scores.validate_score_dataframe = _validate_score_dataframe

sys.path.insert(0, '/root/reference')
from pb_sed.models.weak_label.crnn import CRNN as RefFBCRNN     # noqa: E402
from pb_sed.models.strong_label.crnn import CRNN as RefBiCRNN   # noqa: E402
from pb_sed import filters as ref_filters                       # noqa: E402
import pb_sed.models.base                                      # noqa: E402
ref_inf = sys.modules['pb_sed.models.base.inference']   # (the package re-exports a function of this name)

OUT = os.path.join(REPO, 'tests', 'golden')
os.makedirs(OUT, exist_ok=True)


def make_targets(rng, b, k, t, seq_len, unlabeled_rows=(), weak_only_rows=()):
    """Targets shaped like pb_sed/data_preparation/transform.py:56-124 (values in {0,.5,1})."""
    weak = (rng.random((b, k)) < .3).astype(np.float32)
    for i in range(b):
        if weak[i].sum() == 0:
            weak[i, rng.integers(k)] = 1
    bnd = np.zeros((b, k, t), np.float32)
    for i in range(b):
        for c in range(k):
            if weak[i, c] and i not in weak_only_rows:
                on = rng.integers(0, max(seq_len[i] - 4, 1))
                off = min(on + rng.integers(2, 12), seq_len[i])
                bnd[i, c, on:off] = 1
            elif weak[i, c]:
                bnd[i, c, :seq_len[i]] = .5     # weakly labelled class: 0.5 over the clip
    for i in unlabeled_rows:
        weak[i] += (1 - weak[i]) * .5
        bnd[i] += (1 - bnd[i]) * .5
    return weak, bnd


def gen_fbcrnn_loss():
    rng = np.random.default_rng(11)
    cases = {}
    specs = [
        dict(name='ragged_strong', b=6, k=10, t=50, ragged=True, bwd=True, kw={}),
        dict(name='full_len', b=4, k=10, t=40, ragged=False, bwd=True, kw={}),
        dict(name='no_bwd', b=5, k=10, t=30, ragged=True, bwd=False, kw={}),
        dict(name='slat', b=5, k=10, t=30, ragged=True, bwd=True, kw=dict(slat=True)),
        dict(name='weak_only', b=5, k=10, t=30, ragged=True, bwd=True,
             kw=dict(strong_fwd_bwd_loss_weight=0.)),
        dict(name='half_weight_smooth', b=5, k=7, t=33, ragged=True, bwd=True,
             kw=dict(strong_fwd_bwd_loss_weight=.5, label_smoothing=.05)),
        dict(name='class_weights', b=4, k=6, t=20, ragged=True, bwd=True,
             kw=dict(class_weights=[1., 2., .5, 1., 3., 1.])),
    ]
    for s in specs:
        b, k, t = s['b'], s['k'], s['t']
        seq_len = np.sort(rng.integers(t // 2, t + 1, b))[::-1].copy() if s['ragged'] \
            else np.full(b, t)
        seq_len[0] = t
        weak, bnd = make_targets(rng, b, k, t, seq_len, unlabeled_rows=(b - 1,),
                                 weak_only_rows=(1,))
        yf = torch.tensor(rng.uniform(1e-5, 1 - 1e-5, (b, k, t)).astype(np.float32),
                          requires_grad=True)
        yb = torch.tensor(rng.uniform(1e-5, 1 - 1e-5, (b, k, t)).astype(np.float32),
                          requires_grad=True) if s['bwd'] else None
        m = RefFBCRNN(None, None, None, None, **s['kw'])
        m.train()
        inputs = {'seq_len': seq_len.tolist()}
        outputs = (yf, yb, seq_len, torch.zeros(b, 1, 4, t), seq_len,
                   (torch.tensor(weak), torch.tensor(bnd)))
        review = m.review(inputs, outputs)
        review['loss'].backward()
        n = s['name']
        cases.update({
            f'{n}/y_fwd': yf.detach().numpy(), f'{n}/seq_len': seq_len,
            f'{n}/weak_targets': weak, f'{n}/boundary_targets': bnd,
            f'{n}/loss': review['loss'].detach().numpy(),
            f'{n}/grad_y_fwd': yf.grad.numpy(),
            f'{n}/y_weak': review['buffers']['y_weak'],
            f'{n}/targets_weak': review['buffers']['targets_weak'],
            f'{n}/weak_label_rate': np.float64(review['scalars']['weak_label_rate']),
            f'{n}/boundary_label_rate': np.float64(review['scalars']['boundary_label_rate']),
            f'{n}/kw': np.array(repr(s['kw'])),
        })
        if yb is not None:
            cases.update({f'{n}/y_bwd': yb.detach().numpy(), f'{n}/grad_y_bwd': yb.grad.numpy()})
    np.savez_compressed(os.path.join(OUT, 'ref_fbcrnn_loss.npz'), **cases)
    print('ref_fbcrnn_loss', {k: float(v) for k, v in cases.items() if k.endswith('/loss')})


def gen_bicrnn_loss():
    rng = np.random.default_rng(12)
    cases = {}
    for n, (b, k, t, seg) in {'a': (4, 10, 50, 1), 'b': (3, 5, 21, 1), 'c': (5, 10, 50, 3), 'd': (4, 6, 37, 4)}.items():
        seq_len = np.sort(rng.integers(t // 2, t + 1, b))[::-1].copy()
        seq_len[0] = t
        weak, st = make_targets(rng, b, k, t, seq_len, unlabeled_rows=(b - 1,))
        y = torch.tensor(rng.uniform(.001, .999, (b, k, t)).astype(np.float32), requires_grad=True)
        m = RefBiCRNN(None, None, None, tag_conditioning=True, eval_segment_length=seg)
        review = m.review({'seq_len': seq_len.tolist()},
                          (y, seq_len, torch.zeros(b, 1, 4, t), seq_len,
                           (torch.tensor(weak), torch.tensor(st))))
        review['loss'].backward()
        cases.update({f'{n}/y': y.detach().numpy(), f'{n}/seq_len': seq_len,
                      f'{n}/strong_targets': st, f'{n}/loss': review['loss'].detach().numpy(),
                      f'{n}/grad_y': y.grad.numpy(),
                      f'{n}/y_strong': review['buffers']['y_strong'],
                      f'{n}/targets_strong': review['buffers']['targets_strong'],
                      f'{n}/eval_segment_length': np.int64(seg),
                      f'{n}/strong_label_rate': np.float64(review['scalars']['strong_label_rate'])})
    np.savez_compressed(os.path.join(OUT, 'ref_bicrnn_loss.npz'), **cases)
    print('ref_bicrnn_loss', {k: float(v) for k, v in cases.items() if k.endswith('/loss')})


def gen_fbcrnn_heads():
    rng = np.random.default_rng(13)
    b, c, t, k = 3, 6, 23, 4
    h = rng.standard_normal((b, c, t)).astype(np.float32)
    a_f = rng.standard_normal((k, c)).astype(np.float32)
    a_b = rng.standard_normal((k, c)).astype(np.float32)
    seq_len = np.array([23, 19, 12])
    cases = dict(h=h, a_fwd=a_f, a_bwd=a_b, seq_len=seq_len)
    for bwd in (True, False):
        m = RefFBCRNN(StubFeatures(), StubCNN(), StubRNN(a_f, False),
                      StubRNN(a_b, True) if bwd else None)
        m.eval()
        tag = 'fb' if bwd else 'f'
        inputs = {'stft': torch.tensor(h), 'seq_len': seq_len.tolist()}
        with torch.no_grad():
            y, sl = m.tagging(dict(inputs))
            cases[f'{tag}/tagging'], cases[f'{tag}/tagging_seq_len'] = y.numpy(), sl
            if bwd:
                y, sl = m.boundaries_detection(dict(inputs))
                cases[f'{tag}/boundaries'] = y.numpy()
            for wl, ws in ((5, 1), (4, 2), (1, 1), (7, 3)):
                y, sl = m.sound_event_detection(dict(inputs), wl, ws)
                cases[f'{tag}/sed_{wl}_{ws}'], cases[f'{tag}/sed_{wl}_{ws}_seq_len'] = y.numpy(), sl
            wl1 = [3, 5, 3, 9]
            y, sl = m.sound_event_detection(dict(inputs), wl1, 1)
            cases[f'{tag}/sed_1d'], cases['wl_1d'] = y.numpy(), np.array(wl1)
            wl2 = [[3, 5, 3, 9], [5, 5, 7, 3]]
            y, sl = m.sound_event_detection(dict(inputs), wl2, 2)
            cases[f'{tag}/sed_2d'], cases['wl_2d'] = y.numpy(), np.array(wl2)
    np.savez_compressed(os.path.join(OUT, 'ref_fbcrnn_heads.npz'), **cases)
    print('ref_fbcrnn_heads', sorted(cases))


def gen_filters():
    rng = np.random.default_rng(14)
    x = rng.random((3, 5, 61)).astype(np.float32)
    cases = dict(x=x)
    for n in (1, 3, 5, 11, 41, 101):
        cases[f'medfilt_{n}'] = ref_filters.medfilt(x.copy(), n, axis=-1)
    cases['medfilt_axis1_3'] = ref_filters.medfilt(x.copy(), 3, axis=1)
    for n in (2, 4, 10, 20):
        cases[f'stepfilt_{n}'] = ref_filters.stepfilt(x.copy(), n, axis=-1)
    for n in (0, 2, 6, 20):
        cases[f'boundariesfilt_{n}'] = ref_inf.boundariesfilt(x.copy(), n, axis=-1)
    l1 = np.array([1, 3, 5, 7, 11])
    cases['len_1d'] = l1
    cases['filtering_med_0d'] = ref_inf.filtering(x.copy(), ref_filters.medfilt, np.array(5))
    cases['filtering_med_1d'] = ref_inf.filtering(x.copy(), ref_filters.medfilt, l1)
    l2 = np.array([[1, 3, 5, 7, 11], [3, 3, 1, 21, 5]])
    cases['len_2d'] = l2
    cases['filtering_med_2d'] = ref_inf.filtering(x.copy(), ref_filters.medfilt, l2)
    l2b = np.array([[3], [9]])
    cases['len_2d_bcast'] = l2b
    cases['filtering_med_2d_bcast'] = ref_inf.filtering(x.copy(), ref_filters.medfilt, l2b)
    s1 = np.array([0, 2, 4, 10, 6])
    cases['steplen_1d'] = s1
    cases['filtering_bnd_1d'] = ref_inf.filtering(x.copy(), ref_inf.boundariesfilt, s1)
    # long rows / long filters: the reference tunes median filters up to 301 frames on 10 s clips
    # (pb_sed/experiments/strong_label_crnn/tuning.py:64)
    rng2 = np.random.default_rng(141)
    xl = rng2.random((2, 3, 500)).astype(np.float32)
    xl[0, 0, 100:300] = np.round(xl[0, 0, 100:300] * 4) / 4            # many ties
    xl[1, 2, :40] = 0.
    cases['x_long'] = xl
    for n in (5, 151, 301):
        cases[f'medfilt_long_{n}'] = ref_filters.medfilt(xl.copy(), n, axis=-1)
    np.savez_compressed(os.path.join(OUT, 'ref_filters.npz'), **cases)
    print('ref_filters', {k: (v.dtype.name, v.shape) for k, v in cases.items()})


class _FakeModel:
    def __init__(self, scores, seq_len):
        self.scores, self.seq_len = scores, seq_len

    def to(self, device):
        return self

    def eval(self):
        return self

    def example_to_device(self, ex, device=None):
        return ex

    def sound_event_detection(self, batch):
        i = batch['batch_idx']
        return torch.tensor(self.scores[i]), self.seq_len[i]

    boundaries_detection = sound_event_detection
    tagging = sound_event_detection


def gen_inference():
    rng = np.random.default_rng(15)
    n_models, n_batches, b, k, t = 3, 2, 4, 5, 47
    seq_len = [np.array([47, 40, 33, 20]), np.array([45, 44, 30, 9])]
    scores = rng.random((n_models, n_batches, b, k, t)).astype(np.float32)
    ids = [[f'clip{j}_{i}' for i in range(b)] for j in range(n_batches)]
    tags = {aid: (rng.random(k) < .5).astype(np.float64) for batch in ids for aid in batch}
    cases = dict(scores=scores, seq_len=np.stack(seq_len), ids=np.array(ids),
                 tags=np.stack([tags[a] for batch in ids for a in batch]))

    def dataset():
        return [{'batch_idx': j, 'example_id': ids[j], 'seq_len': seq_len[j].tolist(),
                 'weak_targets': 0} for j in range(n_batches)]
    models = [_FakeModel(scores[i], seq_len) for i in range(n_models)]

    def dump(name, out):
        if isinstance(out, dict):
            cases[name] = np.concatenate([out[a].reshape(-1) for batch in ids for a in batch])
            cases[name + '_dtype'] = np.array(out[ids[0][0]].dtype.name)
            cases[name + '_shape0'] = np.array(out[ids[0][0]].shape)

    dump('sed_med_scalar', ref_inf.sound_event_detection(models, dataset(), None, medfilt_length=5))
    ml = np.array([[1, 3, 5, 7, 11], [3, 3, 1, 21, 5]])
    am = np.array([[True, False, True, True, False], [False, False, True, True, True]])
    cases['medfilt_2d'], cases['apply_mask_2d'] = ml, am
    dump('sed_med_2d_masked', ref_inf.sound_event_detection(
        models, dataset(), None, medfilt_length=ml, apply_mask=am, masks=tags))
    dump('bnd_step', ref_inf.boundaries_detection(
        models, dataset(), None, stepfilt_length=np.array([0, 2, 4, 10, 6]),
        apply_mask=True, masks=tags))
    dump('tagging', ref_inf.tagging(models, dataset(), None))
    np.savez_compressed(os.path.join(OUT, 'ref_inference.npz'), **cases)
    print('ref_inference', {k: v.shape for k, v in cases.items()})


class _FakeSegModel(_FakeModel):
    """Scores are a fixed function of the (segment of the) input itself, so segmenting the input segments the scores."""

    def __init__(self, gain):
        self.gain = gain

    def sound_event_detection(self, batch):
        x = torch.as_tensor(batch['stft'])                     # [B, 1, T, K, 2]
        return (x[:, 0, :, :, 0] * self.gain + x[:, 0, :, :, 1]).transpose(1, 2), np.array(batch['seq_len'])


def gen_segments():
    """pb_sed/utils/segment.py (segment_batch, merge_segments) and the segmenting branch of
    pb_sed/models/base/inference.py:121-128,185-197."""
    import pb_sed.utils.segment as ref_seg
    rng = np.random.default_rng(18)
    b, k, t = 3, 4, 50
    seq_len = [50, 47, 41]
    stft = rng.random((b, 1, t, k, 2)).astype(np.float32)
    ids = ['a', 'b', 'c']
    cases = dict(stft=stft, seq_len=np.array(seq_len), ids=np.array(ids))
    for max_len, overlap in ((12, 2), (20, 5), (16, 0), (64, 4)):
        segs = ref_seg.segment_batch({'example_id': list(ids), 'stft': stft.copy(), 'seq_len': list(seq_len)}, max_len, overlap)
        tag = f'seg_{max_len}_{overlap}'
        cases[f'{tag}/n'] = np.int64(len(segs))
        for i, sgm in enumerate(segs):
            cases[f'{tag}/{i}/stft'] = np.asarray(sgm['stft'])
            cases[f'{tag}/{i}/seq_len'] = np.array(sgm['seq_len'])
            cases[f'{tag}/{i}/ids'] = np.array(sgm['example_id'])
        # merge of per-segment score arrays [T_seg, K] (and a 3-d variant [n, T_seg, K])
        if len(segs) > 1:
            out2 = {aid: np.asarray(sgm['stft'])[j, 0, :sl, :, 0] for sgm in segs
                    for j, (aid, sl) in enumerate(zip(sgm['example_id'], sgm['seq_len']))}
            merged = ref_seg.merge_segments(out2, overlap)
            for a in ids:
                cases[f'{tag}/merged/{a}'] = merged[a]
            out3 = {aid: np.stack([v, 2 * v]) for aid, v in out2.items()}
            merged3 = ref_seg.merge_segments(out3, overlap)
            for a in ids:
                cases[f'{tag}/merged3/{a}'] = merged3[a]
    # the driver with segmenting + merging (and a median filter applied per segment, as the reference does)
    models = [_FakeSegModel(1.), _FakeSegModel(.5)]
    for max_len, overlap, med in ((12, 2, 1), (20, 6, 3)):
        ds = [{'example_id': list(ids), 'stft': torch.tensor(stft), 'seq_len': list(seq_len), 'weak_targets': 0}]
        out = ref_inf.sound_event_detection(models, ds, None, medfilt_length=med, max_segment_length=max_len,
                                            segment_overlap=overlap, merge_score_segments=True)
        for a in ids:
            cases[f'driver_{max_len}_{overlap}_{med}/{a}'] = out[a]
    np.savez_compressed(os.path.join(OUT, 'ref_segments.npz'), **cases)
    print('ref_segments', len(cases), 'arrays')


class _ShimSTFT:
    """Stands in for padertorch's STFT inside Transform.__call__ (transform.py:53): a dummy 'stft' of the right length and
    the sample -> frame alignment rule of pb_sed_amd.data.samples_to_frames (third-party behaviour, restated)."""

    def __init__(self, shift=320, window_length=960):
        self.shift, self.window_length = shift, window_length

    def __call__(self, example):
        from pb_sed_amd.data import samples_to_frames
        from pb_sed_amd.modules import num_frames
        example = dict(example)
        t = num_frames(example['audio_data'].shape[-1], self.shift, self.window_length)
        example['stft'] = np.zeros((1, t, 3, 2), np.float32)
        example['events_start_frames'], example['events_stop_frames'] = samples_to_frames(
            example['events_start_samples'], example['events_stop_samples'], self.shift)
        return example


class _ShimLabelEncoder:
    """MultiHotAlignmentEncoder restated: encode(label) -> index; encode_alignment([(start, stop, idx)], seq_len) ->
    [seq_len, K] multi-hot; __call__(example) -> {'events': alignment of all events with their frames}."""
    label_key = 'events'

    def __init__(self, labels):
        self.label_mapping = {l: i for i, l in enumerate(labels)}

    def encode(self, label):
        return self.label_mapping[label]

    def encode_alignment(self, labels, seq_len):
        out = np.zeros((seq_len, len(self.label_mapping)), np.float32)
        for start, stop, idx in labels:
            out[max(start, 0):stop, idx] = 1
        return out

    def __call__(self, example):
        labels = [(a, o, self.encode(l)) for l, a, o in zip(example['events'], example['events_start_frames'], example['events_stop_frames'])]
        return {'events': self.encode_alignment(labels, example['stft'].shape[1])}


def gen_data_front_end():
    """pb_sed/data_preparation/mix.py SuperposeEvents (seeded np.random) and transform.py Transform.__call__ (under the
    STFT / label-encoder shims above) on float32 waveforms."""
    import pb_sed.data_preparation.mix as ref_mix
    import pb_sed.data_preparation.transform as ref_tr
    rng = np.random.default_rng(19)
    labels = ['alarm', 'dog', 'dishes', 'speech', 'water']
    cases = dict(labels=np.array(labels))

    def example(i, n, events, unlabeled=None, weak_only=False):
        ex = {'example_id': f'ex{i}', 'dataset': f'ds{i % 2}', 'audio_data': rng.standard_normal((1, n)).astype(np.float32)}
        if events is not None:
            ex['events'] = [e[0] for e in events]
            if not weak_only:
                ex['events_start_samples'] = [e[1] for e in events]
                ex['events_stop_samples'] = [e[2] for e in events]
                ex['label_types'] = [e[3] for e in events]
        if unlabeled is not None:
            ex['unlabeled'] = unlabeled
        return ex

    exs = [
        example(0, 16000, [('dog', 2000, 9000, 'strong'), ('dog', 12000, 15000, 'strong'), ('speech', 0, 16000, 'weak')]),
        example(1, 12345, [('alarm', 100, 6000, 'boundaries'), ('alarm', 7000, 12000, 'boundaries'), ('water', 3000, 4000, 'strong')]),
        example(2, 20000, [('dishes',), ('speech',)], weak_only=True),
        example(3, 9000, None),
        example(4, 16000, [('dog', 640, 3200, 'strong'), ('water', 0, 16000, 'weak')], unlabeled=True),
    ]
    for i, ex in enumerate(exs):
        cases[f'ex{i}/audio'] = ex['audio_data']
    # ---- mixing
    mixes = {'m01': ([0, 1], dict(min_overlap=.5, fade_length=0)), 'm203': ([2, 0, 3], dict(min_overlap=.25, fade_length=200)),
             'm41': ([4, 1], dict(min_overlap=1., max_length_in_samples=24000, fade_length=64))}
    for name, (idx, kw) in mixes.items():
        np.random.seed(hash(name) % 1000 if False else sum(map(ord, name)))
        out = ref_mix.SuperposeEvents(**kw)([{k: (v.copy() if isinstance(v, np.ndarray) else list(v) if isinstance(v, list) else v)
                                              for k, v in exs[i].items()} for i in idx])
        cases[f'{name}/idx'] = np.array(idx)
        cases[f'{name}/kw'] = np.array(repr(kw))
        cases[f'{name}/seed'] = np.int64(sum(map(ord, name)))
        cases[f'{name}/audio'] = out['audio_data']
        cases[f'{name}/events'] = np.array(out['events'])
        cases[f'{name}/start'] = np.array(out['events_start_samples'], dtype=np.int64)
        cases[f'{name}/stop'] = np.array(out['events_stop_samples'], dtype=np.int64)
        cases[f'{name}/label_types'] = np.array(out['label_types'])
        cases[f'{name}/unlabeled'] = np.bool_(out['unlabeled'])
        cases[f'{name}/example_id'] = np.array(out['example_id'])
    # ---- target encoding (incl. a mixture)
    enc = _ShimLabelEncoder(labels)
    tr = ref_tr.Transform.__new__(ref_tr.Transform)
    tr.stft, tr.label_encoder = _ShimSTFT(), enc
    tr.provide_boundary_targets = tr.provide_strong_targets = True
    tr.pop_audio_data, tr.anchor_sampling_fn, tr.anchor_shift_sampling_fn = True, None, None
    np.random.seed(7)
    mixed = ref_mix.SuperposeEvents(min_overlap=.5)([dict(exs[0]), dict(exs[1])])
    for name, ex in [(f'ex{i}', exs[i]) for i in range(5)] + [('mix', mixed)]:
        ex = {k: (list(v) if isinstance(v, list) else v) for k, v in ex.items()}
        out = tr(dict(ex))
        cases[f'targets/{name}/weak'] = out['weak_targets']
        cases[f'targets/{name}/boundary'] = out['boundary_targets']
        cases[f'targets/{name}/strong'] = out['strong_targets']
        cases[f'targets/{name}/seq_len'] = np.int64(out['seq_len'])
    cases['targets/mix/start'] = np.array(mixed['events_start_samples'], dtype=np.int64)
    cases['targets/mix/stop'] = np.array(mixed['events_stop_samples'], dtype=np.int64)
    cases['targets/mix/events'] = np.array(mixed['events'])
    cases['targets/mix/label_types'] = np.array(mixed['label_types'])
    cases['targets/mix/n'] = np.int64(mixed['audio_data'].shape[1])
    for i, ex in enumerate(exs):
        e = ref_mix.add_label_types({k: (list(v) if isinstance(v, list) else v) for k, v in ex.items()})
        cases[f'ex{i}/events'] = np.array(e['events'], dtype='<U16')
        cases[f'ex{i}/start'] = np.array(e['events_start_samples'], dtype=np.int64)
        cases[f'ex{i}/stop'] = np.array(e['events_stop_samples'], dtype=np.int64)
        cases[f'ex{i}/label_types'] = np.array(e['label_types'], dtype='<U16')
        cases[f'ex{i}/unlabeled'] = np.bool_(e['unlabeled'])
    np.savez_compressed(os.path.join(OUT, 'ref_data_front_end.npz'), **cases)
    print('ref_data_front_end', len(cases), 'arrays')


def gen_instance_based():
    """Validation metrics (pb_sed/evaluation/instance_based.py, pure numpy): threshold searches on score matrices with
    ties, the rate constraints, binary-decision metrics and lwlrap."""
    import pb_sed.evaluation.instance_based as ref_ib
    rng = np.random.default_rng(16)
    n, k = 240, 6
    targets = (rng.random((n, k)) < np.array([.5, .3, .1, .7, .02, .4])).astype(np.float64)
    scores = np.clip(targets * .35 + rng.random((n, k)) * .8, 0, 1)
    scores_q = np.round(scores * 12) / 12                       # many ties
    cases = dict(targets=targets, scores=scores, scores_q=scores_q)

    def put(name, out):
        for i, o in enumerate(out):
            cases[f'{name}_{i}'] = np.asarray(o, dtype=np.float64)

    for tag, sc in (('c', scores), ('q', scores_q)):
        put(f'best_f_2d_{tag}', ref_ib.get_best_fscore_thresholds(targets, sc))
        put(f'best_f_2d_minp_{tag}', ref_ib.get_best_fscore_thresholds(targets, sc, min_precision=.6))
        put(f'best_f_2d_minr_{tag}', ref_ib.get_best_fscore_thresholds(targets, sc, min_recall=.8))
        put(f'best_f_2d_beta2_{tag}', ref_ib.get_best_fscore_thresholds(targets, sc, beta=2.))
        put(f'best_er_2d_{tag}', ref_ib.get_best_er_thresholds(targets, sc))
        put(f'best_er_2d_maxi_{tag}', ref_ib.get_best_er_thresholds(targets, sc, max_insertion_rate=.1))
        put(f'best_er_2d_maxd_{tag}', ref_ib.get_best_er_thresholds(targets, sc, max_deletion_rate=.2))
        put(f'curve_f_2d_{tag}', ref_ib.fscore_curve(targets, sc))
        put(f'curve_er_2d_{tag}', ref_ib.er_curve(targets, sc))
        for c in (0, 2, 4):
            put(f'best_f_1d_{tag}{c}', ref_ib.get_best_fscore_thresholds(targets[:, c], sc[:, c]))
            put(f'best_f_1d_beta2_bias_{tag}{c}', ref_ib.get_best_fscore_thresholds(
                targets[:, c], sc[:, c], beta=2., tp_bias=1, n_ref_bias=2, n_pos_bias=3))
            put(f'best_er_1d_{tag}{c}', ref_ib.get_best_er_thresholds(targets[:, c], sc[:, c]))
    put('lwlrap', ref_ib.lwlrap(targets, scores))
    decisions = (scores[None] > np.array([.3, .5, .7])[:, None, None]).astype(np.float64)
    cases['decisions'] = decisions
    for ew in (False, True):
        put(f'fscore_ew{int(ew)}', ref_ib.fscore(targets, decisions, event_wise=ew))
        put(f'fscore_beta2_ew{int(ew)}', ref_ib.fscore(targets, decisions, beta=2., event_wise=ew))
        put(f'error_rate_ew{int(ew)}', ref_ib.error_rate(targets, decisions[1], event_wise=ew))
    # the module's own known-answer example (docstrings of fscore_curve / get_best_*_thresholds)
    t9 = np.array([1.0, 1.0, 0.0, 1.0, 0.0, 0.0, 0.0, 0.0, 0.0])
    s9 = np.array([0.6, 0.2, 0.5, 0.4, 0.3, 0.1, 0.7, 0.0, 0.0])
    cases['t9'], cases['s9'] = t9, s9
    put('t9_curve_f', ref_ib.fscore_curve(t9, s9))
    put('t9_best_f', ref_ib.get_best_fscore_thresholds(t9, s9))
    put('t9_best_f_minp', ref_ib.get_best_fscore_thresholds(t9, s9, min_precision=.51))
    put('t9_best_er', ref_ib.get_best_er_thresholds(t9, s9))
    np.savez_compressed(os.path.join(OUT, 'ref_instance_based.npz'), **cases)
    print('ref_instance_based', len(cases), 'arrays')


def gen_summary_metrics():
    """SoundEventModel.add_metrics_to_summary (pb_sed/models/base/model.py:44-88) called as a plain function on a
    stand-in object: summary scalars for label subsets / label-wise metrics / the mAP-mAUC branch."""
    import pb_sed.models.base.model as ref_model
    fn = ref_model.SoundEventModel.__dict__['add_metrics_to_summary']
    rng = np.random.default_rng(17)
    n, k = 90, 5
    targets = (rng.random((n, k)) < np.array([.5, .3, .2, .6, .4])).astype(np.float64)
    scores = np.clip(targets * .3 + rng.random((n, k)) * .8, 0, 1)
    rare = targets.copy(); rare[:, 2] = 0; rare[0, 2] = 1          # a class with a single positive: no mAP / mAUC
    labels = ['alarm', 'dog', 'dishes', 'speech', 'water']
    cases = dict(targets=targets, scores=scores, targets_rare=rare, labels=np.array(labels))
    configs = {
        'plain': dict(labelwise_metrics=(), label_mapping=None, test_labels=None),
        'labelwise': dict(labelwise_metrics=('fscore_weak', 'lwlrap_weak', 'ap_weak'), label_mapping=labels, test_labels=None),
        'subset_idx': dict(labelwise_metrics=('error_rate_weak',), label_mapping=None, test_labels=[0, 3, 4]),
        'subset_names': dict(labelwise_metrics=('fscore_weak', 'auc_weak'), label_mapping=labels, test_labels=['dog', 'water']),
    }
    for name, cfg in configs.items():
        for tname, t in (('all', targets), ('rare', rare)):
            me = types.SimpleNamespace(**cfg)
            summary = dict(scalars={}, buffers={'y_weak': [scores[:40], scores[40:]], 'targets_weak': [t[:40], t[40:]]})
            fn(me, summary, 'weak')
            keys = sorted(summary['scalars'])
            cases[f'{name}/{tname}/keys'] = np.array(keys)
            cases[f'{name}/{tname}/values'] = np.array([float(summary['scalars'][q]) for q in keys])
    np.savez_compressed(os.path.join(OUT, 'ref_summary_metrics.npz'), **cases)
    print('ref_summary_metrics', len(cases), 'arrays')


def gen_pseudo_label():
    """pb_sed/models/base/pseudo_label.py (pseudo_label, set_onset_offset_times): the hand-off from the ensemble inference to the
    next training round.  Pure dictionary work: inputs and the reference's outputs go into a JSON fixture."""
    import json
    import importlib
    ref_pl = importlib.import_module('pb_sed.models.base.pseudo_label')      # (the package re-exports the FUNCTION under the module's name)
    rng = np.random.default_rng(23)
    classes = ['Alarm_bell_ringing', 'Blender', 'Cat', 'Dishes', 'Dog', 'Speech']
    dataset, tags, events, boundaries = {}, {}, {}, {}
    for i in range(14):
        aid = f'clip{i:02d}'
        length = float(np.round(rng.uniform(4, 10), 3))
        present = [c for c in classes if rng.random() < .35]
        if i == 3:
            present = []                                             # an untagged clip
        dataset[aid] = {'audio_path': f'/data/{aid}.wav', 'audio_length': length, 'events': list(present)}
        tags[aid] = [float(np.round(rng.random() * (.9 if c in present else .6) + (.3 if c in present else 0.), 4)) for c in classes]
        det = []
        for c in classes:
            for _ in range(int(rng.integers(0, 3))):
                on = float(np.round(rng.uniform(0, length - .5), 2))
                det.append((on, float(np.round(min(on + rng.uniform(.2, 3.), length), 2)), c))
        if i == 5:
            det = []                                                 # tagged, nothing detected: weak labels over the whole clip
        events[aid] = det
        boundaries[aid] = [(on, off, c) for on, off, c in det if rng.random() < .6]
    cases = {}
    for name, (pt, pb, pe) in {'none': (False, False, False), 'tags': (True, False, False), 'events': (False, False, True),
                               'boundaries': (False, True, False), 'tags_events': (True, False, True), 'tags_boundaries': (True, True, False)}.items():
        out = ref_pl.pseudo_label(dataset, classes, pt, pb, pe, tags, boundaries, events)
        cases[name] = {'flags': [pt, pb, pe], 'same_object': out is dataset, 'dataset': out}
    fixture = {'event_classes': classes, 'dataset': dataset, 'tags': tags, 'events': events, 'boundaries': boundaries, 'cases': cases}
    with open(os.path.join(OUT, 'ref_pseudo_label.json'), 'w') as f:
        json.dump(fixture, f, indent=0, sort_keys=True)
    print('ref_pseudo_label', len(cases), 'cases')


def tuning_inputs():
    """Synthetic score DataFrames / tags / targets for the tuning drivers: 9 clips (two of them shorter), 3 classes, float32
    scores with class activity where the target says so.  Shared with tests/test_host_logic.py."""
    import pandas as pd
    rng = np.random.default_rng(31)
    classes = ['Blender', 'Dog', 'Speech']
    scores, tags, targets = {}, {}, {}
    for i in range(9):
        t = 33 if i in (2, 6) else 40
        tgt = (rng.random(3) < .5).astype(np.float64)
        x = rng.random((t, 3)) * .45
        for k in range(3):
            if tgt[k]:
                on = int(rng.integers(0, t - 12))
                x[on:on + int(rng.integers(6, 12)), k] += .5
            for _ in range(2):                                          # isolated spikes: what a median filter removes
                x[int(rng.integers(0, t)), k] = rng.random()
        x = np.clip(x, 0, 1).astype(np.float32)
        ts = np.round(np.arange(t + 1) * .02, 6)
        aid = f'clip{i}'
        scores[aid] = pd.DataFrame(np.concatenate((ts[:-1, None], ts[1:, None], x.astype(np.float64)), axis=1), columns=['onset', 'offset', *classes])
        noisy = tgt.copy()
        if i in (1, 5):
            noisy[i % 3] = 1 - noisy[i % 3]                             # tags that disagree with the targets
        tags[aid], targets[aid] = noisy, tgt
    return classes, scores, tags, targets


def _board_arrays(prefix, leaderboard, classes, cases):
    for metric_name, (values, params, scores) in leaderboard.items():
        cases[f'{prefix}/{metric_name}/values'] = np.array([values[c] for c in classes + ['macro_average']])
        cases[f'{prefix}/{metric_name}/params'] = np.array(repr({c: dict(sorted(params[c].items())) for c in classes}))
        for aid in sorted(scores):
            cases[f'{prefix}/{metric_name}/scores/{aid}'] = scores[aid][classes].to_numpy()


def gen_tuning():
    """pb_sed/models/base/tuning.py: update_leaderboard, tune_tagging, tune_boundaries_detection, tune_sound_event_detection and
    boundaries_from_events, run on synthetic score DataFrames with two deterministic metric functions (tests/stubs.py)."""
    import importlib
    from tests.stubs import make_tuning_metrics
    ref_tu = importlib.import_module('pb_sed.models.base.tuning')
    classes, scores, tags, targets = tuning_inputs()
    metrics = make_tuning_metrics(targets, classes)
    cases = {'classes': np.array(classes)}
    for aid in sorted(scores):
        cases[f'inputs/scores/{aid}'] = scores[aid].to_numpy()
        cases[f'inputs/tags/{aid}'], cases[f'inputs/targets/{aid}'] = tags[aid], targets[aid]
    _board_arrays('tagging', ref_tu.tune_tagging(scores, [1, 3, 7], metrics, minimize=['leak']), classes, cases)
    _board_arrays('boundaries', ref_tu.tune_boundaries_detection(scores, [1, 5], [0, 4, 10], tags, metrics, minimize={'hit_rate': False, 'leak': True},
                                                                 tag_masking='?'), classes, cases)
    _board_arrays('sed', ref_tu.tune_sound_event_detection(scores, [1, 5, 11], tags, metrics, minimize=['leak'],
                                                           tag_masking={'hit_rate': True, 'leak': '?'}), classes, cases)
    gt = {'a': [(0.5, 1.0, 'Dog'), (2.0, 2.5, 'Dog'), (0.1, 4.0, 'Speech'), (3.0, 3.5, 'Dog')], 'b': [], 'c': [(1.0, 2.0, 'Blender')]}
    cases['boundaries_from_events'] = np.array(repr(ref_tu.boundaries_from_events(gt)))
    np.savez_compressed(os.path.join(OUT, 'ref_tuning.npz'), **cases)
    print('ref_tuning', len(cases), 'arrays')


if __name__ == '__main__':
    gen_tuning()
    gen_pseudo_label()
    gen_fbcrnn_loss()
    gen_bicrnn_loss()
    gen_fbcrnn_heads()
    gen_filters()
    gen_inference()
    gen_segments()
    gen_data_front_end()
    gen_instance_based()
    gen_summary_metrics()
    assert not os.path.exists('/root/reference/pb_sed/__pycache__'), 'bytecode written to reference'
