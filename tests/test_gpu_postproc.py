"""GPU post-processing kernels vs the reference's own golden vectors (bit-exact) and the oracle."""
import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu
DEV = 'cuda:0'


def dev(a):
    return torch.as_tensor(np.ascontiguousarray(a)).to(DEV)


def test_medfilt_bit_exact_vs_reference(golden):
    from pb_sed_amd import ops
    g = golden('ref_filters.npz')
    x = dev(g['x'])
    for n in (1, 3, 5, 11, 41, 101):
        np.testing.assert_array_equal(ops.medfilt(x, n).cpu().numpy(), g[f'medfilt_{n}'])
    np.testing.assert_array_equal(ops.medfilt(x, g['len_1d']).cpu().numpy(), g['filtering_med_1d'])


def test_boundariesfilt_vs_reference(golden):
    from pb_sed_amd import ops
    g = golden('ref_filters.npz')
    x = dev(g['x'])
    np.testing.assert_array_equal(ops.boundariesfilt(x, 0).cpu().numpy(), g['boundariesfilt_0'])
    for n in (2, 6, 20):
        out = ops.boundariesfilt(x, n, want_f64=True).cpu().numpy()
        np.testing.assert_allclose(out, g[f'boundariesfilt_{n}'], rtol=0, atol=1e-15)
    np.testing.assert_array_equal(ops.boundariesfilt(x, g['steplen_1d']).cpu().numpy(), g['filtering_bnd_1d'])


def test_filtering_variants_bit_exact(golden):
    from pb_sed_amd import inference as inf, ops
    g = golden('ref_filters.npz')
    x = dev(g['x'])
    np.testing.assert_array_equal(inf.filtering(x, ops.medfilt, np.array(5)).cpu().numpy(), g['filtering_med_0d'])
    np.testing.assert_array_equal(inf.filtering(x, ops.medfilt, g['len_2d']).cpu().numpy(), g['filtering_med_2d'])
    np.testing.assert_array_equal(inf.filtering(x, ops.medfilt, g['len_2d_bcast']).cpu().numpy(),
                                  g['filtering_med_2d_bcast'])


def test_ensemble_postprocess_bit_exact_vs_reference_inference(golden):
    from pb_sed_amd import inference as inf
    g = golden('ref_inference.npz')
    scores, seq_len, ids = g['scores'], g['seq_len'], g['ids']
    flat_ids = [a for batch in ids for a in batch]
    tags = dict(zip(flat_ids, g['tags']))

    def run(**kw):
        out = {}
        for j in range(scores.shape[1]):
            out.update(inf.postprocess_batch([dev(scores[i, j]) for i in range(scores.shape[0])], seq_len[j],
                                             list(ids[j]), **kw))
        return out
    flat = lambda o: np.concatenate([o[a].reshape(-1) for a in flat_ids])
    o = run(medfilt_length=5)
    assert o[flat_ids[0]].dtype.name == str(g['sed_med_scalar_dtype'])
    np.testing.assert_array_equal(flat(o), g['sed_med_scalar'])
    o = run(medfilt_length=g['medfilt_2d'], apply_mask=g['apply_mask_2d'], masks=tags)
    assert tuple(o[flat_ids[0]].shape) == tuple(g['sed_med_2d_masked_shape0'])
    np.testing.assert_array_equal(flat(o), g['sed_med_2d_masked'])
    o = run(stepfilt_length=np.array([0, 2, 4, 10, 6]), apply_mask=True, masks=tags)
    assert o[flat_ids[0]].dtype.name == str(g['bnd_step_dtype'])
    np.testing.assert_array_equal(flat(o), g['bnd_step'])
    o = run(post_processing_fn=lambda x: x.max(-2, keepdims=True))
    np.testing.assert_array_equal(flat(o), g['tagging'])


def test_event_frames_bit_exact_vs_oracle():
    from oracle import postproc as pp
    from pb_sed_amd import inference as inf
    rng = np.random.default_rng(5)
    classes = [f'c{i}' for i in range(6)]
    scores = {f'clip{i}': rng.random((int(t), 6)).astype(np.float32) for i, t in enumerate((50, 37, 1, 64))}
    scores['clip4'] = np.ones((20, 6), np.float32)          # event running to the last frame
    scores['clip5'] = np.zeros((20, 6), np.float32)         # no events
    thr = np.array([.5, .3, .9, .1, .7, .5], np.float32)
    ts = np.round(np.arange(0, 1000) * .02, 6)
    got = inf.scores_to_event_list(scores, thr, classes, ts, device=DEV)
    for a, s in scores.items():
        assert got[a] == pp.scores_to_event_list(s, ts, thr, classes), a


def test_inference_driver_end_to_end_vs_oracle():
    """Config-5 shape in miniature: 2 FBCRNN taggers -> tags -> 2 tag-conditioned BiCRNN detectors."""
    from oracle import frontend as ofe, models as om, postproc as pp
    from pb_sed_amd import inference as inf
    from pb_sed_amd.models import strong_label, weak_label
    from tests.test_gpu_model import TINY, synth_batch
    torch.manual_seed(3)
    kw = dict(num_events=10, number_of_filters=128, hidden_size=64, num_layers=2, net=TINY)
    pairs_w, pairs_s = [], []
    for _ in range(2):
        r = om.FBCRNN.build(**kw).eval()
        m = weak_label.CRNN.build(**kw)
        m.load_state_dict(r.state_dict())
        pairs_w.append((r, m))
        r = om.BiCRNN.build(tag_conditioning=True, **kw).eval()
        m = strong_label.CRNN.build(tag_conditioning=True, **kw)
        m.load_state_dict(r.state_dict())
        pairs_s.append((r, m))
    wav, seq, *_ = synth_batch(4, 16000, 10, seed=7)
    ids = [f'a{i}' for i in range(4)]
    batch = {'audio_data': wav, 'seq_len': seq.tolist(), 'example_id': ids}
    stft = ofe.stft(wav)
    # taggers
    tag_scores = inf.tagging([m for _, m in pairs_w], [dict(batch)], DEV)
    with torch.no_grad():
        ref = pp.postprocess([r.tagging({'stft': stft, 'seq_len': seq.tolist()})[0].numpy() for r, _ in pairs_w],
                             np.ones(4, int), ids, tagging=True)
    for a in ids:
        np.testing.assert_allclose(tag_scores[a], ref[a], atol=1e-4)
    tags = {a: (ref[a][0] > .5).astype(np.float32) for a in ids}
    tag_cond = torch.tensor(np.stack([tags[a] for a in ids]))
    # detectors, per-class median filters, two variants, masked by tags
    ml = np.array([[1, 3, 5, 7, 9, 1, 3, 5, 7, 9], [3] * 10])
    sed = inf.sound_event_detection([m for _, m in pairs_s], [dict(batch, tag_condition=tag_cond)], DEV,
                                    medfilt_length=ml, apply_mask=True, masks=tags)
    with torch.no_grad():
        ref = pp.postprocess([r.sound_event_detection({'stft': stft, 'seq_len': seq.tolist(),
                                                       'tag_condition': tag_cond})[0].numpy() for r, _ in pairs_s],
                             seq, ids, medfilt_length=ml, apply_mask=True, masks=tags)
    for a in ids:
        assert sed[a].shape == ref[a].shape
        np.testing.assert_allclose(sed[a], ref[a], atol=1e-4)


def test_inference_driver_keeps_a_batch_in_flight_without_changing_results():
    """inference() queues batch n + 1 before it waits for batch n (pinned buffer + event per batch, reused buffers).  Five
    batches of different content and (for one) different shape through one call must give, bit for bit, what five
    one-batch calls give; a generator works as dataset and is pulled one batch ahead at most."""
    from oracle import models as om
    from pb_sed_amd import inference as inf
    from pb_sed_amd.models import strong_label
    from tests.test_gpu_model import TINY, synth_batch
    torch.manual_seed(5)
    kw = dict(num_events=10, number_of_filters=128, hidden_size=64, num_layers=2, net=TINY)
    models = []
    for _ in range(2):
        m = strong_label.CRNN.build(tag_conditioning=True, **kw)
        m.load_state_dict(om.BiCRNN.build(tag_conditioning=True, **kw).state_dict())
        models.append(m)
    batches, masks = [], {}
    for j in range(5):
        n = 3 if j == 2 else 4
        wav, seq, *_ = synth_batch(n, 16000, 10, ragged=True, seed=20 + j)
        ids = [f'b{j}_{i}' for i in range(n)]
        cond = (torch.rand(n, 10, generator=torch.Generator().manual_seed(j)) > .5).float()
        masks.update({a: cond[i].numpy() for i, a in enumerate(ids)})
        batches.append({'audio_data': wav, 'seq_len': seq.tolist(), 'example_id': ids, 'tag_condition': cond})
    ml = np.array([[1, 3, 5, 7, 9, 1, 3, 5, 7, 9], [3] * 10])
    kwargs = dict(medfilt_length=ml, apply_mask=True, masks=masks)
    one_by_one = {}
    for b in batches:
        one_by_one.update(inf.sound_event_detection(models, [dict(b)], DEV, **kwargs))
    pulled = []

    def dataset():
        for j, b in enumerate(batches):
            pulled.append(j)
            yield dict(b)
    together = inf.sound_event_detection(models, dataset(), DEV, **kwargs)
    # the same batches handed over from pinned host memory: the copies run on a side stream, one batch ahead
    pinned = [dict(b, audio_data=b['audio_data'].pin_memory(), tag_condition=b['tag_condition'].pin_memory()) for b in batches]
    from_host = inf.sound_event_detection(models, pinned, DEV, **kwargs)
    for a in one_by_one:
        assert np.array_equal(from_host[a], one_by_one[a]), a
    assert pulled == list(range(5)) and sorted(together) == sorted(one_by_one)
    for a in one_by_one:
        assert together[a].shape == one_by_one[a].shape
        assert np.array_equal(together[a], one_by_one[a]), a
    from pb_sed_amd import ops
    assert not any(slot[2] for pool in ops._PINNED.values() for slot in pool), 'a pinned buffer is still held'
    big = {k: len(v) for k, v in ops._PINNED.items() if k >= 4096}          # pool keys = byte size classes (powers of two)
    assert big and max(big.values()) <= 3, big                    # score buffers: the pipeline depth, not one per batch
    assert len(ops._PINNED) <= 8, sorted(ops._PINNED)             # ... and the two score shapes share size classes, not one pool per shape


@pytest.mark.parametrize('hidden,batch', [(64, 5), (256, 40)])
def test_detectors_sharing_scan_launches_give_the_scores_of_separate_runs(hidden, batch):
    """strong_label.CRNN.sound_event_detection_jointly: the ensemble's networks run their GRU layers in shared persistent-scan
    launches (up to three networks = six chains per launch; four networks here = a group of three and a single one; the
    larger case switches the launch to two batch tiles per block).  Per network the arithmetic is that of its own run: the
    scores must be identical, and inference() must take that path."""
    from oracle import models as om
    from pb_sed_amd import inference as inf
    from pb_sed_amd.models import strong_label
    from tests.test_gpu_model import TINY, synth_batch
    torch.manual_seed(11)
    kw = dict(num_events=10, number_of_filters=128, hidden_size=hidden, num_layers=2, net=TINY)
    models = []
    for _ in range(4):
        m = strong_label.CRNN.build(tag_conditioning=True, **kw)
        m.load_state_dict(om.BiCRNN.build(tag_conditioning=True, **kw).state_dict())
        models.append(m.to(DEV).eval())
    wav, seq, *_ = synth_batch(batch, 16000, 10, ragged=True, seed=3)
    cond = (torch.rand(batch, 10, generator=torch.Generator().manual_seed(1)) > .5).float()
    example = {'audio_data': wav.to(DEV), 'seq_len': seq.tolist(), 'example_id': [f'e{i}' for i in range(batch)],
               'tag_condition': cond.to(DEV)}
    assert strong_label.CRNN.can_run_jointly(models)
    with torch.no_grad():
        separate = [m.sound_event_detection(dict(example)) for m in models]
    joint = strong_label.CRNN.sound_event_detection_jointly(models, dict(example))
    assert len(joint) == 4
    for (ys, ls), (yj, lj) in zip(separate, joint):
        assert (np.asarray(ls) == np.asarray(lj)).all()
        assert torch.equal(ys, yj), (ys - yj).abs().max().item()
    calls = []
    orig = strong_label.CRNN.sound_event_detection_jointly.__func__
    try:
        strong_label.CRNN.sound_event_detection_jointly = classmethod(lambda cls, ms, seg: (calls.append(len(ms)), orig(cls, ms, seg))[1])
        out = inf.sound_event_detection(models, [dict(example)], DEV)
    finally:
        strong_label.CRNN.sound_event_detection_jointly = classmethod(orig)
    assert calls == [4] and len(out) == batch
    models[1].train()                                   # a network in training mode (batch statistics): one model after the other
    assert not strong_label.CRNN.can_run_jointly(models)


def test_medfilt_long_filters_bit_exact(golden):
    """Median filters up to 301 frames on 500-frame rows (the reference's tuning range,
    pb_sed/experiments/strong_label_crnn/tuning.py:64) incl. rows with ties and zero runs: the bisection-select kernel
    (O(32 n) per output) against the reference's scipy medfilt, bit for bit."""
    from pb_sed_amd import ops
    g = golden('ref_filters.npz')
    x = dev(g['x_long'])
    for n in (5, 151, 301):
        np.testing.assert_array_equal(ops.medfilt(x, n).cpu().numpy(), g[f'medfilt_long_{n}'])
    neg = dev(-g['x_long'] + .25)                       # negative values and sign changes: compare with scipy directly
    from scipy import signal
    want = np.stack([signal.medfilt(r, 31) for r in neg.cpu().numpy().reshape(-1, 500)]).reshape(neg.shape)
    np.testing.assert_array_equal(ops.medfilt(neg, 31).cpu().numpy(), want)


class _StftScoreModel:
    """Fake model of tests/golden/gen_golden.py::_FakeSegModel: scores are a fixed function of the input segment."""

    def __init__(self, gain):
        self.gain = gain

    def to(self, device):
        return self

    def eval(self):
        return self

    def example_to_device(self, ex, device=None):
        return {k: (v.to(device) if isinstance(v, torch.Tensor) else v) for k, v in ex.items()}

    def sound_event_detection(self, batch):
        x = batch['stft']
        return (x[:, 0, :, :, 0] * self.gain + x[:, 0, :, :, 1]).transpose(1, 2), np.array(batch['seq_len'])


def test_inference_driver_segments_and_merges_like_the_reference(golden):
    """pb_sed/models/base/inference.py:121-128,185-197 + pb_sed/utils/segment.py: the reference's own driver was run with
    max_segment_length / segment_overlap / merge_score_segments on a fake model; same inputs through the build's driver."""
    from pb_sed_amd import inference as inf
    g = golden('ref_segments.npz')
    ids, seq = g['ids'].tolist(), g['seq_len'].tolist()
    models = [_StftScoreModel(1.), _StftScoreModel(.5)]
    for max_len, overlap, med in ((12, 2, 1), (20, 6, 3)):
        ds = [{'example_id': list(ids), 'stft': torch.tensor(g['stft']), 'seq_len': list(seq), 'weak_targets': 0}]
        out = inf.sound_event_detection(models, ds, DEV, medfilt_length=med, max_segment_length=max_len,
                                        segment_overlap=overlap, merge_score_segments=True)
        assert sorted(out) == sorted(ids)
        for a in ids:
            np.testing.assert_array_equal(out[a], g[f'driver_{max_len}_{overlap}_{med}/{a}'])
    # without merging the per-segment entries are returned under the reference's segment ids
    ds = [{'example_id': list(ids), 'stft': torch.tensor(g['stft']), 'seq_len': list(seq)}]
    out = inf.sound_event_detection(models, ds, DEV, max_segment_length=12, segment_overlap=2)
    assert 'a_!segment!_0_5' in out and len(out) == 15


def test_audio_segments_give_the_features_of_the_whole_clip():
    """Segmenting the waveform (fused front-end input) must be indistinguishable from segmenting the reference's STFT:
    features of every audio segment == the matching frame range of the whole clip's features, bit for bit."""
    from pb_sed_amd.models import weak_label
    from pb_sed_amd.utils.segment import segment_batch
    from tests.test_gpu_model import TINY
    torch.manual_seed(0)
    model = weak_label.CRNN.build(num_events=10, hidden_size=64, net=TINY).to(DEV).eval()
    n = 16000 * 5 + 321
    wav = torch.randn(3, n, device=DEV)
    with torch.no_grad():
        from pb_sed_amd.modules import num_frames
        t = num_frames(n)
        seq = [t, t - 11, t - 51]
        full = model({'audio_data': wav, 'seq_len': seq})[3]
        segs = segment_batch({'audio_data': wav, 'seq_len': seq, 'example_id': ['a', 'b', 'c']}, 64, 6)
        assert len(segs) >= 4
        for s in segs:
            x = model(dict(s))[3]
            a = s['segment_start']
            for j, sl in enumerate(s['seq_len']):
                sl = max(sl, 0)
                assert torch.equal(x[j, :, :, :sl], full[j, :, :, a:a + sl]), (a, j)


def _golden_examples(g, device=None):
    exs = []
    for i in range(5):
        audio = torch.as_tensor(g[f'ex{i}/audio'])
        ex = {'example_id': f'ex{i}', 'dataset': f'ds{i % 2}', 'audio_data': audio if device is None else audio.to(device),
              'events': [str(v) for v in g[f'ex{i}/events']], 'events_start_samples': g[f'ex{i}/start'].tolist(),
              'events_stop_samples': g[f'ex{i}/stop'].tolist(), 'label_types': [str(v) for v in g[f'ex{i}/label_types']],
              'unlabeled': bool(g[f'ex{i}/unlabeled'])}
        exs.append(ex)
    return exs


def test_superpose_events_matches_the_reference(golden):
    """pb_sed/data_preparation/mix.py::SuperposeEvents executed by the reference on seeded float32 clips: the build draws
    the offsets with the same np.random calls, shifts the event boundaries the same way and mixes on the device in one
    launch - audio bit-exact without fades, within one float32 ulp where the float64 raised-cosine fade is applied."""
    from pb_sed_amd.data import SuperposeEvents
    g = golden('ref_data_front_end.npz')
    exs = _golden_examples(g, DEV)
    for name in ('m01', 'm203', 'm41'):
        kw = eval(str(g[f'{name}/kw']))
        np.random.seed(int(g[f'{name}/seed']))
        out = SuperposeEvents(**kw)([dict(exs[i]) for i in g[f'{name}/idx']])
        assert out['example_id'] == str(g[f'{name}/example_id']) and out['unlabeled'] == bool(g[f'{name}/unlabeled'])
        assert out['events'] == [str(v) for v in g[f'{name}/events']]
        assert out['events_start_samples'] == g[f'{name}/start'].tolist() and out['events_stop_samples'] == g[f'{name}/stop'].tolist()
        assert out['label_types'] == [str(v) for v in g[f'{name}/label_types']]
        got, want = out['audio_data'].cpu().numpy(), g[f'{name}/audio']
        assert got.shape == want.shape and out['seq_len'] == want.shape[1]
        if kw.get('fade_length', 0) == 0:
            np.testing.assert_array_equal(got, want)
        else:
            np.testing.assert_allclose(got, want, rtol=2.5e-7, atol=1e-9)
            assert (got == want).mean() > .999


def test_target_encoding_bit_exact_vs_the_reference(golden):
    """pb_sed/data_preparation/transform.py:56-124 executed by the reference (under an STFT / label-encoder shim):
    strongly, boundary-, weakly and un-labelled clips, an unlabeled clip with events, and a mixture - weak, boundary and
    strong targets of a ragged batch in one launch, bit for bit."""
    from pb_sed_amd.data import encode_targets
    g = golden('ref_data_front_end.npz')
    labels = {str(l): i for i, l in enumerate(g['labels'])}
    exs = _golden_examples(g)
    exs.append({'audio_data': torch.zeros(1, int(g['targets/mix/n'])), 'events': [str(v) for v in g['targets/mix/events']],
                'events_start_samples': g['targets/mix/start'].tolist(), 'events_stop_samples': g['targets/mix/stop'].tolist(),
                'label_types': [str(v) for v in g['targets/mix/label_types']], 'unlabeled': False})
    names = [f'ex{i}' for i in range(5)] + ['mix']
    seq = [int(g[f'targets/{n}/seq_len']) for n in names]
    weak, bnd, strong = encode_targets(exs, labels, max(seq), DEV, seq_len=seq)
    for i, n in enumerate(names):
        np.testing.assert_array_equal(weak[i].cpu().numpy(), g[f'targets/{n}/weak'])
        np.testing.assert_array_equal(bnd[i, :, :seq[i]].cpu().numpy(), g[f'targets/{n}/boundary'])
        np.testing.assert_array_equal(strong[i, :, :seq[i]].cpu().numpy(), g[f'targets/{n}/strong'])
        assert not bnd[i, :, seq[i]:].any() and not strong[i, :, seq[i]:].any()


def test_tuning_drivers_reproduce_the_reference_leaderboards():
    """pb_sed_amd.tuning on the GPU against tests/golden/ref_tuning.npz (leaderboards of the reference's own
    pb_sed/models/base/tuning.py): metric values, tuned hyper-parameters and every winning score column bit for bit."""
    from tests import tuning_case
    assert tuning_case.replay(DEV) == 3 * 2 * 9


def test_model_level_tuning_wrappers_search_window_and_filter_lengths():
    """weak_label.tune_tagging / tune_boundary_detection / tune_sound_event_detection (pb_sed/models/weak_label/crnn.py:343-421) on a
    two-model FBCRNN ensemble in miniature: each equals the inference pass + the leaderboard search done by hand; the windowed
    search keeps, per class, the best (window length, median filter, tag masking) of all passes."""
    from pb_sed_amd import inference as inf, tuning
    from pb_sed_amd.models import weak_label
    from tests.stubs import make_tuning_metrics
    from tests.test_gpu_model import TINY, synth_batch
    torch.manual_seed(5)
    kw = dict(num_events=4, number_of_filters=128, hidden_size=64, num_layers=2, net=TINY)
    models = [weak_label.CRNN.build(**kw) for _ in range(2)]
    wav, seq, *_ = synth_batch(6, 16000 * 2, 4, seed=9)
    ids = [f'c{i}' for i in range(6)]
    dataset = [{'audio_data': wav[:3], 'seq_len': seq[:3].tolist(), 'example_id': ids[:3]},
               {'audio_data': wav[3:], 'seq_len': seq[3:].tolist(), 'example_id': ids[3:]}]
    classes = ['Blender', 'Cat', 'Dog', 'Speech']
    timestamps = np.round(np.arange(0, 1000) * .02, 6)
    rng = np.random.RandomState(1)
    targets = {a: np.array([(i + k) % 2 for k in range(4)], np.float64) for i, a in enumerate(ids)}     # every class has both kinds of clips
    tags = {a: np.maximum(targets[a], (rng.rand(4) < .2).astype(np.float64)) for a in ids}
    metrics = make_tuning_metrics(targets, classes)
    board = weak_label.tune_tagging(models, dataset, DEV, timestamps, classes, metrics, minimize=['leak'])
    by_hand = tuning.tune_tagging(inf.tagging(models, dataset, DEV, timestamps=timestamps, event_classes=classes), [1], metrics,
                                  minimize=['leak'], device=DEV, verbose=False)
    assert board['hit_rate'][0] == by_hand['hit_rate'][0] and board['leak'][1] == by_hand['leak'][1]
    board = weak_label.tune_boundary_detection(models, dataset, DEV, timestamps, classes, tags, metrics, [0, 4], minimize=['leak'])
    assert all(p['stepfilt_length'] in (0, 4) and p['tag_masked'] in (False, True) for p in board['leak'][1].values())
    board = weak_label.tune_sound_event_detection(models, dataset, DEV, timestamps, classes, tags, metrics, window_lengths=[10, 20],
                                                  window_shift=2, medfilt_lengths=[1, 5], minimize=['leak'], tag_masking='?')
    per_window = {}
    for win in (10, 20):
        scores = inf.sound_event_detection(models, dataset, DEV, model_kwargs={'window_length': win, 'window_shift': 2},
                                           timestamps=timestamps[::2], event_classes=classes)
        per_window[win] = tuning.tune_sound_event_detection(scores, [1, 5], tags, metrics, minimize=['leak'], tag_masking='?',
                                                            device=DEV, verbose=False)
    for name, sign in (('hit_rate', 1.), ('leak', -1.)):
        for c in classes:
            best = max(per_window[w][name][0][c] * sign for w in (10, 20))
            assert board[name][0][c] * sign == best, (name, c)
            assert board[name][1][c]['window_length'] in (10, 20) and board[name][1][c]['window_shift'] == 2
            assert np.isfinite(board[name][0][c])
