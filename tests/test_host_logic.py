"""CPU-side checks of the host logic around the kernels: config plumbing of the model API, long-clip segmenting against
vectors produced by the reference's own code, event boundary correction, batch sharding."""
import json
import os

import numpy as np
import pytest
import torch


def _reference_style_updates(num_events=10, width=1):
    """The model part of the reference's experiment config (pb_sed/experiments/weak_label_crnn/training.py:186-262)."""
    return {
        'feature_extractor': {'sample_rate': 16000, 'stft_size': 1024, 'number_of_filters': 128,
                              'n_time_masks': 1, 'max_masked_time_steps': 70, 'max_masked_time_rate': .2,
                              'n_frequency_masks': 1, 'max_masked_frequency_bands': 20, 'max_masked_frequency_rate': .2,
                              'max_noise_scale': .2},
        'cnn': {
            'cnn_2d': {'out_channels': [16 * width, 16 * width, 32 * width, 32 * width, 64 * width, 64 * width, 128 * width,
                                        128 * width, min(256 * width, 512)],
                       'pool_size': 4 * [1, (2, 1)] + [1], 'kernel_size': 3, 'residual_connections': None, 'norm': 'batch',
                       'norm_kwargs': {'eps': 1e-3}, 'activation_fn': 'relu', 'pre_activation': True, 'dropout': .0,
                       'output_layer': False},
            'cnn_1d': {'out_channels': 5 * [256 * width], 'kernel_size': [1, 3, 3, 3, 1], 'residual_connections': None,
                       'norm': 'batch', 'norm_kwargs': {'eps': 1e-3}, 'activation_fn': 'relu', 'pre_activation': True,
                       'dropout': .0, 'output_layer': False},
        },
        'rnn_fwd': {'rnn': {'hidden_size': 256 * width, 'num_layers': 2, 'dropout': .0},
                    'output_net': {'out_channels': [256 * width, num_events], 'kernel_size': 1, 'norm': 'batch',
                                   'norm_kwargs': {'eps': 1e-3}, 'activation_fn': 'relu', 'dropout': .0}},
        'labelwise_metrics': ('fscore_weak',), 'strong_fwd_bwd_loss_weight': 1.,
    }


def test_fbcrnn_config_is_completed_like_the_reference_does():
    """weak_label/crnn.py:304-340: in_channels / input_height from the extractor, the 1-D stack's width from the pooled
    height, GRU input width, rnn_bwd = rnn_fwd with reverse=True; the caller's entries win over derived ones."""
    from pb_sed_amd.models import weak_label
    cfg = weak_label.CRNN.get_config(_reference_style_updates())
    assert cfg['cnn']['cnn_2d']['in_channels'] == 1 and cfg['cnn']['input_height'] == 128
    assert cfg['cnn']['cnn_1d']['in_channels'] == 256 * 8
    assert cfg['rnn_fwd']['rnn']['input_size'] == 256 and cfg['rnn_fwd']['output_net']['in_channels'] == 256
    assert cfg['rnn_fwd']['reverse'] is False and cfg['rnn_bwd']['reverse'] is True
    assert cfg['rnn_bwd']['rnn']['hidden_size'] == 256 and cfg['rnn_bwd']['rnn']['num_layers'] == 2
    assert cfg['rnn_bwd']['output_net']['out_channels'] == [256, 10]
    assert cfg['minimum_score'] == 1e-5 and cfg['strong_fwd_bwd_loss_weight'] == 1.
    model = weak_label.CRNN.from_config(cfg)
    built = weak_label.CRNN.build()
    assert [k for k in model.state_dict()] == [k for k in built.state_dict()]
    assert sum(p.numel() for p in model.parameters()) == 3493188
    assert model.rnn_bwd.reverse and not model.rnn_fwd.reverse and model.labelwise_metrics == ('fscore_weak',)
    # rnn_bwd=None survives (the user's value wins over the derived copy)
    upd = _reference_style_updates()
    upd['rnn_bwd'] = None
    assert weak_label.CRNN.from_config(weak_label.CRNN.get_config(upd)).rnn_bwd is None


def test_bicrnn_config_tag_conditioning_and_gru_defaults():
    """strong_label/crnn.py:155-198: tag conditioning adds num_events planes / GRU inputs; the GRU defaults to one
    bidirectional layer and the experiment's explicit num_layers=2 wins (training.py:246-251)."""
    from pb_sed_amd.models import strong_label
    upd = _reference_style_updates()
    upd = {'feature_extractor': upd['feature_extractor'], 'cnn': upd['cnn'],
           'rnn': {'rnn': {'hidden_size': 256, 'num_layers': 2, 'dropout': 0.}, 'output_net': upd['rnn_fwd']['output_net']},
           'tag_conditioning': True}
    cfg = strong_label.CRNN.get_config(upd)
    assert cfg['cnn']['cnn_2d']['in_channels'] == 11 and cfg['cnn']['conditional_dims'] == 10
    assert cfg['rnn']['rnn']['input_size'] == 266 and cfg['rnn']['rnn']['bidirectional'] is True
    assert cfg['rnn']['rnn']['num_layers'] == 2 and cfg['rnn']['output_net']['in_channels'] == 512
    model = strong_label.CRNN.from_config(cfg)
    assert sum(p.numel() for p in model.parameters()) == 3899866
    del upd['rnn']['rnn']['num_layers']
    assert strong_label.CRNN.get_config(upd)['rnn']['rnn']['num_layers'] == 1
    upd['tag_conditioning'] = False
    cfg = strong_label.CRNN.get_config(upd)
    assert cfg['cnn']['cnn_2d']['in_channels'] == 1 and cfg['rnn']['rnn']['input_size'] == 256


def test_from_storage_dir_round_trip(tmp_path):
    """experiments/weak_label_crnn/inference.py:407-413: CRNN.from_storage_dir(dir, config_name='1/config.json',
    checkpoint_name=...) with the config under trainer.model and the state_dict under ['model'] of the checkpoint;
    factories stored as the REFERENCE's import paths resolve to the build's classes."""
    from pb_sed_amd.configurable import _jsonable
    from pb_sed_amd.models import weak_label
    torch.manual_seed(0)
    cfg = weak_label.CRNN.get_config(_reference_style_updates())
    model = weak_label.CRNN.from_config(cfg)
    with torch.no_grad():
        model.feature_extractor.running_mean.normal_(-7, 1)
        model.feature_extractor.running_power.copy_(model.feature_extractor.running_mean ** 2 + 4.)
    js = _jsonable(cfg)
    js['factory'] = 'pb_sed.models.weak_label.crnn.CRNN'
    js['feature_extractor']['factory'] = 'padertorch.contrib.je.modules.features.NormalizedLogMelExtractor'
    js['cnn']['factory'] = 'padertorch.contrib.je.modules.hybrid.CNN'
    js['cnn']['cnn_2d']['factory'] = 'padertorch.contrib.je.modules.conv.CNN2d'
    js['cnn']['cnn_1d']['factory'] = 'padertorch.contrib.je.modules.conv.CNN1d'
    for r in ('rnn_fwd', 'rnn_bwd'):
        js[r]['factory'] = 'padertorch.contrib.je.modules.rnn.GRU'
        js[r]['rnn']['factory'] = 'torch.nn.modules.rnn.GRU'
        js[r]['output_net']['factory'] = 'padertorch.contrib.je.modules.conv.CNN1d'
    os.makedirs(tmp_path / '1')
    os.makedirs(tmp_path / 'checkpoints')
    json.dump({'trainer': {'model': js, 'optimizer': {'lr': 5e-4}}, 'batch_size': 32}, open(tmp_path / '1' / 'config.json', 'w'))
    sd = dict(model.state_dict())
    # the reference's extractor keeps its statistics under norm.* with padertorch's broadcast shape
    for k in ('running_mean', 'running_power', 'num_tracked_values'):
        v = sd.pop(f'feature_extractor.{k}')
        sd[f'feature_extractor.norm.{k}'] = v.reshape(1, 1, -1, 1)
    sd.pop('feature_extractor.mean'), sd.pop('feature_extractor.inv_std')
    # ... and so do its batch norms: gamma / beta / statistics broadcast-shaped ([1,C,1,1] in CNN2d, [1,C,1] in CNN1d)
    # plus a num_tracked_values counter the build does not keep
    for k in [k for k in sd if '.norm.' in k and not k.startswith('feature_extractor.')]:
        lead = k.rsplit('.norm.', 1)[0] + '.norm.'
        sd[k] = sd[k].reshape((1, -1, 1, 1) if '.cnn_2d.' in k else (1, -1, 1))
        sd[lead + 'num_tracked_values'] = torch.zeros(1)
    assert any(k.endswith('.norm.num_tracked_values') and 'cnn_2d' in k for k in sd)
    torch.save({'model': sd, 'iteration': 1234}, tmp_path / 'checkpoints' / 'ckpt_best_macro_fscore_weak.pth')
    loaded = weak_label.CRNN.from_storage_dir(str(tmp_path), config_name='1/config.json',
                                              checkpoint_name='ckpt_best_macro_fscore_weak.pth')
    for (k, a), (_, b) in zip(model.state_dict().items(), loaded.state_dict().items()):
        if k in ('feature_extractor.mean', 'feature_extractor.inv_std'):
            continue
        assert torch.equal(a, b), k
    fe = loaded.feature_extractor
    assert torch.allclose(fe.mean, fe.running_mean)
    assert torch.allclose(fe.inv_std, 1 / torch.sqrt(fe.running_power - fe.running_mean ** 2 + 1e-5))


def test_normalization_accepts_scale_shift_spelling_in_a_strict_load():
    """a checkpoint whose Normalization stores the affine pair as scale / shift (broadcast-shaped, with a
    num_tracked_values counter) loads STRICTLY - no missing or unexpected keys - into gamma / beta."""
    from pb_sed_amd import modules
    torch.manual_seed(0)
    net = modules.CNN2d(1, [4, 6], 3, pool_size=1)
    sd = {}
    for k, v in net.state_dict().items():
        if k.endswith('.norm.gamma'):
            sd[k[:-5] + 'scale'] = torch.randn_like(v).reshape(1, -1, 1, 1)
        elif k.endswith('.norm.beta'):
            sd[k[:-4] + 'shift'] = torch.randn_like(v).reshape(1, -1, 1, 1)
        else:
            sd[k] = torch.randn_like(v).reshape((1, -1, 1, 1)) if '.norm.running' in k else torch.randn_like(v)
    norms = [k for k in sd if k.endswith('.norm.scale')]
    assert norms
    for k in norms:
        sd[k[:-5] + 'num_tracked_values'] = torch.zeros(1)
    want = {k: v.clone() for k, v in sd.items()}
    missing, unexpected = net.load_state_dict(sd, strict=True)
    assert not missing and not unexpected
    for k in norms:
        assert torch.equal(dict(net.state_dict())[k[:-5] + 'gamma'], want[k].reshape(-1))
        assert torch.equal(dict(net.state_dict())[k[:-5] + 'beta'], want[k[:-5] + 'shift'].reshape(-1))


def test_unsupported_reference_options_fail_loudly():
    from pb_sed_amd import modules
    with pytest.raises(NotImplementedError, match='dropout'):
        modules.CNN2d(1, [16], 3, dropout=.1)
    with pytest.raises(NotImplementedError, match='activation_fn'):
        modules.CNN1d(4, [16], 1, activation_fn='leaky_relu')


def test_segment_batch_and_merge_vs_reference_vectors(golden):
    """pb_sed/utils/segment.py executed by tests/golden/gen_golden.py (under a restated padertorch Segmenter): segment
    ids / lengths / trimmed inputs and the merged outputs for several (max_length, overlap) pairs."""
    from pb_sed_amd.utils import segment as sg
    g = golden('ref_segments.npz')
    stft, seq, ids = g['stft'], g['seq_len'].tolist(), g['ids'].tolist()
    for max_len, overlap in ((12, 2), (20, 5), (16, 0), (64, 4)):
        tag = f'seg_{max_len}_{overlap}'
        segs = sg.segment_batch({'example_id': ids, 'stft': stft, 'seq_len': seq}, max_len, overlap)
        assert len(segs) == int(g[f'{tag}/n'])
        for i, s in enumerate(segs):
            np.testing.assert_array_equal(np.asarray(s['stft']), g[f'{tag}/{i}/stft'])
            assert s['seq_len'] == g[f'{tag}/{i}/seq_len'].tolist()
            assert s['example_id'] == g[f'{tag}/{i}/ids'].tolist()
        if len(segs) > 1:
            out2 = {aid: np.asarray(s['stft'])[j, 0, :sl, :, 0] for s in segs
                    for j, (aid, sl) in enumerate(zip(s['example_id'], s['seq_len']))}
            merged = sg.merge_segments(out2, overlap)
            merged3 = sg.merge_segments({a: np.stack([v, 2 * v]) for a, v in out2.items()}, overlap)
            for a in ids:
                np.testing.assert_array_equal(merged[a], g[f'{tag}/merged/{a}'])
                np.testing.assert_array_equal(merged3[a], g[f'{tag}/merged3/{a}'])


def test_audio_segments_cover_exactly_the_frames_of_the_stft_segments():
    """The waveform form of segment_batch: a slice plus its 'stft_pad_front' must frame to the same samples as frames
    [start, start + T_seg) of the whole clip (frame t covers samples [320 t - 320, 320 t + 640))."""
    from pb_sed_amd.modules import num_frames
    from pb_sed_amd.utils import segment as sg
    n = 16000 * 4 + 77
    audio = torch.arange(2 * n, dtype=torch.float32).reshape(2, n)
    t = num_frames(n)
    segs = sg.segment_batch({'example_id': ['a', 'b'], 'audio_data': audio, 'seq_len': [t, t - 30]}, 64, 8)
    assert len(segs) > 2
    padded = torch.nn.functional.pad(audio, (320, 960))
    for s in segs:
        x = torch.nn.functional.pad(s['audio_data'], (s['stft_pad_front'], 960))
        for j in (0, s['num_frames'] - 1):
            g = s['segment_start'] + j
            assert torch.equal(x[:, 320 * j:320 * j + 960], padded[:, 320 * g:320 * g + 960]), (s['segment_start'], j)


def test_shift_and_widen_events():
    """experiments/strong_label_crnn/inference.py:177-184."""
    from pb_sed_amd.inference import shift_and_widen_events
    ev = {'a': [(0.1, 0.5, 'dog'), (2.0, 2.1, 'cat'), (3.0, 3.2, 'dog')], 'b': []}
    out = shift_and_widen_events(ev, pseudo_widening=.1, onset_bias={'dog': .15}, offset_bias={'dog': -.05, 'cat': .4})
    assert out['b'] == []
    assert out['a'][0] == (0., pytest.approx(0.65), 'dog')          # onset clamped at 0
    assert all(lbl != 'cat' for *_, lbl in out['a'])                 # 2.0-.1 .. 2.1+.1-.4 is empty
    assert out['a'][1] == (pytest.approx(2.75), pytest.approx(3.35), 'dog')
    assert shift_and_widen_events(ev) == {'a': ev['a'], 'b': []}


def test_shard_batch_with_remainder():
    from pb_sed_amd.trainer import shard_batch
    batch = {'audio_data': torch.arange(7)[:, None], 'seq_len': list(range(7)), 'example_id': list('abcdefg'), 'meta': 1}
    got = [shard_batch(batch, r, 3) for r in range(3)]
    assert [len(s['seq_len']) for s in got] == [3, 2, 2]
    assert sum((s['example_id'] for s in got), []) == list('abcdefg')
    assert torch.equal(torch.cat([s['audio_data'] for s in got]), batch['audio_data']) and got[1]['meta'] == 1
    assert [len(shard_batch(batch, r, 8)['seq_len']) for r in range(8)] == [1] * 7 + [0]


def test_sharded_inference_skips_an_empty_share():
    """A ragged last batch with fewer clips than ranks: the rank whose share is empty neither segments nor launches
    (pb_sed_amd/inference.py; the reference has no sharding)."""
    from pb_sed_amd.inference import inference

    class Untouchable:
        def to(self, device):
            return self

        def eval(self):
            return self

        def tagging(self, *a, **k):
            raise AssertionError('the model must not run on an empty share')

        example_to_device = tagging

    batch = {'audio_data': torch.zeros(2, 16000), 'seq_len': [50, 50], 'example_id': ['a', 'b']}
    for seg in (None, 20):
        assert inference(Untouchable(), 'tagging', [batch], 'cpu', max_segment_length=seg, merge_score_segments=seg is not None,
                         rank=2, world_size=3) == {}


def test_mel_warping_is_monotone_and_fixes_the_band_edge():
    from pb_sed_amd.modules import LogTruncatedNormal, MelWarping, TruncatedExponential
    w = MelWarping(LogTruncatedNormal(scale=.08, truncation=np.log(1.3), seed=1), TruncatedExponential(scale=.5, truncation=5., seed=2), 8000.)
    f = np.linspace(50., 8000., 130)
    out = w(f, 16)
    assert out.shape == (16, 130) and (np.diff(out, axis=-1) > 0).all()
    np.testing.assert_allclose(out[:, -1], 8000., rtol=1e-9)
    assert (np.abs(out[:, 1] / f[1] - 1) < .3 + 1e-9).all() and np.ptp(out[:, 10]) > 0


def test_superpose_placement_draws_match_the_reference(golden):
    """Host half of SuperposeEvents (mix.py:95-117): with the reference's seed the same offsets are drawn, i.e. the
    shifted event boundaries equal the reference's (the device half - the mixing - is in tests/test_gpu_postproc.py)."""
    from pb_sed_amd.data import SuperposeEvents, add_label_types, samples_to_frames
    g = golden('ref_data_front_end.npz')
    for name in ('m01', 'm203', 'm41'):
        kw = eval(str(g[f'{name}/kw']))
        idx = g[f'{name}/idx']
        np.random.seed(int(g[f'{name}/seed']))
        starts, stops = SuperposeEvents(**kw).place([g[f'ex{i}/audio'].shape[1] for i in idx])
        shifted = [int(v + s) for i, s in zip(idx, starts) for v in g[f'ex{i}/start']]
        assert shifted == g[f'{name}/start'].tolist()
        assert int(stops.max()) == g[f'{name}/audio'].shape[1]
    ex = add_label_types({'audio_data': np.zeros((1, 1000)), 'events': ['a']})
    assert ex['events_stop_samples'] == [1000] and ex['label_types'] == ['weak'] and ex['unlabeled'] is False
    assert add_label_types({'audio_data': np.zeros((1, 10))})['unlabeled'] is True
    assert samples_to_frames([0, 319, 320, 641], [320, 321, 640, 641]) == ([0, 0, 1, 2], [1, 2, 2, 3])


def test_time_warp_maps_are_inverse_monotone_and_identity_without_shift():
    """pb_sed_amd.data.TimeWarp (time-warped STFT of pb_sed/data_preparation/transform.py:36-45, samplers of
    provider.py:329-338): no shift -> the base STFT's regular grid; the event map inverts the frame map; frames advance
    monotonically and stay within the stretch factors the samplers allow."""
    from pb_sed_amd import data, modules
    tw = data.TimeWarp(modules.Uniform(.4, .6, seed=1), modules.Uniform(-.1, .1, seed=2))
    n, t = 160000, modules.num_frames(160000)
    pos = tw.frame_positions(n, t, np.array([.5, .4]), np.array([0., 0.]))
    assert (pos == (np.arange(t) * 320 - 320)[None]).all()          # 'half' fading: frame t starts at 320 t - 320
    a, s = tw.sample(64)
    assert ((a >= .4) & (a <= .6)).all() and (np.abs(s) <= .1 + 1e-12).all()
    pos = tw.frame_positions(n, t, a, s)
    hop = np.diff(pos, axis=1)
    assert (hop > 0).all() and hop.min() >= 320 * .4 / .7 - 1 and hop.max() <= 320 * .6 / .3 + 1
    v = np.linspace(0, n, 57)
    for i in range(8):
        u = tw.warped_of(v, n, a[i], s[i])
        assert (np.diff(u) > 0).all() and abs(u[0]) < 1e-9 and abs(u[-1] - n) < 1e-6
        np.testing.assert_allclose(tw.source_of(u, n, a[i], s[i]), v, atol=1e-6)
    # an event keeps covering the audio it labelled: the frames inside the warped event look at source samples inside it
    ex = [{'audio_data': np.zeros((1, n)), 'events': ['x'], 'events_start_samples': [40000], 'events_stop_samples': [90000]}]
    tw2 = data.TimeWarp(lambda shape: np.full(shape, .45), lambda shape: np.full(shape, .08))
    fp, warped = tw2(ex, t)
    (on,), (off,) = data.samples_to_frames(warped[0]['events_start_samples'], warped[0]['events_stop_samples'])
    centre = fp[0, on + 1:off - 1] + 480
    assert (centre >= 40000 - 320).all() and (centre <= 90000 + 320).all()
    outside = np.r_[fp[0, :max(on - 2, 0)] + 480, fp[0, off + 2:] + 480]
    assert ((outside < 40000) | (outside > 90000)).all()


def test_dynamic_bucket_batcher_rules():
    """pb_sed_amd.data.DynamicBucketBatcher (pb_sed/data_preparation/fetcher.py:38-51): padding bound, batch size, sort
    order, every example exactly once, drop_incomplete, expiration, buffer limit, label diversity."""
    from pb_sed_amd import data
    rng = np.random.RandomState(0)
    k = 5
    exs = [{'example_id': i, 'seq_len': int(n), 'dataset': 'a' if i % 3 else 'b',
            'weak_targets': np.eye(k)[rng.randint(k)]} for i, n in enumerate(rng.randint(100, 501, 400))]
    batches = list(data.DynamicBucketBatcher(8, max_padding_rate=.05)(exs))
    ids = [ex['example_id'] for b in batches for ex in b]
    assert sorted(ids) == list(range(400))
    for b in batches:
        lens = [ex['seq_len'] for ex in b]
        assert len(b) <= 8 and lens == sorted(lens, reverse=True)
        assert min(lens) >= max(lens) * (1 - .05) - 1e-9
    assert sum(len(b) == 8 for b in batches) >= 35
    full = list(data.DynamicBucketBatcher(8, max_padding_rate=.05, drop_incomplete=True)(exs))
    assert full and all(len(b) == 8 for b in full) and len(full) == sum(len(b) == 8 for b in batches)
    # expiration closes a lonely bucket after that many further examples; the buffer limit bounds what waits
    odd = [dict(exs[0], seq_len=5000, example_id=-1)] + exs[:50]
    out = list(data.DynamicBucketBatcher(8, expiration=10)(odd))
    first_small = next(i for i, b in enumerate(out) if b[0]['example_id'] == -1)
    assert len(out[first_small]) == 1 and sum(len(b) for b in out[:first_small]) <= 10
    waiting, longest = 0, 0
    bb = data.DynamicBucketBatcher(8, max_padding_rate=.01, max_buffered_examples=20)
    emitted = 0
    for i, b in enumerate(bb(exs)):
        emitted += len(b)
    assert emitted == 400
    # label diversity: every full batch covers at least 3 classes
    div = list(data.DynamicBucketBatcher(8, max_padding_rate=.5, min_label_diversity=3, drop_incomplete=True)(exs))
    assert div and all(len({int(np.argmax(ex['weak_targets'])) for ex in b}) >= 3 for b in div)
    # examples per dataset
    mix = list(data.DynamicBucketBatcher(8, max_padding_rate=.5, min_dataset_examples={'b': 2}, drop_incomplete=True)(exs))
    assert mix and all(sum(ex['dataset'] == 'b' for ex in b) >= 2 for b in mix)


def test_collate_pads_along_time():
    from pb_sed_amd import data
    batch = [{'example_id': 'a', 'seq_len': 7, 'audio_data': np.ones((1, 50), np.float32), 'stft': np.ones((1, 7, 3, 2), np.float32)},
             {'example_id': 'b', 'seq_len': 5, 'audio_data': np.ones((1, 40), np.float32), 'stft': np.ones((1, 5, 3, 2), np.float32)}]
    out = data.collate(batch)
    assert out['example_id'] == ['a', 'b'] and out['seq_len'] == [7, 5]
    assert out['audio_data'].shape == (2, 1, 50) and out['audio_data'][1, 0, 40:].abs().sum() == 0
    assert out['stft'].shape == (2, 1, 7, 3, 2) and out['stft'][1, 0, 5:].abs().sum() == 0 and out['stft'][1, 0, :5].min() == 1


def test_device_prefetcher_passes_batches_through_in_order():
    """data.DevicePrefetcher on a CPU device: same batches, same order, one batch pulled ahead at most (the GPU path - side
    stream, events - is exercised by tests/test_gpu_model.py)."""
    from pb_sed_amd.data import DevicePrefetcher
    pulled = []

    def loader():
        for i in range(4):
            pulled.append(i)
            yield {'audio_data': torch.full((2, 8), float(i)), 'seq_len': [3, 2], 'example_id': [f'a{i}', f'b{i}']}
    seen = []
    for batch in DevicePrefetcher(loader(), 'cpu'):
        seen.append(int(batch['audio_data'][0, 0]))
        assert len(pulled) <= len(seen) + 1
        assert batch['seq_len'] == [3, 2]
    assert seen == [0, 1, 2, 3] and pulled == [0, 1, 2, 3]
    assert list(DevicePrefetcher([], 'cpu')) == []


def _strict_loads(text):
    def refuse(tok):
        raise ValueError(f'non-strict JSON constant {tok}')
    return json.loads(text, parse_constant=refuse)


@pytest.mark.parametrize('canned', ['r04_bench_c2.json', 'r04_bench_c3.json', 'r04_bench_c5.json', 'r04_bench_deep.json'])
def test_bench_contract_line_is_compact(canned, capsys, tmp_path, monkeypatch):
    """bench.py's stdout is ONE short strict-JSON line with the contract keys, `roofline` and `cpu_baseline` (round 4's
    23.7 KB line was not parsed by the driver); the full result goes to stderr / gpurun_out/bench_detail.json."""
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    sys.path.insert(0, root)
    import bench
    full = json.load(open(os.path.join(root, 'profiles', canned)))
    full['roofline']['achieved'] = float('nan')                        # a stray NaN must not reach the line
    if canned.endswith('c2.json'):
        assert 'other_configs' in full and 'cpu_baseline' in full
        full['rendezvous'] = {'backend': 'nccl (RCCL)', 'ranks_seen': 8, 'allreduce_check': 36.0}
        full['allreduce'] = {'implementation': 'pbsed_allreduce', 'bytes_per_step': 13972752, 'exposed_ms_per_step': 0.05,
                             'busbw_GBs_if_fully_exposed': 400.0, 'time_at_ring_bound_ms': 0.16, 'note': 'x' * 500, 'buckets': ['a'] * 40}
    monkeypatch.setattr(bench, 'ROOT', str(tmp_path))
    bench.emit(full)
    cap = capsys.readouterr()
    lines = [l for l in cap.out.split('\n') if l]
    assert len(lines) == 1
    assert len(lines[0]) < 6000, len(lines[0])
    line = _strict_loads(lines[0])
    for key in ('metric', 'value', 'unit', 'n_gpus', 'steps', 'warmup', 'ms_per_step', 'higher_is_better', 'scaling',
                'vs_baseline', 'dtype', 'data', 'config', 'roofline'):
        assert key in line, key
    assert line['dtype'] in ('f32', 'bf16', 'bf16x3')
    assert 'workload' in line['config']
    assert 'frac' in line['roofline'] and 'traffic' in line['roofline'] and 'note' not in line['roofline']
    assert line['roofline']['achieved'] is None
    if 'cpu_baseline' in full:
        assert line['cpu_baseline']['value'] == full['cpu_baseline']['value']
        assert len(line['cpu_baseline']['sample']) <= 100
    if canned.endswith('c2.json'):
        assert set(line['other_configs']) == {'c3', 'c5', 'deep'}
        assert line['rendezvous']['ranks_seen'] == 8 and line['allreduce']['exposed_ms_per_step'] == 0.05
        assert 'note' not in line['allreduce']
    detail = _strict_loads(open(tmp_path / 'gpurun_out' / 'bench_detail.json').read())
    assert detail['metric'] == full['metric']


def test_bench_contract_line_degrades_instead_of_asserting():
    """an oversized line (unbounded `rendezvous` / `loss` / `threads_scan` / strings) must still reach stdout with the
    contract keys, `roofline` and `cpu_baseline` - optional objects are dropped in a fixed order and named in `dropped`."""
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    sys.path.insert(0, root)
    import bench
    full = json.load(open(os.path.join(root, 'profiles', 'r05a_bench_c2.json'))) if os.path.exists(os.path.join(root, 'profiles', 'r05a_bench_c2.json')) \
        else json.load(open(os.path.join(root, 'profiles', 'r04_bench_c2.json')))
    full['rendezvous'] = {'hosts': ['node-%04d' % i for i in range(2000)]}
    full['loss'] = [0.5] * 3000
    full.setdefault('cpu_baseline', {'value': 1.0, 'unit': 'clips/s', 'cores': 8, 'kind': 'port'})['threads_scan'] = {str(n): {'clips_per_s': 1.0, 'note': 'y' * 400} for n in range(64)}
    full['cpu_baseline']['sample'] = 'z' * 5000
    text = bench.fit_line(bench.contract_line(full))
    assert len(text) < bench.LINE_LIMIT and '\n' not in text
    line = _strict_loads(text)
    for key in ('metric', 'value', 'unit', 'n_gpus', 'steps', 'warmup', 'ms_per_step', 'higher_is_better', 'scaling', 'vs_baseline', 'dtype',
                'data', 'config', 'roofline', 'cpu_baseline'):
        assert key in line, key
    assert line['value'] == full['value'] and line['cpu_baseline']['value'] == full['cpu_baseline']['value']
    assert 'rendezvous' in line['dropped'] and 'loss' in line['dropped'] and 'rendezvous' not in line
    # a line that fits is passed through unchanged
    small = bench.contract_line(json.load(open(os.path.join(root, 'profiles', 'r04_bench_c3.json'))))
    assert _strict_loads(bench.fit_line(small)) == _strict_loads(json.dumps(small)) and 'dropped' not in small


def test_bench_scan_poll_volume_model():
    """bench.scan_polled_bytes: the poll volume of one persistent scan launch as DESIGN.md section 3 counts it - every ring /
    projection workgroup (16 units x 16 rows) polls its source's whole 16 x H state tile per step, upper-layer gate threads
    their projected inputs (3 words forward, 1 word BPTT)."""
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    sys.path.insert(0, root)
    import bench
    # configs[1]: 2 chains x 2 layers, T 500, B 32, H 256: (8 rings + 4 projection groups) x 16 blocks... = 192 blocks x 16 KB per step
    fwd = bench.scan_polled_bytes('forward_scan', 2, 2, 500, 32, 256)
    assert fwd == 500 * (192 * 16 * 256 * 4 + 2 * 2 * 16 * 256 * 4 * 3)
    bwd = bench.scan_polled_bytes('bptt_scan', 2, 2, 500, 32, 256)
    assert bwd == 500 * (192 * 16 * 256 * 4 + 2 * 2 * 16 * 256 * 4 * 1)
    # one-layer BiGRU scans (configs[2]): rings only
    assert bench.scan_polled_bytes('forward_scan', 2, 1, 500, 32, 256) == 500 * 64 * 16 * 256 * 4
    new_state = 500 * 2 * 2 * 32 * 256 * 4
    assert 24 < fwd / new_state < 27                      # 16 ring blocks + 16 projection blocks each read the whole state tile


def test_trainer_checkpoint_round_trip_in_padertorch_layout(tmp_path):
    """Trainer.save_checkpoint / load_checkpoint (checkpoint / resume): the file is a padertorch-style
    trainer checkpoint - ``ckpt['model']`` is what ``CRNN.from_storage_dir`` loads (experiments/weak_label_crnn/inference.py:
    407-413), ``ckpt['optimizer']`` is a ``torch.optim.Adam`` state_dict over ``model.parameters()`` - and a second Trainer
    resumes from it with identical parameters, Adam moments and iteration count."""
    from pb_sed_amd.models import weak_label
    from pb_sed_amd.trainer import Trainer
    net = dict(out_channels_2d=[16, 16, 32], pool_sizes_2d=[1, (2, 1), (2, 1)], kernel_size_2d=3, out_channels_1d=[64, 64],
               kernel_size_1d=[1, 3])
    kw = dict(num_events=10, hidden_size=64, num_layers=2, net=net)
    torch.manual_seed(1)
    model = weak_label.CRNN.build(**kw)
    tr = Trainer(model, lr=3e-4, betas=(.8, .99))
    with torch.no_grad():                         # as after 17 steps
        tr.m.normal_(0, 1e-3), tr.v.uniform_(0, 1e-5)
        model.feature_extractor.running_mean.normal_(-6, 1), model.feature_extractor.num_tracked_values.fill_(4242.)
    tr.iteration = 17
    path = tr.save_checkpoint(str(tmp_path))
    assert path.endswith(os.path.join('checkpoints', 'ckpt_17.pth')) and os.path.exists(path)
    ckpt = torch.load(path, weights_only=False)
    assert set(ckpt) >= {'model', 'iteration', 'optimizer'} and ckpt['iteration'] == 17
    # the optimizer part is a torch.optim.Adam state_dict over model.parameters()
    ref_model = weak_label.CRNN.build(**kw)
    adam = torch.optim.Adam(ref_model.parameters(), lr=1.)
    adam.load_state_dict(ckpt['optimizer'])
    assert adam.param_groups[0]['lr'] == 3e-4 and tuple(adam.param_groups[0]['betas']) == (.8, .99)
    p0 = next(iter(ref_model.parameters()))
    assert torch.equal(adam.state[p0]['exp_avg'], tr.m[:p0.numel()].view(p0.shape)) and float(adam.state[p0]['step']) == 17.
    # a fresh Trainer resumes from it
    torch.manual_seed(2)
    model2 = weak_label.CRNN.build(**kw)
    tr2 = Trainer(model2)
    tr2.load_checkpoint(path)
    assert tr2.iteration == 17 and tr2.lr == 3e-4 and tuple(tr2.betas) == (.8, .99)
    assert torch.equal(tr2.flat_param, tr.flat_param) and torch.equal(tr2.m, tr.m) and torch.equal(tr2.v, tr.v)
    for (k, a), (_, b) in zip(model.state_dict().items(), model2.state_dict().items()):
        assert torch.equal(a, b), k
    # ... and its cumulative-statistics baseline is the loaded state (not the constructor's)
    base_n, _ = tr2._synced_stats['feature_extractor.']
    assert base_n.item() == 4242.
    # a checkpoint without optimizer state (e.g. a model-only file) restarts the moments
    torch.save({'model': ckpt['model'], 'iteration': 3}, tmp_path / 'model_only.pth')
    tr2.load_checkpoint(str(tmp_path / 'model_only.pth'))
    assert tr2.iteration == 3 and not tr2.m.any() and not tr2.v.any()


def test_parked_kernel_patches_still_apply_in_stack_order(tmp_path):
    """tools/micro/attic holds kernel patches that wait for a GPU measurement; they are stacked in the order
    tools/build_variants.sh applies them.  A tree edit that makes one of them rot should fail here, not on the GPU box."""
    import re
    import shutil
    import subprocess
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    script = open(os.path.join(root, 'tools', 'build_variants.sh')).read()
    patches = re.findall(r'attic/(\w+\.patch)\)', script)
    assert len(patches) >= 9 and patches[0] == 'scalar_tile_loads.patch'
    work = tmp_path / 'pb_sed_amd'
    work.mkdir()
    shutil.copytree(os.path.join(root, 'pb_sed_amd', 'csrc'), work / 'csrc', ignore=shutil.ignore_patterns('build'))
    for p in patches:
        r = subprocess.run(['patch', '-s', '-p1', '-i', os.path.join(root, 'tools', 'micro', 'attic', p)], cwd=tmp_path,
                           capture_output=True, text=True)
        assert r.returncode == 0, (p, r.stdout[-500:], r.stderr[-500:])
    # every patch listed in the attic README is in the stack, and nothing else is parked silently
    parked = sorted(f for f in os.listdir(os.path.join(root, 'tools', 'micro', 'attic')) if f.endswith('.patch'))
    assert parked == sorted(patches)


def _jsonable(x):
    """tuples -> lists, recursively (the fixture went through JSON)"""
    if isinstance(x, dict):
        return {k: _jsonable(v) for k, v in x.items()}
    if isinstance(x, (list, tuple)):
        return [_jsonable(v) for v in x]
    return x


def test_pseudo_label_hand_off_matches_the_reference(tmp_path):
    """pb_sed_amd.pseudo_label against tests/golden/ref_pseudo_label.json - inputs and outputs of the reference's OWN
    pb_sed/models/base/pseudo_label.py (six flag combinations over 14 clips incl. an untagged clip, a tagged clip without detections,
    detections of classes a clip is not tagged with): every relabelled example equal, the input never modified, the untouched
    dataset handed back as the same object when nothing is asked for; the TSV of
    pb_sed/experiments/strong_label_crnn/inference.py:393-400 row by row."""
    import copy
    from pb_sed_amd import pseudo_label as pl
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    fx = json.load(open(os.path.join(root, 'tests', 'golden', 'ref_pseudo_label.json')))
    events = {a: [tuple(e) for e in v] for a, v in fx['events'].items()}
    boundaries = {a: [tuple(e) for e in v] for a, v in fx['boundaries'].items()}
    assert set(fx['cases']) == {'none', 'tags', 'events', 'boundaries', 'tags_events', 'tags_boundaries'}
    for name, case in fx['cases'].items():
        dataset = copy.deepcopy(fx['dataset'])
        before = copy.deepcopy(dataset)
        out = pl.pseudo_label(dataset, fx['event_classes'], *case['flags'], fx['tags'], boundaries, events)
        assert dataset == before, name                                   # relabelling works on a copy
        assert (out is dataset) == case['same_object'], name
        assert _jsonable(out) == case['dataset'], name
    with pytest.raises(AssertionError):
        pl.pseudo_label(fx['dataset'], fx['event_classes'], False, True, True, fx['tags'], boundaries, events)
    rates = pl.label_rates(pl.pseudo_label(fx['dataset'], fx['event_classes'], True, True, False, fx['tags'], boundaries, events))
    assert rates['label_rate'] == pytest.approx(0.7857142857142857) and rates['boundaries'] == pytest.approx(0.5217391304347826)
    assert rates['weak'] == pytest.approx(0.4782608695652174) and rates['strong'] == 0.      # what the reference printed for 'tags_boundaries'
    path = tmp_path / 'validation_pseudo_labeled.tsv'
    pl.write_event_tsv(path, {'a': [(0.5, 1.25, 'Dog'), (2.0, 3.0, 'Cat')], 'b': []})
    assert open(path).read() == 'filename\tonset\toffset\tevent_label\na.wav\t0.5\t1.25\tDog\na.wav\t2.0\t3.0\tCat\nb.wav\t\t\t\n'


def test_tuned_hyper_params_round_trip_into_the_inference_arguments(tmp_path):
    """The JSON files the tuning drivers write (`<stage>_hyper_params_<metric>.json`) read back into what ONE ensemble pass over
    several tuned parameter sets takes (pb_sed/experiments/weak_label_crnn/inference.py:213-262, strong_label_crnn/inference.py:117-122):
    [variants, classes] arrays of median-filter lengths / tag masking / window lengths, a single window shift (two are refused),
    thresholds per variant (None for a metric without one); tags from tuned thresholds; the boundaries path's millisecond rounding."""
    import json
    from pb_sed_amd import inference as inf
    classes = ['Cat', 'Dog']
    f = {'Cat': {'medfilt_length': 5, 'tag_masked': True, 'window_length': 20, 'window_shift': 4, 'threshold': .4, 'f': .7},
         'Dog': {'medfilt_length': 11, 'tag_masked': False, 'window_length': 40, 'window_shift': 4, 'threshold': .6, 'f': .8}}
    psds = {c: {'medfilt_length': 1, 'tag_masked': False, 'window_length': 10, 'window_shift': 4, 'psds1': .3} for c in classes}
    for name, hp in (('f', f), ('psds1', psds)):
        json.dump(hp, open(tmp_path / f'sed_hyper_params_{name}.json', 'w'))
    hps = inf.load_hyper_params(tmp_path, 'sed', ['f', 'psds1'])
    arrs = inf.sed_hyper_param_arrays(hps, classes)
    assert arrs['medfilt_length'].tolist() == [[5, 11], [1, 1]] and arrs['apply_mask'].tolist() == [[1, 0], [0, 0]]
    assert arrs['model_kwargs']['window_length'].tolist() == [[20, 40], [10, 10]] and arrs['model_kwargs']['window_shift'] == 4
    assert arrs['timestamp_stride'] == 4 and arrs['thresholds'] == [{'Cat': .4, 'Dog': .6}, None]
    psds['Dog']['window_shift'] = 2
    with pytest.raises(ValueError):
        inf.sed_hyper_param_arrays([f, psds], classes)
    plain = inf.sed_hyper_param_arrays({c: {'medfilt_length': 3, 'tag_masked': True, 'threshold': .5} for c in classes}, classes)
    assert plain['model_kwargs'] is None and plain['timestamp_stride'] == 1 and plain['medfilt_length'].shape == (1, 2)
    tags, scores = inf.tags_from_scores({'a': np.array([[.5, .5]], np.float32), 'b': np.array([[.3, .9]], np.float32)}, f, classes)
    assert tags['a'].tolist() == [True, False] and tags['b'].tolist() == [False, True] and scores['b'].shape == (2,)
    ev = {'a': [(0.1234, 1.0, 'Dog'), (2.0, 2.0004, 'Cat')]}
    out = inf.shift_and_widen_events(ev, 0., {'Dog': .2}, {'Dog': 0.}, decimals=3)
    assert out == {'a': [(0.0, 1.0, 'Dog')]}                 # onset clamped at 0; the 0.4 ms event rounds to nothing
    assert inf.shift_and_widen_events(ev, 0., {'Dog': .02}, None, decimals=3)['a'][0][0] == 0.103
