"""Pin the oracle's restatement of reference-OWNED maths against golden vectors produced by the
reference's own source (tools/gen_golden.py -> tests/golden/ref_*.npz)."""
import ast

import numpy as np
import pytest
import torch

from oracle import models as om
from oracle import postproc as pp
from tests.stubs import StubCNN, StubFeatures, StubRNN

LOSS_CASES = ['ragged_strong', 'full_len', 'no_bwd', 'slat', 'weak_only', 'half_weight_smooth',
              'class_weights']


@pytest.mark.parametrize('name', LOSS_CASES)
def test_fbcrnn_loss_and_grads(golden, name):
    g = golden('ref_fbcrnn_loss.npz')
    kw = ast.literal_eval(str(g[f'{name}/kw']))
    yf = torch.tensor(g[f'{name}/y_fwd'], requires_grad=True)
    yb = torch.tensor(g[f'{name}/y_bwd'], requires_grad=True) if f'{name}/y_bwd' in g else None
    seq_len = g[f'{name}/seq_len']
    m = om.FBCRNN(None, None, None, None, **kw)
    out = (yf, yb, seq_len, torch.zeros(len(seq_len), 1, 4, yf.shape[-1]), seq_len,
           (torch.tensor(g[f'{name}/weak_targets']), torch.tensor(g[f'{name}/boundary_targets'])))
    review = m.review({'seq_len': seq_len.tolist()}, out)
    review['loss'].backward()
    assert review['loss'].item() == pytest.approx(float(g[f'{name}/loss']), rel=1e-6)
    np.testing.assert_allclose(yf.grad.numpy(), g[f'{name}/grad_y_fwd'], rtol=1e-5, atol=1e-8)
    if yb is not None:
        np.testing.assert_allclose(yb.grad.numpy(), g[f'{name}/grad_y_bwd'], rtol=1e-5, atol=1e-8)
    np.testing.assert_array_equal(review['buffers']['y_weak'], g[f'{name}/y_weak'])
    np.testing.assert_array_equal(review['buffers']['targets_weak'], g[f'{name}/targets_weak'])
    assert review['scalars']['weak_label_rate'] == float(g[f'{name}/weak_label_rate'])
    assert review['scalars']['boundary_label_rate'] == float(g[f'{name}/boundary_label_rate'])


@pytest.mark.parametrize('name', ['a', 'b'])
def test_bicrnn_loss_and_grads(golden, name):
    g = golden('ref_bicrnn_loss.npz')
    y = torch.tensor(g[f'{name}/y'], requires_grad=True)
    loss = om.bicrnn_loss(y, g[f'{name}/seq_len'], torch.tensor(g[f'{name}/strong_targets']))
    loss.backward()
    assert loss.item() == pytest.approx(float(g[f'{name}/loss']), rel=1e-6)
    np.testing.assert_allclose(y.grad.numpy(), g[f'{name}/grad_y'], rtol=1e-5, atol=1e-9)


@pytest.mark.parametrize('bwd', [True, False])
def test_fbcrnn_heads(golden, bwd):
    g = golden('ref_fbcrnn_heads.npz')
    tag = 'fb' if bwd else 'f'
    m = om.FBCRNN(StubFeatures(), StubCNN(), StubRNN(g['a_fwd'], False),
                  StubRNN(g['a_bwd'], True) if bwd else None).eval()
    inputs = {'stft': torch.tensor(g['h']), 'seq_len': g['seq_len'].tolist()}
    with torch.no_grad():
        y, sl = m.tagging(dict(inputs))
        np.testing.assert_array_equal(y.numpy(), g[f'{tag}/tagging'])
        np.testing.assert_array_equal(sl, g[f'{tag}/tagging_seq_len'])
        if bwd:
            y, sl = m.boundaries_detection(dict(inputs))
            np.testing.assert_array_equal(y.numpy(), g[f'{tag}/boundaries'])
        for wl, ws in ((5, 1), (4, 2), (1, 1), (7, 3)):
            y, sl = m.sound_event_detection(dict(inputs), wl, ws)
            np.testing.assert_array_equal(y.numpy(), g[f'{tag}/sed_{wl}_{ws}'])
            np.testing.assert_array_equal(sl, g[f'{tag}/sed_{wl}_{ws}_seq_len'])
        y, _ = m.sound_event_detection(dict(inputs), g['wl_1d'].tolist(), 1)
        np.testing.assert_array_equal(y.numpy(), g[f'{tag}/sed_1d'])
        y, _ = m.sound_event_detection(dict(inputs), g['wl_2d'].tolist(), 2)
        np.testing.assert_array_equal(y.numpy(), g[f'{tag}/sed_2d'])


def test_filters_bit_exact(golden):
    g = golden('ref_filters.npz')
    x = g['x']
    for n in (1, 3, 5, 11, 41, 101):
        out = pp.medfilt(x.copy(), n)
        assert out.dtype == g[f'medfilt_{n}'].dtype
        np.testing.assert_array_equal(out, g[f'medfilt_{n}'])
    np.testing.assert_array_equal(pp.medfilt(x.copy(), 3, axis=1), g['medfilt_axis1_3'])
    for n in (2, 4, 10, 20):
        out = pp.stepfilt(x.copy(), n)
        assert out.dtype == np.float64
        np.testing.assert_array_equal(out, g[f'stepfilt_{n}'])
    for n in (0, 2, 6, 20):
        out = pp.boundariesfilt(x.copy(), n)
        assert out.dtype == g[f'boundariesfilt_{n}'].dtype
        np.testing.assert_array_equal(out, g[f'boundariesfilt_{n}'])
    np.testing.assert_array_equal(pp.filtering(x.copy(), pp.medfilt, np.array(5)), g['filtering_med_0d'])
    np.testing.assert_array_equal(pp.filtering(x.copy(), pp.medfilt, g['len_1d']), g['filtering_med_1d'])
    np.testing.assert_array_equal(pp.filtering(x.copy(), pp.medfilt, g['len_2d']), g['filtering_med_2d'])
    np.testing.assert_array_equal(pp.filtering(x.copy(), pp.medfilt, g['len_2d_bcast']),
                                  g['filtering_med_2d_bcast'])
    out = pp.filtering(x.copy(), pp.boundariesfilt, g['steplen_1d'])
    assert out.dtype == np.float32
    np.testing.assert_array_equal(out, g['filtering_bnd_1d'])


def _flat(out, ids):
    return np.concatenate([out[a].reshape(-1) for a in ids])


def test_ensemble_postprocess_bit_exact(golden):
    g = golden('ref_inference.npz')
    scores, seq_len, ids = g['scores'], g['seq_len'], g['ids']
    flat_ids = [a for batch in ids for a in batch]
    tags = dict(zip(flat_ids, g['tags']))

    def run(**kw):
        out = {}
        for j in range(scores.shape[1]):
            out.update(pp.postprocess([scores[i, j] for i in range(scores.shape[0])], seq_len[j],
                                      list(ids[j]), **kw))
        return out
    o = run(medfilt_length=5)
    assert o[flat_ids[0]].dtype.name == str(g['sed_med_scalar_dtype'])
    np.testing.assert_array_equal(_flat(o, flat_ids), g['sed_med_scalar'])
    o = run(medfilt_length=g['medfilt_2d'], apply_mask=g['apply_mask_2d'], masks=tags)
    assert tuple(o[flat_ids[0]].shape) == tuple(g['sed_med_2d_masked_shape0'])
    assert o[flat_ids[0]].dtype.name == str(g['sed_med_2d_masked_dtype'])
    np.testing.assert_array_equal(_flat(o, flat_ids), g['sed_med_2d_masked'])
    o = run(stepfilt_length=np.array([0, 2, 4, 10, 6]), apply_mask=True, masks=tags)
    assert o[flat_ids[0]].dtype.name == str(g['bnd_step_dtype'])
    np.testing.assert_array_equal(_flat(o, flat_ids), g['bnd_step'])
    o = run(tagging=True)
    np.testing.assert_array_equal(_flat(o, flat_ids), g['tagging'])


def test_event_extraction_internal():
    s = np.array([[.1, .9], [.6, .9], [.7, .2], [.2, .8], [.9, .1]])
    ts = np.round(np.arange(0, 6) * .02, 6)
    ev = pp.scores_to_event_list(s, ts, .5, ['a', 'b'])
    assert ev == [(0.0, 0.04, 'b'), (0.02, 0.06, 'a'), (0.06, 0.08, 'b'), (0.08, 0.1, 'a')]
    fr = pp.event_frames(s, [.5, .5])
    np.testing.assert_array_equal(fr[0], [[1, 3], [4, 5]])
    np.testing.assert_array_equal(fr[1], [[0, 2], [3, 4]])
