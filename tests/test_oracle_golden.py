"""Pin the oracle's restatement of reference-OWNED maths against golden vectors produced by the
reference's own source (tests/golden/gen_golden.py -> tests/golden/ref_*.npz)."""
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


# ------------------------------------------------------------------ validation metrics
def test_instance_based_metrics_match_reference(golden):
    """pb_sed_amd.evaluation.instance_based against vectors produced by executing the reference's
    pb_sed/evaluation/instance_based.py (tests/golden/gen_golden.py: gen_instance_based): threshold searches with and without
    ties, rate constraints, beta / bias arguments (honoured for vectors, ignored for matrices exactly as the reference
    does), binary-decision metrics, lwlrap, and the module's own docstring example."""
    from pb_sed_amd.evaluation import instance_based as ib
    g = golden('ref_instance_based.npz')
    targets, decisions = g['targets'], g['decisions']

    def check(name, out):
        out = out if isinstance(out, tuple) else (out,)
        for i, o in enumerate(out):
            ref = g[f'{name}_{i}']
            np.testing.assert_allclose(np.asarray(o, dtype=np.float64), ref, rtol=1e-12, atol=1e-12, err_msg=f'{name}[{i}]')

    for tag, sc in (('c', g['scores']), ('q', g['scores_q'])):
        check(f'best_f_2d_{tag}', ib.get_best_fscore_thresholds(targets, sc))
        check(f'best_f_2d_minp_{tag}', ib.get_best_fscore_thresholds(targets, sc, min_precision=.6))
        check(f'best_f_2d_minr_{tag}', ib.get_best_fscore_thresholds(targets, sc, min_recall=.8))
        check(f'best_f_2d_beta2_{tag}', ib.get_best_fscore_thresholds(targets, sc, beta=2.))
        check(f'best_er_2d_{tag}', ib.get_best_er_thresholds(targets, sc))
        check(f'best_er_2d_maxi_{tag}', ib.get_best_er_thresholds(targets, sc, max_insertion_rate=.1))
        check(f'best_er_2d_maxd_{tag}', ib.get_best_er_thresholds(targets, sc, max_deletion_rate=.2))
        check(f'curve_f_2d_{tag}', ib.fscore_curve(targets, sc))
        check(f'curve_er_2d_{tag}', ib.er_curve(targets, sc))
        for c in (0, 2, 4):
            check(f'best_f_1d_{tag}{c}', ib.get_best_fscore_thresholds(targets[:, c], sc[:, c]))
            check(f'best_f_1d_beta2_bias_{tag}{c}', ib.get_best_fscore_thresholds(
                targets[:, c], sc[:, c], beta=2., tp_bias=1, n_ref_bias=2, n_pos_bias=3))
            check(f'best_er_1d_{tag}{c}', ib.get_best_er_thresholds(targets[:, c], sc[:, c]))
    check('lwlrap', ib.lwlrap(targets, g['scores']))
    for ew in (False, True):
        check(f'fscore_ew{int(ew)}', ib.fscore(targets, decisions, event_wise=ew))
        check(f'fscore_beta2_ew{int(ew)}', ib.fscore(targets, decisions, beta=2., event_wise=ew))
        check(f'error_rate_ew{int(ew)}', ib.error_rate(targets, decisions[1], event_wise=ew))
    t9, s9 = g['t9'], g['s9']
    check('t9_curve_f', ib.fscore_curve(t9, s9))
    check('t9_best_f', ib.get_best_fscore_thresholds(t9, s9))
    check('t9_best_f_minp', ib.get_best_fscore_thresholds(t9, s9, min_precision=.51))
    check('t9_best_er', ib.get_best_er_thresholds(t9, s9))
    # the known answers printed in the reference's docstrings (instance_based.py:283-287, 339-343)
    thr, f, p, r = ib.get_best_fscore_thresholds(t9[:, None], s9[:, None])
    assert np.allclose([thr[0], f[0], p[0], r[0]], [0.15, 2 / 3, 0.5, 1.0])
    assert ib.get_best_er_thresholds(t9, s9) == (np.inf, 1.0, 0.0, 1.0)


def test_summary_metrics_match_reference(golden):
    """SoundEventModel.add_metrics_to_summary of the build against the scalars the reference's own method produced
    (tests/golden/gen_golden.py: gen_summary_metrics): label subsets by index and by name, label-wise keys, and the
    mAP / mAUC branch that is skipped when a class has a single positive."""
    from pb_sed_amd.models import base
    g = golden('ref_summary_metrics.npz')
    labels = [str(x) for x in g['labels']]

    class Model(base.SoundEventModel):
        def tagging(self, inputs, **params): pass
        def boundaries_detection(self, inputs, **params): pass
        def sound_event_detection(self, inputs, **params): pass

    configs = {
        'plain': dict(labelwise_metrics=(), label_mapping=None, test_labels=None),
        'labelwise': dict(labelwise_metrics=('fscore_weak', 'lwlrap_weak', 'ap_weak'), label_mapping=labels, test_labels=None),
        'subset_idx': dict(labelwise_metrics=('error_rate_weak',), label_mapping=None, test_labels=[0, 3, 4]),
        'subset_names': dict(labelwise_metrics=('fscore_weak', 'auc_weak'), label_mapping=labels, test_labels=['dog', 'water']),
    }
    scores = g['scores']
    for name, cfg in configs.items():
        for tname in ('all', 'rare'):
            t = g['targets'] if tname == 'all' else g['targets_rare']
            summary = dict(scalars={}, images={}, buffers={'y_weak': [scores[:40], scores[40:]], 'targets_weak': [t[:40], t[40:]]})
            Model(**cfg).add_metrics_to_summary(summary, 'weak')
            keys = sorted(summary['scalars'])
            assert keys == [str(k) for k in g[f'{name}/{tname}/keys']], (name, tname)
            np.testing.assert_allclose([float(summary['scalars'][k]) for k in keys], g[f'{name}/{tname}/values'],
                                       rtol=1e-12, atol=1e-12, err_msg=f'{name}/{tname}')
            assert not summary['buffers']
    # modify_summary: scalar lists -> means, images -> one normalised column
    m = Model()
    out = m.modify_summary(dict(scalars=dict(loss=[1., 2., 6.]), images=dict(x=torch.arange(24.).reshape(2, 1, 3, 4)), buffers={}))
    assert out['scalars']['loss'] == 3.0 and out['images']['x'].shape == (3, 2 * (3 + 2) + 2, 4 + 4)
    assert out['images']['x'].max().item() == 1.0 and out['images']['x'][:, 2, 2].eq(out['images']['x'][0, 2, 2]).all()
