"""The WHOLE product path on the CPU: pb_sed_amd's models, engine and trainer - unchanged - over the C-ABI, with every kernel of
``csrc/`` executed by the emulator of tests/emu (tests/emu/cpu_device.py: HIP threads as fibers, MFMA / DPP / barriers as rendezvous
points, the persistent scans' workgroups as OS threads).  These are the GPU parity tests of tests/test_gpu_model.py in miniature
(same oracle, same tolerances), run where no GPU is: they pin the arithmetic of the device code and the launch sequence of the host
side, not timing and not the GPU's memory model.

Reference path: pb_sed/models/weak_label/crnn.py:80-300 (FBCRNN forward / review), pb_sed/models/strong_label/crnn.py:60-136.
"""
import copy
import os

import numpy as np
import pytest
import torch

from tests.emu import cpu_device
from tests.test_gpu_model import TINY, _copy_weights, rel_close, synth_batch

pytestmark = pytest.mark.skipif(not os.path.exists(cpu_device.CLANG), reason='needs the ROCm clang++ (ext_vector_type, __bf16)')


@pytest.fixture(scope='module')
def library(tmp_path_factory):
    return cpu_device.EmulatedLibrary(tmp_path_factory.mktemp('emu_whole'))


@pytest.fixture
def device(monkeypatch, library):
    with cpu_device.emulated_device(monkeypatch, library):
        library.calls.clear()
        yield library


def _same_gradients(grads, grads0):
    """Bit for bit - but: SIDE_WGRAD groups the deferred jobs into other launches, and under a SHUFFLED fiber schedule
    (tools/emu_schedules.sh, seeded per block index) the fp32 atomics of the bias gradients then add in another order, as they do
    from run to run on the GPU."""
    shuffled = int(os.environ.get('HIPEMU_SCHEDULE', '0')) >= 2
    for name, g in grads0.items():
        if shuffled:
            torch.testing.assert_close(grads[name], g, rtol=1e-5, atol=1e-7 * g.abs().max().item(), msg=name)
        else:
            assert torch.equal(grads[name], g), name


def _nontrivial_norms(ref):
    with torch.no_grad():
        for name, p in ref.named_parameters():
            if name.endswith('gamma'):
                p.uniform_(.7, 1.3)
            elif name.endswith('beta') or name.endswith('conv.bias'):
                p.normal_(0, .1)
        ref.feature_extractor.mean.fill_(-7.)
        ref.feature_extractor.inv_std.fill_(.4)


def test_fbcrnn_train_step_on_the_cpu_follows_the_oracle(monkeypatch, device):
    """tests/test_gpu_model.py::test_fbcrnn_train_step_parity[tiny_ragged] at B = 3, 0.5 s clips, 64 mel bands: features, both directions' scores,
    the loss, the review summary, every parameter gradient (against the float64 oracle, bar = the fp32 oracle's own error) and the
    running statistics."""
    from oracle import frontend as ofe, models as om
    from pb_sed_amd.models import weak_label
    torch.manual_seed(0)
    kw = dict(num_events=10, number_of_filters=64, hidden_size=64, num_layers=2, net=dict(TINY))
    ref = om.FBCRNN.build(**kw)
    _nontrivial_norms(ref)
    model = weak_label.CRNN.build(**kw)
    _copy_weights(model, ref)
    state0 = copy.deepcopy(model.state_dict())
    wav, seq, weak, bnd, t = synth_batch(3, 8000, 10)
    ref.train()
    inputs_ref = {'stft': ofe.stft(wav), 'seq_len': seq.tolist(), 'weak_targets': weak, 'boundary_targets': bnd}
    out_ref = ref(inputs_ref)
    rev_ref = ref.review(inputs_ref, out_ref)
    rev_ref['loss'].backward()
    ref64 = copy.deepcopy(ref).double()
    for m_ in ref64.modules():
        if hasattr(m_, 'running_mean'):
            m_.running_mean.zero_(), m_.running_power.fill_(1.)
    in64 = {'stft': ofe.stft(wav).double(), 'seq_len': seq.tolist(), 'weak_targets': weak.double(), 'boundary_targets': bnd.double()}
    ref64.review(in64, ref64(in64))['loss'].backward()

    model.train()
    inputs = {'audio_data': wav, 'seq_len': seq.tolist(), 'weak_targets': weak, 'boundary_targets': bnd}
    model.flat_parameters()[1].zero_()
    out = model(dict(inputs))
    rev = model.review(inputs, out)
    rev['loss'].backward()

    rel_close(out[3], out_ref[3], 1e-4, 'features')
    assert (out[0] - out_ref[0]).abs().max() < 1e-4 and (out[1] - out_ref[1]).abs().max() < 1e-4
    assert rev['loss'].item() == pytest.approx(rev_ref['loss'].item(), rel=1e-4)
    np.testing.assert_allclose(rev['buffers']['y_weak'], rev_ref['buffers']['y_weak'], atol=1e-4)
    np.testing.assert_array_equal(rev['buffers']['targets_weak'], np.asarray(rev_ref['buffers']['targets_weak']))
    refp, refp32 = dict(ref64.named_parameters()), dict(ref.named_parameters())
    bad = []
    for name, p in model.named_parameters():
        g64 = refp[name].grad
        err32 = (refp32[name].grad.double() - g64).abs().max().item() / (g64.abs().max().item() + 1e-12)
        try:
            rel_close(p.grad, g64, max(2e-3, 3 * err32), name)
        except AssertionError as e:
            bad.append(str(e))
    assert not bad, '\n'.join(bad)
    refb = dict(ref.named_buffers())
    for name, buf in model.named_buffers():
        if 'running' in name:
            rel_close(buf, refb[name], 1e-4, name)
    # the launches of a training step really went through the emulated kernels
    ran = set(device.calls)
    assert {'pbsed_logmel_fwd', 'pbsed_gru_stack_fwd_granule', 'pbsed_gru_stack_bwd_granule', 'pbsed_fbcrnn_loss', 'pbsed_conv_bwd_weight',
            'pbsed_gru_wgrad_multi', 'pbsed_tm_gemm'} <= ran, ' '.join(sorted(ran))
    # the same step with engine.SIDE_WGRAD (the heads' and the upper GRU layers' weight gradients deferred beside the BPTT scans of
    # both directions' stacks - engine._stack_rnn_backward, the headline configuration's path): the serial order's gradients
    from pb_sed_amd import engine
    grads0 = {n: p.grad.clone() for n, p in model.named_parameters()}
    monkeypatch.setattr(engine, 'SIDE_WGRAD', True)
    monkeypatch.setattr(engine, '_has_streams', lambda t: True)
    deferred = []
    run_beside = engine._DeferredLaunches.run_beside
    monkeypatch.setattr(engine._DeferredLaunches, 'run_beside', lambda self, *a, **k: (deferred.append(len(self)), run_beside(self, *a, **k))[1])
    model2 = weak_label.CRNN.build(**kw)           # (a training step moves the feature normalisation's statistics: a fresh copy)
    model2.load_state_dict(state0)
    model2.train()
    model2.flat_parameters()[1].zero_()
    out2 = model2(dict(inputs))
    model2.review(inputs, out2)['loss'].backward()
    assert sum(deferred) > 0, 'nothing was deferred'
    assert torch.equal(out2[0], out[0]) and torch.equal(out2[1], out[1])
    _same_gradients({n: p.grad for n, p in model2.named_parameters()}, grads0)


def _bicrnn_step(seed=2):
    """One tag-conditioned BiCRNN training step in the default fp32-class mode on 16-byte aligned rows: engine._prec then picks the
    bf16x3 kernels of the bench configurations - few-channel MFMA (conv_s16.hip), Winograd-domain 3x3 (conv_winox3.hip), producer /
    consumer 1-D (conv1d_pc.hip), pipelined bf16-MFMA with three-way splits (conv_bf16.hip); returns (scores, loss, gradients)."""
    from pb_sed_amd.models import strong_label
    from oracle import models as om
    torch.manual_seed(seed)
    net = dict(out_channels_2d=[16, 16, 32, 32], pool_sizes_2d=[1, (2, 1), 1, (2, 1)], kernel_size_2d=3,
               out_channels_1d=[128, 128, 64], kernel_size_1d=[1, 3, 1])
    kw = dict(num_events=10, number_of_filters=32, hidden_size=64, num_layers=2, net=net, tag_conditioning=True)
    ref = om.BiCRNN.build(**kw)
    _nontrivial_norms(ref)
    model = strong_label.CRNN.build(**kw)
    _copy_weights(model, ref)
    wav, seq, weak, strong, t = synth_batch(2, 8800, 10, seed=5)          # 28 frames: 16-byte aligned rows, the bf16x3 kernels' form
    tag = (weak > .99).float()
    model.train()
    inp = {'audio_data': wav, 'seq_len': seq.tolist(), 'weak_targets': weak, 'strong_targets': strong, 'tag_condition': tag}
    model.flat_parameters()[1].zero_()
    out = model(dict(inp))
    loss = model.review(inp, out)['loss']
    loss.backward()
    grads = {n: p.grad.clone() for n, p in model.named_parameters()}
    return ref, dict(inp, seq=seq, wav=wav), out[0].detach().clone(), loss.item(), grads


@pytest.fixture(scope='module')
def bicrnn_on_the_tree(library):
    mp = pytest.MonkeyPatch()
    try:
        with cpu_device.emulated_device(mp, library):
            library.calls.clear()
            res = _bicrnn_step()
            return res + (set(library.calls),)
    finally:
        mp.undo()


def test_bicrnn_train_step_through_the_bf16x3_kernels_on_the_cpu_follows_the_oracle(bicrnn_on_the_tree):
    """tests/test_gpu_model.py::test_bicrnn_train_step_parity in miniature."""
    from oracle import frontend as ofe
    ref, inp, y, loss, grads, ran = bicrnn_on_the_tree
    ref.train()
    inp_ref = {'stft': ofe.stft(inp['wav']), 'seq_len': inp['seq'].tolist(), 'weak_targets': inp['weak_targets'],
               'strong_targets': inp['strong_targets'], 'tag_condition': inp['tag_condition']}
    out_ref = ref(inp_ref)
    loss_ref = ref.review(inp_ref, out_ref)['loss']
    loss_ref.backward()
    assert (y - out_ref[0]).abs().max() < 1e-4
    assert loss == pytest.approx(loss_ref.item(), rel=1e-4)
    bad = []
    for name, p in ref.named_parameters():
        try:
            rel_close(grads[name], p.grad, 2e-3, name)
        except AssertionError as e:
            bad.append(str(e))
    assert not bad, '\n'.join(bad)
    assert {'pbsed_conv_fwd_s16', 'pbsed_conv_bwd_data_s16', 'pbsed_conv_fwd_winox3', 'pbsed_conv_bwd_data_winox3', 'pbsed_conv1d_fwd_x3',
            'pbsed_conv1d_bwd_data_x3', 'pbsed_bicrnn_loss'} <= ran, ' '.join(sorted(ran))


def test_weight_gradients_beside_the_scans_change_nothing_on_the_cpu(monkeypatch, device, bicrnn_on_the_tree):
    """engine.SIDE_WGRAD (the heads' and the upper GRU layer's weight gradients deferred to a second stream beside
    the next persistent scan - parked behind PBSED_SIDE_WGRAD=1 until it is measured): the deferred launches write buffers nothing in
    between reads, so the step's gradients must be those of the serial order BIT FOR BIT.  (Streams run in program order here: this
    pins the bookkeeping - what is deferred, that everything deferred runs, joins before the consumers - not the overlap.)"""
    from pb_sed_amd import engine
    monkeypatch.setattr(engine, 'SIDE_WGRAD', True)
    monkeypatch.setattr(engine, '_has_streams', lambda t: True)
    deferred = []
    run_beside = engine._DeferredLaunches.run_beside
    monkeypatch.setattr(engine._DeferredLaunches, 'run_beside', lambda self, *a, **k: (deferred.append(len(self)), run_beside(self, *a, **k))[1])
    _, _, y, loss, grads = _bicrnn_step()
    assert sum(deferred) > 0, 'nothing was deferred'
    _, _, y0, loss0, grads0, _ = bicrnn_on_the_tree
    assert torch.equal(y, y0) and loss == loss0
    _same_gradients(grads, grads0)


def test_parked_kernel_patches_change_no_bit_of_a_training_step(monkeypatch, tmp_path, bicrnn_on_the_tree):
    """The whole patch stack of tools/micro/attic (loads re-ordered around the in-order vmcnt, awaiting a GPU
    measurement) built as a library of its own: scores, loss and every gradient of the step identical to the tree's."""
    import re
    import shutil
    import subprocess
    work = tmp_path / 'patched'
    (work / 'pb_sed_amd').mkdir(parents=True)
    shutil.copytree(os.path.join(cpu_device.ROOT, 'pb_sed_amd', 'csrc'), work / 'pb_sed_amd' / 'csrc', ignore=shutil.ignore_patterns('build'))
    script = open(os.path.join(cpu_device.ROOT, 'tools', 'build_variants.sh')).read()
    patches = re.findall(r'attic/(\w+\.patch)\)', script)
    assert patches
    for p in patches:
        subprocess.run(['patch', '-s', '-p1', '-i', os.path.join(cpu_device.ROOT, 'tools', 'micro', 'attic', p)], cwd=work, check=True)
    patched = cpu_device.EmulatedLibrary(tmp_path, csrc=str(work / 'pb_sed_amd' / 'csrc'))
    with cpu_device.emulated_device(monkeypatch, patched):
        _, _, y, loss, grads = _bicrnn_step()
    _, _, y0, loss0, grads0, _ = bicrnn_on_the_tree
    assert torch.equal(y, y0) and loss == loss0
    for name, g in grads0.items():
        assert torch.equal(grads[name], g), name


def test_inference_driver_on_the_cpu_follows_the_oracle(device):
    """tests/test_gpu_postproc.py::test_inference_driver_end_to_end_vs_oracle in miniature (BASELINE config 5's shape: taggers ->
    tags -> tag-conditioned detectors sharing their scan launches -> ensemble mean, per-class median filters, masking, event
    lists): pb_sed/models/base/inference.py:121-283 restated in oracle/postproc.py."""
    from oracle import frontend as ofe, models as om, postproc as pp
    from pb_sed_amd import inference as inf
    from pb_sed_amd.models import strong_label, weak_label
    torch.manual_seed(3)
    kw = dict(num_events=10, number_of_filters=64, hidden_size=64, num_layers=2, net=TINY)
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
    wav, seq, *_ = synth_batch(3, 8800, 10, seed=7)
    ids = [f'a{i}' for i in range(3)]
    batch = {'audio_data': wav, 'seq_len': seq.tolist(), 'example_id': ids}
    stft = ofe.stft(wav)
    tag_scores = inf.tagging([m for _, m in pairs_w], [dict(batch)], 'cpu')
    with torch.no_grad():
        ref = pp.postprocess([r.tagging({'stft': stft, 'seq_len': seq.tolist()})[0].numpy() for r, _ in pairs_w],
                             np.ones(3, int), ids, tagging=True)
    for a in ids:
        np.testing.assert_allclose(tag_scores[a], ref[a], atol=1e-4)
    tags = {a: (ref[a][0] > .5).astype(np.float32) for a in ids}
    tag_cond = torch.tensor(np.stack([tags[a] for a in ids]))
    ml = np.array([[1, 3, 5, 7, 9, 1, 3, 5, 7, 9], [3] * 10])
    sed = inf.sound_event_detection([m for _, m in pairs_s], [dict(batch, tag_condition=tag_cond)], 'cpu',
                                    medfilt_length=ml, apply_mask=True, masks=tags)
    with torch.no_grad():
        ref = pp.postprocess([r.sound_event_detection({'stft': stft, 'seq_len': seq.tolist(), 'tag_condition': tag_cond})[0].numpy()
                              for r, _ in pairs_s], seq, ids, medfilt_length=ml, apply_mask=True, masks=tags)
    for a in ids:
        assert sed[a].shape == ref[a].shape
        np.testing.assert_allclose(sed[a], ref[a], atol=1e-4)
    # thresholds -> event lists (onset, offset, class), bit-exact against the oracle on the build's own scores
    classes = [f'c{i}' for i in range(10)]
    ts = np.round(np.arange(0, 100) * .02, 6)
    thr = np.full(10, .5, np.float32)
    scores = {a: np.ascontiguousarray(sed[a][0]) for a in ids}              # [T, K]
    got = inf.scores_to_event_list(scores, thr, classes, ts, device='cpu')
    for a in ids:
        assert got[a] == pp.scores_to_event_list(scores[a], ts, thr, classes), a
    ran = set(device.calls)
    assert {'pbsed_ensemble_mean_mask', 'pbsed_medfilt', 'pbsed_event_frames', 'pbsed_gru_stack_fwd_granule'} <= ran, ' '.join(sorted(ran))


def _dp_batch(b):
    g = torch.Generator().manual_seed(7)
    weak = (torch.rand(b, 10, generator=g) < .3).float()
    weak[:, 0] = 1
    bnd = torch.zeros(b, 10, 28)
    bnd[:, 0, 5:15] = 1
    return {'audio_data': torch.randn(b, 8800, generator=g), 'seq_len': [28] * b, 'weak_targets': weak, 'boundary_targets': bnd}


def _dp_model():
    """Statistics independent of the batch (frozen norm / feature statistics): clips become independent, so the average of the
    ranks' gradients IS the single-rank gradient (tests/test_gpu_dp.py::test_two_ranks_on_one_gpu[True])."""
    from pb_sed_amd.models import weak_label
    from pb_sed_amd.modules import Normalization
    torch.manual_seed(0)
    model = weak_label.CRNN.build(num_events=10, number_of_filters=32, hidden_size=64, num_layers=2, net=dict(TINY))
    model.feature_extractor.freeze_stats = True
    g = torch.Generator().manual_seed(1)
    for m in model.modules():
        if isinstance(m, Normalization):
            m.freeze_stats = True
            with torch.no_grad():
                m.running_mean.copy_(torch.randn(m.running_mean.shape, generator=g) * .1)
                m.running_power.copy_(torch.rand(m.running_power.shape, generator=g) + .8)
    return model


def _dp_worker(rank, world, port, units_dir, out_dir):
    import sys
    sys.path.insert(0, cpu_device.ROOT)
    import torch.distributed as dist
    os.environ.update(MASTER_ADDR='127.0.0.1', MASTER_PORT=str(port))
    dist.init_process_group('gloo', rank=rank, world_size=world)
    mp = pytest.MonkeyPatch()
    with cpu_device.emulated_device(mp, cpu_device.EmulatedLibrary(units_dir, reuse=True)):
        from pb_sed_amd.trainer import Trainer, shard_batch
        trainer = Trainer(_dp_model(), lr=1e-3, gradient_clipping=5., allreduce='torch')
        rev = trainer.step(shard_batch(_dp_batch(4), rank, world))
        out = {'grad': (trainer.flat_grad / world).clone(), 'param': trainer.flat_param.clone(), 'loss': float(rev['loss'].item())}
        # the scans' error words are agreed on over the ranks (Trainer._share_flags; on a GPU always, here switched on by hand):
        # (a) nobody flagged -> the step is the step it was; (b) a word raised on rank 1 ONLY (bit 1: a workgroup of an XCD-local ring
        # on the wrong XCD) -> BOTH ranks skip the update on the device and BOTH raise, naming the cause
        from pb_sed_amd import ops
        trainer._flags_on = True
        trainer.step(shard_batch(_dp_batch(4), rank, world))
        trainer.finish()
        out['param_after_clean_step'] = trainer.flat_param.clone()
        if rank == 1:
            ops.gru_flags('cpu')[0][7] = 2
        trainer.step(shard_batch(_dp_batch(4), rank, world))
        out['param_after_flagged_step'] = trainer.flat_param.clone()
        out['agreed'] = trainer._flags_agreed.clone()
        try:
            trainer.finish()
            out['raised'] = ''
        except RuntimeError as ex:
            out['raised'] = str(ex)
        torch.save(out, os.path.join(out_dir, f'rank{rank}.pt'))
    mp.undo()
    dist.destroy_process_group()


def test_two_data_parallel_ranks_on_the_cpu_reproduce_the_single_rank_step(monkeypatch, tmp_path, library):
    """SURVEY.md 8(e) with the real kernels and no GPU: two processes (gloo), each with the emulated device and its shard of the
    batch, one Trainer step (bucketed all-reduce of the flat gradient, averaged Adam through pbsed_adam_step): the ranks end
    bit-identical, and - statistics frozen - their averaged gradient is the gradient of one rank on the whole batch."""
    import socket
    import torch.multiprocessing as tmp_mp
    with socket.socket() as s:
        s.bind(('127.0.0.1', 0))
        port = s.getsockname()[1]
    tmp_mp.spawn(_dp_worker, args=(2, port, library.outdir, str(tmp_path)), nprocs=2, join=True)
    r0, r1 = (torch.load(tmp_path / f'rank{r}.pt') for r in range(2))
    assert np.isfinite(r0['loss']) and np.isfinite(r1['loss'])
    assert torch.equal(r0['grad'], r1['grad']) and torch.equal(r0['param'], r1['param']), 'ranks diverged'
    for r in (r0, r1):
        assert not torch.equal(r['param_after_clean_step'], r['param'])                          # a clean step with the words shared: updated
        assert torch.equal(r['param_after_flagged_step'], r['param_after_clean_step'])           # rank 1's word: no update on EITHER rank
        assert int(r['agreed'][7]) == 2 and int(r['agreed'].sum()) == 2
        assert 'XCD' in r['raised'], r['raised']                                                 # both raise, with the cause of bit 1
    assert torch.equal(r0['param_after_clean_step'], r1['param_after_clean_step'])
    with cpu_device.emulated_device(monkeypatch, library):
        from pb_sed_amd.trainer import Trainer
        batch = _dp_batch(4)
        w = ((batch['weak_targets'] < .01) | (batch['weak_targets'] > .99)).float().sum(-1)
        assert w[:2].sum() == w[2:].sum()          # equal per-rank loss-weight sums: average of the ranks' gradients = global gradient
        trainer = Trainer(_dp_model(), lr=1e-3, gradient_clipping=5.)
        trainer.step(batch)
        single, param = trainer.flat_grad.clone(), trainer.flat_param.clone()
    rel = ((r0['grad'] - single).norm() / single.norm()).item()
    assert rel < 2e-5, rel
    assert (r0['param'] - param).abs().max().item() < 1e-5


def test_tuning_drivers_on_the_emulated_device_reproduce_the_reference_leaderboards(device, tmp_path):
    """pb_sed_amd.tuning (median-filter / step-filter / tag-masking leaderboard search, pb_sed/models/base/tuning.py) with its filters
    running as the HIP kernels of csrc/postproc.hip on the emulated device: the leaderboards the REFERENCE's drivers produced for the
    same score DataFrames and metric functions (tests/golden/ref_tuning.npz) - metric values, tuned hyper-parameters per class and
    the winning score column of every clip, bit for bit; clips of two lengths (one launch per length).  Then the JSON hand-off
    (`<stage>_hyper_params_<metric>.json`) and the TSV reader behind boundaries_from_events."""
    import json
    from tests import tuning_case
    assert tuning_case.replay('cpu') == 3 * 2 * 9
    assert {'pbsed_medfilt', 'pbsed_boundariesfilt'} <= set(device.calls)
    import pandas as pd
    from pb_sed_amd import tuning
    ts = np.round(np.arange(11) * .02, 6)
    x = np.random.RandomState(0).rand(10, 2).astype(np.float32).astype(np.float64)
    scores = {'c0': pd.DataFrame(np.concatenate((ts[:-1, None], ts[1:, None], x), 1), columns=['onset', 'offset', 'A', 'B'])}
    metric = {'peak': lambda s: ({'A': float(s['c0']['A'].max()), 'B': float(s['c0']['B'].max()), 'macro_average': 0.}, {'A': {'threshold': .5}})}
    board = tuning.tune_tagging(scores, [1, 3], metric, storage_dir=tmp_path, device='cpu', verbose=False)
    stored = json.load(open(tmp_path / 'tagging_hyper_params_peak.json'))
    assert stored['A'] == {'medfilt_length': board['peak'][1]['A']['medfilt_length'], 'threshold': .5, 'peak': board['peak'][0]['A']}
    with pytest.raises(ValueError):
        tuning.tune_tagging({'c0': scores['c0'] * (1 / 3)}, [3], metric, device='cpu', verbose=False)      # not float32 values: refused, not rounded
    tsv = tmp_path / 'gt.tsv'
    tsv.write_text('filename\tonset\toffset\tevent_label\na.wav\t0.5\t1.0\tDog\na.wav\t2.0\t2.5\tDog\nb.wav\t\t\t\n')
    assert tuning.boundaries_from_events(str(tsv)) == {'a': [(0.5, 2.5, 'Dog')], 'b': []}


def test_launch_cu_budget_cuts_the_weight_gradient_grid_not_its_result(library):
    """pbsed_set_launch_cus (the CU budget SIDE_WGRAD sets around the launches it puts beside a persistent scan): with a budget
    below the device's CU count the weight-gradient launchers cut their persistent grids for fewer CUs - other shares of the
    (clip, row, column) units per workgroup, other partial sums - and the gradient is the same to fp32 summation order; the
    previous budget is handed back, 0 restores the whole device."""
    import ctypes as C
    rng = np.random.RandomState(3)
    b, cin, cout, f, t = 4, 32, 32, 4, 64
    x = rng.randn(b, cin, f, t).astype(np.float32)
    g = rng.randn(b, cout, f, t).astype(np.float32)
    seq = np.array([64, 60, 50, 33], np.int32)
    P = lambda a: a.ctypes.data_as(C.c_void_p)
    out = {}
    assert library.pbsed_set_launch_cus(0) == 0
    for budget in (0, 2):
        assert library.pbsed_set_launch_cus(budget) == 0
        dw, db = np.zeros((cout, cin, 3, 3), np.float32), np.zeros(cout, np.float32)
        rc = library.pbsed_conv_bwd_weight(P(x), None, None, 0, P(seq), P(g), None, P(dw), P(db), b, cin, cout, f, t, 3, 3, None)
        assert rc == 0, library.pbsed_last_error()
        assert library.pbsed_set_launch_cus(0) == budget
        out[budget] = (dw, db)
    xp = np.pad(x.astype(np.float64), ((0, 0), (0, 0), (1, 1), (1, 1)))
    ref = np.zeros((cout, cin, 3, 3))
    for i in range(3):
        for j in range(3):
            ref[:, :, i, j] = np.einsum('boft,bcft->oc', g.astype(np.float64), xp[:, :, i:i + f, j:j + t])
    for budget in (0, 2):
        assert np.abs(out[budget][0] - ref).max() < 3e-5 * np.abs(ref).max(), budget
    assert np.abs(out[2][0] - out[0][0]).max() < 2e-6 * np.abs(ref).max()
    assert np.abs(out[2][1] - out[0][1]).max() < 2e-6 * np.abs(out[0][1]).max()
