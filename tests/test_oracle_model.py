"""Shape doctests of the reference re-expressed (weak_label/crnn.py:16-35, strong_label/crnn.py:15-43)
plus self-consistency checks of the restated third-party layers."""
import numpy as np
import pytest
import torch

from oracle import frontend as fe
from oracle import models as om
from oracle import nn as onn

TINY = dict(out_channels_2d=[32, 32, 32], pool_sizes_2d=1, kernel_size_2d=3,
            out_channels_1d=[32, 32], kernel_size_1d=3)


def test_fbcrnn_doctest_shapes():
    torch.manual_seed(0)
    m = om.FBCRNN.build(num_events=10, number_of_filters=80, stft_size=512, hidden_size=64,
                        num_layers=1, net=TINY)
    np.random.seed(3)
    inputs = {'stft': torch.tensor(np.random.randn(4, 1, 15, 257, 2), dtype=torch.float32),
              'seq_len': [15, 14, 13, 12], 'weak_targets': torch.zeros((4, 10)),
              'boundary_targets': torch.zeros((4, 10, 15))}
    out = m({**inputs})
    assert out[0].shape == (4, 10, 15) and out[1].shape == (4, 10, 15)
    review = m.review(inputs, out)
    assert torch.isfinite(review['loss'])
    review['loss'].backward()


def test_bicrnn_doctest_shapes():
    torch.manual_seed(0)
    m = om.BiCRNN.build(num_events=10, number_of_filters=80, stft_size=512, hidden_size=64,
                        num_layers=1, net=TINY, tag_conditioning=True)
    inputs = {'stft': torch.randn(4, 1, 5, 257, 2), 'seq_len': [5, 4, 3, 2],
              'weak_targets': torch.zeros(4, 10), 'strong_targets': torch.zeros(4, 10, 5),
              'tag_condition': torch.zeros(4, 10)}
    out = m({**inputs})
    assert out[0].shape == (4, 10, 5)
    assert torch.isfinite(m.review(inputs, out)['loss'])


def test_stft_frames_and_window():
    assert fe.num_frames(160000) == 500
    wav = torch.randn(2, 16000, dtype=torch.float64)
    s = fe.stft(wav)
    assert s.shape == (2, 1, 50, 513, 2)
    from scipy.signal.windows import blackman
    np.testing.assert_allclose(fe.blackman_periodic(960), blackman(961)[:-1], atol=1e-15)
    # frame 3 by hand
    x = torch.zeros(16000 + 640, dtype=torch.float64)
    x[320:320 + 16000] = wav[0]
    fr = x[3 * 320:3 * 320 + 960] * torch.from_numpy(fe.blackman_periodic())
    ref = torch.fft.rfft(fr, n=1024)
    np.testing.assert_allclose(s[0, 0, 3, :, 0].numpy(), ref.real.float().numpy(), rtol=0, atol=1e-5)


def test_fbanks_properties():
    fb = fe.get_fbanks()
    assert fb.shape == (128, 513) and fb.dtype == np.float32
    np.testing.assert_allclose(fb.sum(-1), 1., atol=1e-6)
    assert ((fb > 0).sum(0) <= 2).all()          # at most two overlapping triangles per bin


def test_normalization_matches_batchnorm_when_full_length():
    torch.manual_seed(0)
    x = torch.randn(3, 5, 7, 11)
    n = onn.Normalization(5, eps=1e-3).train()
    bn = torch.nn.BatchNorm2d(5, eps=1e-3).train()
    np.testing.assert_allclose(n(x).detach().numpy(), bn(x).detach().numpy(), atol=1e-5)


def test_gru_wrapper_reverse_is_time_flip_for_full_length():
    torch.manual_seed(0)
    g = onn.GRU(6, 8, 2, reverse=True)
    x = torch.randn(2, 6, 9)
    y, _ = g(x, np.array([9, 9]))
    y2, _ = g.rnn(x.transpose(1, 2).flip(1))
    np.testing.assert_allclose(y.detach().numpy(), y2.flip(1).transpose(1, 2).detach().numpy(), atol=1e-6)
    # ragged: outputs past seq_len are zero
    y, _ = g(x, np.array([9, 5]))
    assert (y[1, :, 5:] == 0).all()


def test_load_init_checkpoint_surgery():
    """Reference training.py:327-342: CNN + GRUs loaded completely, output nets without their last layer; a checkpoint
    from a 10-class model initialises a 7-class one; a checkpoint that lacks CNN tensors is refused."""
    from pb_sed_amd.models import weak_label
    from pb_sed_amd.trainer import load_init_checkpoint
    torch.manual_seed(1)
    src = weak_label.CRNN.build(num_events=10)
    torch.manual_seed(2)
    dst = weak_label.CRNN.build(num_events=7)
    before = {k: v.clone() for k, v in dst.state_dict().items()}
    ckpt = {k: v.clone() for k, v in src.state_dict().items()}
    ckpt['cnn.cnn_2d.convs.1.norm.num_tracked_values'] = torch.zeros(())         # a buffer the build does not keep
    loaded = load_init_checkpoint(dst, ckpt)
    after = dst.state_dict()
    for k in after:
        head_last = k.startswith(('rnn_fwd.output_net.convs.1.', 'rnn_bwd.output_net.convs.1.'))
        if k.startswith(('cnn.', 'rnn_fwd.', 'rnn_bwd.')) and not head_last:
            assert k in loaded and torch.equal(after[k], ckpt[k]), k
        else:
            assert k not in loaded and torch.equal(after[k], before[k]), k
    assert after['rnn_fwd.output_net.convs.1.conv.weight'].shape[0] == 7
    broken = {k: v for k, v in ckpt.items() if k != 'cnn.cnn_1d.convs.2.conv.weight'}
    with pytest.raises(KeyError):
        load_init_checkpoint(dst, broken)
