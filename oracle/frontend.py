"""Oracle: STFT -> |X|^2 -> mel -> log -> normalise -> clamp (CPU, stock torch/numpy).

TEST INFRASTRUCTURE ONLY (see oracle/__init__.py).  Third-party semantics restated,
"parity unpinned":
* STFT: padertorch ``STFT(shift=320, window_length=960, size=1024, fading='half',
  pad=True)`` configured at reference ``pb_sed/data_preparation/provider.py:315-323`` and
  called at ``pb_sed/data_preparation/transform.py:53`` (-> paderbox
  ``transform.module_stft.stft``, pinned at paderbox@809b272, README.md:41).
* mel/log/normalise: padertorch ``NormalizedLogMelExtractor`` called at reference
  ``pb_sed/models/weak_label/crnn.py:86-90`` with the config of
  ``pb_sed/experiments/weak_label_crnn/training.py:190-217`` (-> paderbox
  ``transform.module_fbank.get_fbanks``).
"""
import math

import numpy as np
import torch

SHIFT = 320
WINDOW_LENGTH = 960
FFT_SIZE = 1024
N_BINS = FFT_SIZE // 2 + 1


def blackman_periodic(n=WINDOW_LENGTH):
    """Periodic Blackman == scipy.signal.windows.blackman(n + 1)[:-1] (paderbox stft default)."""
    k = np.arange(n, dtype=np.float64)
    return 0.42 - 0.5 * np.cos(2 * np.pi * k / n) + 0.08 * np.cos(4 * np.pi * k / n)


def num_frames(n_samples, shift=SHIFT, window_length=WINDOW_LENGTH):
    """'half' fading pads (window_length-shift)//2 front and ceil((wl-shift)/2) back, then
    segment_axis(end='pad') -> ceil((n + pad - wl) / shift) + 1 frames."""
    pad = window_length - shift
    return max(int(math.ceil((n_samples + pad - window_length) / shift)) + 1, 1)


def stft(wav, shift=SHIFT, window_length=WINDOW_LENGTH, size=FFT_SIZE, frame_pos=None):
    """wav [B, N] float -> [B, 1, T, size//2+1, 2] float32 (re, im), float64 inside like numpy.
    ``frame_pos`` [B, T] ints: first sample of every frame's window (samples outside the clip are zeros) - time-warped
    framing; None: the base STFT's regular grid."""
    wav = torch.as_tensor(wav, dtype=torch.float64)
    b, n = wav.shape
    if frame_pos is not None:
        pos = torch.as_tensor(np.asarray(frame_pos), dtype=torch.long)
        idx = pos[:, :, None] + torch.arange(window_length)[None, None]          # [B, T, wl]
        ok = (idx >= 0) & (idx < n)
        frames = torch.gather(wav, 1, idx.clamp(0, n - 1).reshape(b, -1)).reshape(idx.shape) * ok
        win = torch.from_numpy(blackman_periodic(window_length))
        spec = torch.fft.rfft(frames * win, n=size, dim=-1)
        return torch.stack([spec.real, spec.imag], dim=-1).to(torch.float32)[:, None]
    pad_front = (window_length - shift) // 2
    pad_back = int(math.ceil((window_length - shift) / 2))
    t = num_frames(n, shift, window_length)
    total = (t - 1) * shift + window_length
    x = torch.zeros(b, max(total, n + pad_front + pad_back), dtype=torch.float64)
    x[:, pad_front:pad_front + n] = wav
    frames = x[:, :total].unfold(1, window_length, shift)  # [B, T, wl]
    win = torch.from_numpy(blackman_periodic(window_length))
    spec = torch.fft.rfft(frames * win, n=size, dim=-1)  # zero-pad at the end to `size`
    out = torch.stack([spec.real, spec.imag], dim=-1).to(torch.float32)
    return out[:, None]


def hz2mel(f):
    return 2595.0 * np.log10(1.0 + np.asarray(f, dtype=np.float64) / 700.0)


def mel2hz(m):
    return 700.0 * (10.0 ** (np.asarray(m, dtype=np.float64) / 2595.0) - 1.0)


def get_fbanks(sample_rate=16000, stft_size=FFT_SIZE, number_of_filters=128,
               lowest_frequency=50.0, highest_frequency=None):
    """Triangular HTK-mel filterbank, each filter normalised to unit sum -> [n_filters, bins] f32."""
    if highest_frequency is None:
        highest_frequency = sample_rate / 2
    mel_points = np.linspace(hz2mel(lowest_frequency), hz2mel(highest_frequency),
                             number_of_filters + 2)
    centers_hz = mel2hz(mel_points)
    frac_bins = centers_hz / sample_rate * stft_size
    k = np.arange(stft_size // 2 + 1, dtype=np.float64)[None]
    centers = frac_bins[1:-1, None]
    onsets = frac_bins[:-2, None]
    offsets = frac_bins[2:, None]
    fb = np.maximum(np.minimum((k - onsets) / (centers - onsets),
                               (offsets - k) / (offsets - centers)), 0.0)
    fb = fb / fb.sum(-1, keepdims=True)
    return fb.astype(np.float32)


def compute_mask(x, seq_len, batch_axis=0, sequence_axis=-1):
    """padertorch.ops.sequence.mask.compute_mask restated: same-shape 0/1 mask t < seq_len[b]."""
    if seq_len is None:
        return torch.ones_like(x)
    seq_len = torch.as_tensor(np.asarray(seq_len), device=x.device)
    t = x.shape[sequence_axis]
    shape = [1] * x.dim()
    shape[sequence_axis] = t
    idx = torch.arange(t, device=x.device).reshape(shape)
    lshape = [1] * x.dim()
    lshape[batch_axis] = x.shape[batch_axis]
    return (idx < seq_len.reshape(lshape)).to(x.dtype).expand_as(x)


class LogMelExtractor(torch.nn.Module):
    """NormalizedLogMelExtractor restated: [B,1,T,513,2] -> [B,1,F,T].

    Eval mode (or ``freeze_stats``) normalises with the explicit buffers ``mean[F]``, ``inv_std[F]``.  Training mode
    follows SURVEY A.3: padertorch ``Normalization(statistics_axis='bt', momentum=None, eps=1e-5)`` - per-mel mean and
    power accumulated cumulatively over every valid (clip, frame) seen so far, updated with the current batch and
    then used for the normalisation (parity unpinned: padertorch is absent).  Augmentation is ``augment`` below.
    """

    def __init__(self, sample_rate=16000, stft_size=FFT_SIZE, number_of_filters=128,
                 lowest_frequency=50.0, highest_frequency=None, eps=1e-18, clamp=6.0, norm_eps=1e-5):
        super().__init__()
        self.stft_size = stft_size
        self.number_of_filters = number_of_filters
        self.eps = eps
        self.clamp = clamp
        fb = get_fbanks(sample_rate, stft_size, number_of_filters, lowest_frequency,
                        highest_frequency)
        self.register_buffer('fbanks', torch.from_numpy(fb))
        self.norm_eps = norm_eps
        self.freeze_stats = False
        self.register_buffer('running_mean', torch.zeros(number_of_filters))
        self.register_buffer('running_power', torch.ones(number_of_filters))
        self.register_buffer('num_tracked_values', torch.zeros(1, dtype=torch.float64))
        self.register_buffer('mean', torch.zeros(number_of_filters))
        self.register_buffer('inv_std', torch.ones(number_of_filters))

    def _track(self, logmel, seq_len):
        m = compute_mask(logmel, seq_len).double()
        lm = logmel.double() * m
        n = m[:, 0, 0].sum()                                      # valid (clip, frame) positions
        s, q = lm.sum((0, 1, 3)), (lm * lm).sum((0, 1, 3))
        n0 = self.num_tracked_values.double()
        mu = (n0 * self.running_mean.double() + s) / (n0 + n)
        pw = (n0 * self.running_power.double() + q) / (n0 + n)
        self.running_mean.copy_(mu), self.running_power.copy_(pw)
        self.num_tracked_values.copy_(n0 + n)
        self.mean.copy_(mu)
        self.inv_std.copy_(1. / torch.sqrt((pw - mu * mu).clamp_min(0.) + self.norm_eps))

    @torch.no_grad()
    def forward(self, x, seq_len=None, targets=None, mel_points=None):
        """``mel_points`` [B, F+2]: per-clip fractional bin positions of warped filters (training-time MelWarping,
        training.py:195-208) - filter m of clip b is the unit-sum triangle over points m, m+1, m+2 (get_fbanks above)."""
        power = (x.to(self.fbanks.dtype) ** 2).sum(-1)          # [B,1,T,bins] (f32; f64 if .double())
        if mel_points is not None:
            p = torch.as_tensor(np.asarray(mel_points), dtype=torch.float64)
            k = torch.arange(power.shape[-1], dtype=torch.float64)[None, None]
            lo, c, hi = p[:, :-2, None], p[:, 1:-1, None], p[:, 2:, None]
            fb = torch.minimum((k - lo) / (c - lo), (hi - k) / (hi - c)).clamp_min(0.)
            fb = (fb / fb.sum(-1, keepdim=True)).to(power.dtype)            # [B,F,bins]
            mel = torch.einsum('bctk,bfk->bctf', power, fb)
        else:
            mel = power @ self.fbanks.T                           # [B,1,T,F]
        logmel = torch.log(mel + self.eps).transpose(-1, -2)     # [B,1,F,T]
        if self.training and not self.freeze_stats:
            self._track(logmel, seq_len)
        y = (logmel - self.mean[:, None]) * self.inv_std[:, None]
        if self.clamp is not None:
            y = torch.clamp(y, -self.clamp, self.clamp)
        y = y * compute_mask(y, seq_len)
        if targets is None:
            return y, seq_len
        return y, seq_len, targets


def augment(y, seq_len, masks, noise=None, noise_scale=None):
    """Training-only augmentation of NormalizedLogMelExtractor (config pb_sed/experiments/weak_label_crnn/
    training.py:209-216; padertorch semantics restated from SURVEY.md A.3 - parity unpinned): y [B,1,F,T] (or
    [B,F,T]) + noise_scale[b] * noise, one time mask [t_on, t_off) and one frequency mask [f_on, f_off) per clip
    (``masks`` [B,4]) set to zero, frames >= seq_len[b] zero."""
    y = y.clone()
    b, f, t = y.shape[0], y.shape[-2], y.shape[-1]
    if noise is not None:
        y = y + torch.as_tensor(noise_scale, dtype=y.dtype).reshape(b, *([1] * (y.dim() - 1))) * noise
    tt, ff = torch.arange(t), torch.arange(f)
    for i in range(b):
        t_on, t_off, f_on, f_off = (int(v) for v in masks[i])
        y[i][..., (tt >= t_on) & (tt < t_off)] = 0
        y[i][..., (ff >= f_on) & (ff < f_off), :] = 0
        y[i][..., tt >= int(seq_len[i])] = 0
    return y
