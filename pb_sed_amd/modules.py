"""Host-side mirror of the padertorch building blocks the reference model is assembled from.

These ``nn.Module`` classes own the parameters (same tree / names as the reference's modules:
``cnn.cnn_2d``, ``cnn.cnn_1d``, ``rnn_fwd.rnn`` with torch-GRU parameter names,
``rnn_fwd.output_net`` - see reference pb_sed/experiments/weak_label_crnn/training.py:329-350) and
describe the layer tables; all arithmetic is done by the HIP kernels driven from
``pb_sed_amd.engine``.  Reference call sites: pb_sed/models/weak_label/crnn.py:86-100.
"""
import math
import re

import numpy as np
import torch
from torch import nn


def hz2mel(f):
    return 2595.0 * np.log10(1.0 + np.asarray(f, dtype=np.float64) / 700.0)


def mel2hz(m):
    return 700.0 * (10.0 ** (np.asarray(m, dtype=np.float64) / 2595.0) - 1.0)


def get_fbanks(sample_rate, stft_size, number_of_filters, lowest_frequency=50., highest_frequency=None):
    """HTK-mel triangular filterbank, unit-sum filters (paderbox get_fbanks semantics)."""
    highest_frequency = sample_rate / 2 if highest_frequency is None else highest_frequency
    pts = mel2hz(np.linspace(hz2mel(lowest_frequency), hz2mel(highest_frequency), number_of_filters + 2))
    frac = pts / sample_rate * stft_size
    k = np.arange(stft_size // 2 + 1, dtype=np.float64)[None]
    c, lo, hi = frac[1:-1, None], frac[:-2, None], frac[2:, None]
    fb = np.maximum(np.minimum((k - lo) / (c - lo), (hi - k) / (hi - c)), 0.)
    return (fb / fb.sum(-1, keepdims=True)).astype(np.float32)


def num_frames(n_samples, shift=320, window_length=960):
    pad = window_length - shift
    return max(int(math.ceil((n_samples + pad - window_length) / shift)) + 1, 1)


class LogTruncatedNormal:
    """paderbox.utils.random_utils.LogTruncatedNormal restated: exp(N(loc, scale) truncated to loc +- truncation)."""

    def __init__(self, loc=0., scale=1., truncation=3., seed=None):
        self.loc, self.scale, self.truncation = loc, scale, truncation
        self._rng = np.random.RandomState(seed)

    def __call__(self, shape=()):
        from scipy.stats import truncnorm
        a = self.truncation / self.scale
        return np.exp(truncnorm(-a, a, loc=self.loc, scale=self.scale).rvs(shape, random_state=self._rng))


class TruncatedExponential:
    """paderbox.utils.random_utils.TruncatedExponential restated: Exp(scale) redrawn / clipped below ``truncation``."""

    def __init__(self, loc=0., scale=1., truncation=3., seed=None):
        self.loc, self.scale, self.truncation = loc, scale, truncation
        self._rng = np.random.RandomState(seed)

    def __call__(self, shape=()):
        from scipy.stats import truncexpon
        return truncexpon(self.truncation / self.scale, loc=self.loc, scale=self.scale).rvs(shape, random_state=self._rng)


class Uniform:
    """paderbox.utils.random_utils.Uniform restated: U(low, high)."""

    def __init__(self, low=0., high=1., seed=None):
        self.low, self.high = low, high
        self._rng = np.random.RandomState(seed)

    def __call__(self, shape=()):
        return self._rng.uniform(self.low, self.high, shape)


class MelWarping:
    """paderbox.transform.module_fbank.MelWarping restated (vocal-tract-length style piecewise-linear warping of the mel
    filters' centre frequencies, one draw per clip; reference config pb_sed/experiments/weak_label_crnn/training.py:195-208):
    f -> alpha f below the boundary frequency b = ratio * f_max * min(alpha, 1) / alpha, and linearly on to f_max above it."""

    def __init__(self, warp_factor_sampling_fn, boundary_frequency_ratio_sampling_fn, highest_frequency):
        self.warp_factor_sampling_fn = warp_factor_sampling_fn
        self.boundary_frequency_ratio_sampling_fn = boundary_frequency_ratio_sampling_fn
        self.highest_frequency = highest_frequency

    def __call__(self, f, n=1):
        """f: [P] frequencies in Hz -> [n, P] warped frequencies."""
        f = np.asarray(f, dtype=np.float64)[None]
        alpha = np.asarray(self.warp_factor_sampling_fn((n,)), dtype=np.float64)[:, None]
        ratio = np.minimum(np.asarray(self.boundary_frequency_ratio_sampling_fn((n,)), dtype=np.float64), 1.)[:, None]
        hi = self.highest_frequency
        b = ratio * hi * np.minimum(alpha, 1.) / alpha
        upper = hi - (hi - alpha * b) / np.maximum(hi - b, 1e-9) * (hi - f)
        return np.where(f <= b, alpha * f, upper)


class NormalizedLogMelExtractor(nn.Module):
    """Front-end description (STFT 1024/960/320 Blackman -> mel -> log -> global norm -> clamp)."""

    def __init__(self, sample_rate=16000, stft_size=1024, number_of_filters=128, lowest_frequency=50.,
                 highest_frequency=None, eps=1e-18, clamp=6.0, shift=320, window_length=960,
                 add_deltas=False, add_delta_deltas=False,
                 n_time_masks=0, max_masked_time_steps=70, max_masked_time_rate=.2,
                 n_frequency_masks=0, max_masked_frequency_bands=20, max_masked_frequency_rate=.2,
                 max_noise_scale=0., frequency_warping_fn=None, augmentation_seed=0, norm_eps=1e-5):
        """The augmentation fields are those of the reference's training config
        (pb_sed/experiments/weak_label_crnn/training.py:194-216); all default to off (as in the reference's class
        defaults), training scripts switch them on.  They act in training mode only."""
        super().__init__()
        if add_deltas or add_delta_deltas:
            raise NotImplementedError('delta features are not used by the reference configurations')
        self.add_deltas, self.add_delta_deltas = False, False
        self.frequency_warping_fn = frequency_warping_fn
        self.lowest_frequency = lowest_frequency
        self.highest_frequency = sample_rate / 2 if highest_frequency is None else highest_frequency
        if n_time_masks > 1 or n_frequency_masks > 1:
            raise NotImplementedError('one time mask and one frequency mask per clip (the reference configuration)')
        self.n_time_masks, self.max_masked_time_steps, self.max_masked_time_rate = n_time_masks, max_masked_time_steps, max_masked_time_rate
        self.n_frequency_masks, self.max_masked_frequency_bands = n_frequency_masks, max_masked_frequency_bands
        self.max_masked_frequency_rate, self.max_noise_scale = max_masked_frequency_rate, max_noise_scale
        self._rng = np.random.RandomState(augmentation_seed)
        # any STFT geometry runs through the reference's own input contract (inputs['stft'] -> pbsed_logmel_from_stft, generic in
        # bins / filters: e.g. the doctest models of pb_sed/models/weak_label/crnn.py:16-34 with stft_size 512); the fused
        # WAVEFORM front-end is built for the experiments' 1024 / 960 / 320 (pb_sed/data_preparation/provider.py:315-323)
        self.fused_waveform_frontend = (stft_size, shift, window_length) == (1024, 320, 960)
        self.sample_rate, self.stft_size, self.number_of_filters = sample_rate, stft_size, number_of_filters
        self.shift, self.window_length, self.eps, self.clamp = shift, window_length, eps, clamp
        fb = get_fbanks(sample_rate, stft_size, number_of_filters, lowest_frequency, highest_frequency)
        self.register_buffer('fbanks', torch.from_numpy(fb))
        # Normalisation state.  Training mode tracks cumulative per-mel statistics over every valid frame seen so far
        # (padertorch Normalization(statistics_axis='bt', momentum=None) inside the reference's extractor, SURVEY.md
        # A.3) and normalises with them; eval mode (or ``freeze_stats``) applies the stored ``mean`` / ``inv_std``.
        self.norm_eps = norm_eps
        self.register_buffer('running_mean', torch.zeros(number_of_filters))
        self.register_buffer('running_power', torch.ones(number_of_filters))
        self.register_buffer('num_tracked_values', torch.zeros(1, dtype=torch.float64))
        self.register_buffer('mean', torch.zeros(number_of_filters))
        self.register_buffer('inv_std', torch.ones(number_of_filters))
        self.freeze_stats = False
        self._tables = None

    def sample_mel_points(self, batch_size):
        """Per-clip fractional STFT-bin positions [B, F + 2] of the warped filter edges / centres (training only): the
        HTK-mel spaced points lowest..highest frequency, pushed through ``frequency_warping_fn`` (one draw per clip)."""
        pts_hz = mel2hz(np.linspace(hz2mel(self.lowest_frequency), hz2mel(self.highest_frequency), self.number_of_filters + 2))
        warped = self.frequency_warping_fn(pts_hz, batch_size)
        return (np.asarray(warped, dtype=np.float64) / self.sample_rate * self.stft_size).astype(np.float32)

    def set_statistics(self, mean, std):
        """Fixed normalisation statistics (e.g. from a data-set pass); training stops tracking."""
        with torch.no_grad():
            self.mean.copy_(torch.as_tensor(mean, dtype=torch.float32))
            self.inv_std.copy_(1. / torch.as_tensor(std, dtype=torch.float32))
        self.freeze_stats = True

    def _load_from_state_dict(self, state_dict, prefix, *args, **kwargs):
        """Accept the reference's key layout: ``feature_extractor.norm.{running_mean,running_power,num_tracked_values}``
        with the broadcast shape [1, 1, F, 1] of padertorch's Normalization, and derive ``mean`` / ``inv_std``."""
        for name in ('running_mean', 'running_power', 'num_tracked_values'):
            ref_key = f'{prefix}norm.{name}'
            if ref_key in state_dict:
                v = state_dict.pop(ref_key)
                state_dict[prefix + name] = (v.reshape(-1)[:1].to(torch.float64) if name == 'num_tracked_values'
                                             else v.reshape(-1).to(torch.float32))
        if prefix + 'running_mean' in state_dict and prefix + 'mean' not in state_dict:
            rm, rp = state_dict[prefix + 'running_mean'], state_dict[prefix + 'running_power']
            state_dict[prefix + 'mean'] = rm.clone()
            state_dict[prefix + 'inv_std'] = 1. / torch.sqrt((rp - rm * rm).clamp_min(0.) + self.norm_eps)
        return super()._load_from_state_dict(state_dict, prefix, *args, **kwargs)

    @property
    def augments(self):
        return self.n_time_masks > 0 or self.n_frequency_masks > 0 or self.max_noise_scale > 0.

    def sample_augmentation(self, seq_len):
        """Host-side draws of one training batch: (masks int32 [B,4] = t_on, t_off, f_on, f_off; noise scales [B]).
        Mask widths ~ U{0..min(max_steps, floor(max_rate * extent))}, onsets uniform over the positions where the mask
        fits (time: inside the clip's seq_len), noise scale ~ U(0, max_noise_scale) per clip."""
        seq_len = np.asarray(seq_len)
        b, f = len(seq_len), self.number_of_filters
        masks = np.zeros((b, 4), np.int32)
        for i in range(b):
            if self.n_time_masks:
                w = self._rng.randint(0, int(min(self.max_masked_time_steps, self.max_masked_time_rate * seq_len[i])) + 1)
                on = self._rng.randint(0, max(int(seq_len[i]) - w, 0) + 1)
                masks[i, :2] = on, on + w
            if self.n_frequency_masks:
                w = self._rng.randint(0, int(min(self.max_masked_frequency_bands, self.max_masked_frequency_rate * f)) + 1)
                on = self._rng.randint(0, f - w + 1)
                masks[i, 2:] = on, on + w
        scales = self._rng.uniform(0., self.max_noise_scale, b).astype(np.float32) if self.max_noise_scale > 0. else None
        return masks, scales


class Normalization(nn.Module):
    def __init__(self, num_channels, eps=1e-3, momentum=0.95):
        super().__init__()
        self.num_channels, self.eps, self.momentum = num_channels, eps, momentum
        self.gamma = nn.Parameter(torch.ones(num_channels))
        self.beta = nn.Parameter(torch.zeros(num_channels))
        self.register_buffer('running_mean', torch.zeros(num_channels))
        self.register_buffer('running_power', torch.ones(num_channels))
        self.freeze_stats = False

    def _load_from_state_dict(self, state_dict, prefix, *args, **kwargs):
        """Accept a padertorch-written checkpoint: its Normalization keeps gamma / beta / running statistics in the
        broadcast shape of the normalised tensor ([1,C,1,1] / [1,C,1]) and a ``num_tracked_values`` counter this build has
        no use for (momentum statistics).  Flatten the former, drop the latter (pb_sed/experiments/weak_label_crnn/
        inference.py:407-413 loads such checkpoints through ``from_storage_dir``)."""
        state_dict.pop(prefix + 'num_tracked_values', None)
        # parameter names: this build follows SURVEY.md App. A (gamma / beta); padertorch is not vendored and its Normalization
        # may store the affine pair as scale / shift - accept either spelling (parity unpinned: no real checkpoint to test with)
        for theirs, ours in (('scale', 'gamma'), ('shift', 'beta')):
            if prefix + theirs in state_dict and prefix + ours not in state_dict:
                state_dict[prefix + ours] = state_dict.pop(prefix + theirs)
        for name in ('gamma', 'beta', 'running_mean', 'running_power'):
            v = state_dict.get(prefix + name)
            if v is not None and v.dim() != 1 and v.numel() == self.num_channels:
                state_dict[prefix + name] = v.reshape(-1)
        return super()._load_from_state_dict(state_dict, prefix, *args, **kwargs)


class ConvLayer(nn.Module):
    def __init__(self, ndim, cin, cout, k, pool=1, pre=False, post=False, eps=1e-3):
        super().__init__()
        self.ndim, self.k, self.pool, self.pre, self.post = ndim, k, pool, pre, post
        self.conv = (nn.Conv2d if ndim == 2 else nn.Conv1d)(cin, cout, k)
        nn.init.xavier_uniform_(self.conv.weight)
        nn.init.zeros_(self.conv.bias)
        self.norm = Normalization(cin if pre else cout, eps=eps) if (pre or post) else None
        p = pool if isinstance(pool, (tuple, list)) else (pool, pool)
        if ndim == 2 and tuple(p) not in ((1, 1), (2, 1)):
            raise NotImplementedError(f'pool {pool}: kernels implement frequency-only (2,1) pooling '
                                      '(pb_sed/experiments/weak_label_crnn/training.py:167)')
        if ndim == 1 and tuple(p) != (1, 1):
            raise NotImplementedError('CNN1d pooling is not used by the reference configs')
        self.pool_f = ndim == 2 and tuple(p) == (2, 1)
        if k not in (1, 3):
            raise NotImplementedError('kernel sizes 1 and 3 only (reference shallow config)')


def plan_residuals(ndim, residual_connections, in_channels, out_channels, pool_sizes, pre_activation):
    """Residual connections of padertorch's CNNs as the reference's 'deep' net_config uses them
    (pb_sed/experiments/weak_label_crnn/training.py:170-183): ``residual_connections[i] = j`` adds the INPUT of layer i
    to the INPUT of layer j > i (pre-activation: both are the raw tensors in front of the layers' norm + ReLU).  Restated
    semantics, parity unpinned (padertorch is absent): where the two tensors differ the skip path applies, in this order,
    every (2,1) max-pool of the layers in between and a bias-carrying 1x1 convolution when the channel counts differ.
    Returns (per-layer destination or None, ModuleDict of the skip convolutions keyed '<src>_<dst>')."""
    n = len(out_channels)
    if residual_connections is None or all(r is None for r in residual_connections):
        return [None] * n, nn.ModuleDict()
    if not pre_activation:
        raise NotImplementedError('residual connections are built for pre-activation stacks (the reference configuration)')
    assert len(residual_connections) == n, (len(residual_connections), n)
    cin = [in_channels] + list(out_channels[:-1])                 # channels of the input of layer i
    skips = nn.ModuleDict()
    for src, dst in enumerate(residual_connections):
        if dst is None:
            continue
        if isinstance(dst, (list, tuple)):
            if len(dst) != 1:
                raise NotImplementedError('one destination per residual connection (the reference configuration)')
            dst = dst[0]
        if not (src < dst < n) or src == 0:
            raise NotImplementedError(f'residual connection {src} -> {dst}: destinations are later layers of the same stack, '
                                      'sources are layers behind the first')
        if cin[src] != cin[dst]:
            conv = (nn.Conv2d if ndim == 2 else nn.Conv1d)(cin[src], cin[dst], 1)
            nn.init.xavier_uniform_(conv.weight)
            nn.init.zeros_(conv.bias)
            skips[f'{src}_{dst}'] = conv
    return [None if d is None else (d[0] if isinstance(d, (list, tuple)) else d) for d in residual_connections], skips


def canonical_state_dict(state_dict, own):
    """A flat reference-written ``state_dict`` under THIS build's names and shapes, for code that matches keys itself
    (trainer.load_init_checkpoint; ``Module.load_state_dict`` does the same through the ``_load_from_state_dict`` overrides):
    skip convolutions ``residual_skip_convs.<s>-><d>.conv.*`` -> ``skip_convs.<s>_<d>.*`` (_CNN._REF_SKIP), a Normalization's
    ``scale`` / ``shift`` -> ``gamma`` / ``beta``, tensors padertorch keeps in the broadcast shape of the normalised tensor
    ([1,C,1,1] / [1,C,1]) flattened to ``own``'s shape.  ``own``: the target's ``state_dict()`` (names and shapes only)."""
    out = {}
    for k, v in state_dict.items():
        m = re.match(r'^(.*\.)?residual_skip_convs\.(\d+)->(\d+)\.conv\.(weight|bias)$', k)
        if m:
            k = f'{m.group(1) or ""}skip_convs.{m.group(2)}_{m.group(3)}.{m.group(4)}'
        elif k.endswith('.norm.scale') and k[:-5] + 'gamma' in own and k[:-5] + 'gamma' not in state_dict:
            k = k[:-5] + 'gamma'
        elif k.endswith('.norm.shift') and k[:-5] + 'beta' in own and k[:-5] + 'beta' not in state_dict:
            k = k[:-5] + 'beta'
        if k in own and torch.is_tensor(v) and tuple(v.shape) != tuple(own[k].shape) and v.numel() == own[k].numel() \
                and sorted(d for d in v.shape if d != 1) == sorted(d for d in own[k].shape if d != 1):
            v = v.reshape(own[k].shape)              # only singleton axes differ
        out[k] = v
    return out


def reference_state_dict(model):
    """``model.state_dict()`` under the reference's (padertorch's) parameter names where this build's differ: the skip
    convolutions of residual stacks (``skip_convs.<src>_<dst>.weight`` -> ``residual_skip_convs.<src>-><dst>.conv.weight``, see
    _CNN._REF_SKIP).  Everything else already carries the reference's names (``convs.<i>.conv.weight``, ``convs.<i>.norm.gamma``,
    ``rnn_fwd.rnn.weight_ih_l0`` ...).  The inverse is built into loading: both spellings are accepted."""
    out = {}
    for k, v in model.state_dict().items():
        m = re.match(r'^(.*\.)?skip_convs\.(\d+)_(\d+)\.(weight|bias)$', k)
        out[f'{m.group(1) or ""}residual_skip_convs.{m.group(2)}->{m.group(3)}.conv.{m.group(4)}' if m else k] = v
    return out


class _CNN(nn.Module):
    ndim = None

    def __init__(self, in_channels, out_channels, kernel_size, pool_size=1, norm='batch', eps=None,
                 pre_activation=False, output_layer=True, input_layer=True, norm_kwargs=None, activation_fn='relu',
                 dropout=0., residual_connections=None, dense_connections=False, pad_type='both', dilation=1, stride=1,
                 gated=False, pool_type='max', pool_stride=None, final_norm=False):
        """Constructor fields of padertorch's CNN2d / CNN1d as the reference configs give them
        (pb_sed/experiments/weak_label_crnn/training.py:218-242); what the HIP kernels do not implement is refused.
        ``final_norm`` (not a padertorch field): SURVEY.md A.4 reading (iii) of ``pre_activation=True, output_layer=False`` -
        the stack closes with a norm + ReLU of its own behind the last conv (``out_norm``)."""
        super().__init__()
        if eps is None:
            eps = (norm_kwargs or {}).get('eps', 1e-3)
        unsupported = dict(activation_fn=(activation_fn, 'relu'), dropout=(dropout, 0.), dense_connections=(dense_connections, False),
                           pad_type=(pad_type, 'both'), dilation=(dilation, 1), stride=(stride, 1), gated=(gated, False),
                           pool_type=(pool_type, 'max'), pool_stride=(pool_stride, None), norm=(norm, 'batch'))
        for name, (got, want) in unsupported.items():
            if got != want and not (name == 'norm' and got is None):
                raise NotImplementedError(f'{type(self).__name__}({name}={got!r}): the MI355X kernels implement {name}={want!r}')
        self.norm_kind, self.eps, self.pre_activation = norm, eps, pre_activation
        self.output_layer, self.input_layer = output_layer, input_layer
        n = len(out_channels)
        ks = kernel_size if isinstance(kernel_size, (list, tuple)) else n * [kernel_size]
        ps = pool_size if isinstance(pool_size, list) and len(pool_size) == n else n * [pool_size]
        self.in_channels, self.out_channels = in_channels, list(out_channels)
        self.kernel_sizes, self.pool_sizes = list(ks), list(ps)
        convs, cin = [], in_channels
        for i, cout in enumerate(out_channels):
            if pre_activation:
                pre, post = norm is not None and not (i == 0 and input_layer), False
            else:
                pre, post = False, norm is not None and not (i == n - 1 and output_layer)
            convs.append(ConvLayer(self.ndim, cin, cout, ks[i], ps[i], pre, post, eps))
            cin = cout
        self.convs = nn.ModuleList(convs)
        if final_norm and not (pre_activation and norm is not None):
            raise NotImplementedError('final_norm closes a pre-activation stack with batch norm')
        self.out_norm = Normalization(cin, eps=eps) if final_norm else None
        self.residual_connections, self.skip_convs = plan_residuals(
            self.ndim, residual_connections, in_channels, self.out_channels, ps, pre_activation)

    # padertorch's spelling of the skip convolutions of a stack with residual connections (the 'deep' net_config of
    # pb_sed/experiments/weak_label_crnn/training.py:170-183): its _CNN keeps them in ``residual_skip_convs``, a ModuleDict
    # keyed '<src>-><dst>', each a conv block whose convolution is the attribute ``conv`` (as in ``convs.<i>.conv``), so a
    # reference checkpoint carries ``<stack>.residual_skip_convs.<src>-><dst>.conv.{weight,bias}`` where this build has
    # ``<stack>.skip_convs.<src>_<dst>.{weight,bias}``.  padertorch is not vendored: the spelling is restated from its
    # source as remembered (parity unpinned, like the Normalization names above) - both spellings load, reference_state_dict
    # writes the reference's.
    _REF_SKIP = re.compile(r'^residual_skip_convs\.(\d+)->(\d+)\.conv\.(weight|bias)$')

    def _load_from_state_dict(self, state_dict, prefix, *args, **kwargs):
        for key in [k for k in state_dict if k.startswith(prefix + 'residual_skip_convs.')]:
            m = self._REF_SKIP.match(key[len(prefix):])
            if m is None:
                continue                                 # (e.g. a norm inside the skip block: left for strict loading to report)
            ours = f'{prefix}skip_convs.{m.group(1)}_{m.group(2)}.{m.group(3)}'
            if ours not in state_dict:
                state_dict[ours] = state_dict.pop(key)
        return super()._load_from_state_dict(state_dict, prefix, *args, **kwargs)

    def freeze(self, num_layers=None, freeze_norm_stats=True):
        layers = self.convs if num_layers is None else self.convs[:num_layers]
        for layer in layers:
            for p in layer.parameters():
                p.requires_grad = False
            if freeze_norm_stats and layer.norm is not None:
                layer.norm.freeze_stats = True


class CNN2d(_CNN):
    ndim = 2


class CNN1d(_CNN):
    ndim = 1


def _pooled_height(height, pool_sizes):
    for p in pool_sizes:
        height //= (p[0] if isinstance(p, (tuple, list)) else p)
    return height


class CNN(nn.Module):
    """padertorch's hybrid CNN (CNN2d -> 'b c f t -> b (c f) t' -> CNN1d) as a parameter container."""

    def __init__(self, cnn_2d, cnn_1d, input_height=128, positional_encoding=False, conditional_dims=0):
        super().__init__()
        if positional_encoding:
            raise NotImplementedError('positional_encoding is not used by the reference configurations')
        self.cnn_2d, self.cnn_1d = cnn_2d, cnn_1d
        self.input_height, self.conditional_dims, self.positional_encoding = input_height, conditional_dims, False

    @classmethod
    def finalize_dogmatic_config(cls, config):
        """hybrid CNN rule: the 1-D stack's input width is the 2-D stack's last channel count times the frequency
        axis left after the (f, 1) pools (pb_sed/models/weak_label/crnn.py:326-330 sets in_channels / input_height)."""
        config['cnn_2d'] = {'factory': CNN2d}
        config['cnn_1d'] = {'factory': CNN1d, 'input_layer': False}     # SURVEY.md A.4 choice (ii): the 1-D stack's first conv has a pre-norm
        c2 = config['cnn_2d']
        if c2.get('out_channels') is not None and config.get('input_height') is not None:
            n = len(c2['out_channels'])
            ps = c2.get('pool_size', 1)
            ps = ps if isinstance(ps, list) and len(ps) == n else n * [ps]
            config['cnn_1d']['in_channels'] = c2['out_channels'][-1] * _pooled_height(config['input_height'], ps)


class GRU(nn.Module):
    """padertorch GRU wrapper mirror: ``rnn`` (a torch.nn.GRU used purely as the parameter container: names, shapes,
    U(+-1/sqrt(H)) init - its forward is never called, the scans run in csrc/gru_stack.hip) + ``output_net`` (CNN1d) +
    the ``reverse`` flag (pb_sed/models/weak_label/crnn.py:338-340)."""

    def __init__(self, rnn, output_net=None, reverse=False):
        super().__init__()
        if not isinstance(rnn, nn.GRU):
            raise NotImplementedError(f'recurrent part must be a torch.nn.GRU, got {type(rnn).__name__}')
        if not rnn.bias or not rnn.batch_first or rnn.dropout:
            raise NotImplementedError('torch.nn.GRU(bias=True, batch_first=True, dropout=0) is what the kernels implement')
        self.rnn, self.output_net, self.reverse = rnn, output_net, reverse
        if rnn.hidden_size % 64:
            raise NotImplementedError('GRU hidden size must be a multiple of 64')

    input_size = property(lambda self: self.rnn.input_size)
    hidden_size = property(lambda self: self.rnn.hidden_size)
    num_layers = property(lambda self: self.rnn.num_layers)
    bidirectional = property(lambda self: self.rnn.bidirectional)

    @classmethod
    def finalize_dogmatic_config(cls, config):
        config['rnn'] = {'factory': nn.GRU}
        config['output_net'] = {'factory': CNN1d}
        rnn = config['rnn']
        for k, v in dict(bias=True, batch_first=True, dropout=0., bidirectional=False, num_layers=1).items():
            if rnn.get(k) is None:
                rnn[k] = v
        if rnn.get('hidden_size') is not None:
            config['output_net']['in_channels'] = rnn['hidden_size'] * (2 if rnn.get('bidirectional') else 1)


DEEP = dict(            # net_config == 'deep', width 2 (pb_sed/experiments/weak_label_crnn/training.py:170-183)
    out_channels_2d=4 * [32] + 4 * [64] + 4 * [128] + 4 * [256] + [512, 512],
    pool_sizes_2d=4 * [1, 1, 1, (2, 1)] + [1, 1],
    kernel_size_2d=9 * [3, 1],
    residual_connections_2d=[None, None, 4, None, 6, None, 8, None, 10, None, 12, None, 14, None, 16, None, None, None],
    out_channels_1d=8 * [512],
    kernel_size_1d=[1] + 3 * [3, 1] + [1],
    residual_connections_1d=[None, 3, None, 5, None, 7, None, None],
)

SHALLOW = dict(
    out_channels_2d=[16, 16, 32, 32, 64, 64, 128, 128, 256],
    pool_sizes_2d=4 * [1, (2, 1)] + [1],
    kernel_size_2d=3,
    out_channels_1d=5 * [256],
    kernel_size_1d=[1, 3, 3, 3, 1],
)


def build_cnn(in_channels, out_channels_2d, pool_sizes_2d, kernel_size_2d, out_channels_1d, kernel_size_1d,
              input_height, conditional_dims=0, eps=1e-3, residual_connections_2d=None, residual_connections_1d=None,
              input_layer_2d=True, input_layer_1d=False, final_norm_1d=False):
    """``input_layer_2d`` / ``input_layer_1d``: padertorch's ``input_layer`` of the two stacks (True = the stack's first layer has no
    pre-activation norm + ReLU) - the defaults are SURVEY.md A.4's reading, the other variants are flags; ``final_norm_1d``:
    reading (iii), the 1-D stack closes with its own norm + ReLU behind the last conv."""
    cnn_2d = CNN2d(in_channels + conditional_dims, out_channels_2d, kernel_size_2d, pool_sizes_2d, eps=eps,
                   pre_activation=True, output_layer=False, input_layer=input_layer_2d, residual_connections=residual_connections_2d)
    f = input_height
    ps = pool_sizes_2d if isinstance(pool_sizes_2d, list) else len(out_channels_2d) * [pool_sizes_2d]
    for p in ps:
        f //= (p[0] if isinstance(p, (tuple, list)) else p)
    cnn_1d = CNN1d(out_channels_2d[-1] * f, out_channels_1d, kernel_size_1d, 1, eps=eps,
                   pre_activation=True, output_layer=False, input_layer=input_layer_1d, residual_connections=residual_connections_1d,
                   final_norm=final_norm_1d)
    return CNN(cnn_2d, cnn_1d, input_height, conditional_dims=conditional_dims)


def build_rnn(input_size, hidden_size, num_layers, num_events, head_hidden, bidirectional=False,
              reverse=False, eps=1e-3):
    dirs = 2 if bidirectional else 1
    output_net = CNN1d(hidden_size * dirs, [head_hidden, num_events], 1, 1, eps=eps, pre_activation=False,
                       output_layer=True)
    rnn = nn.GRU(input_size, hidden_size, num_layers, bias=True, batch_first=True, dropout=0., bidirectional=bidirectional)
    return GRU(rnn, output_net, reverse)
