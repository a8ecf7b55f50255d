"""GPU parity tests, per op: HIP kernels (through the C-ABI) vs the CPU oracle on seeded inputs.
Tolerances: fp32 paths 1e-4 abs on O(1) activations (BASELINE.json north_star), gradients rtol 1e-3."""
import numpy as np
import pytest
import torch
import torch.nn.functional as F

pytestmark = pytest.mark.gpu
DEV = 'cuda:0'


def close(a, b, atol=1e-4, rtol=1e-4, name=''):
    a = a.detach().cpu().double().numpy() if isinstance(a, torch.Tensor) else np.asarray(a, dtype=np.float64)
    b = b.detach().cpu().double().numpy() if isinstance(b, torch.Tensor) else np.asarray(b, dtype=np.float64)
    assert a.shape == b.shape, (name, a.shape, b.shape)
    err = np.abs(a - b)
    tol = atol + rtol * np.abs(b)
    assert (err <= tol).all(), f'{name}: max err {err.max():.3e} at {np.unravel_index(err.argmax(), err.shape)} ' \
                               f'(ref {b.flat[err.argmax()]:.4e}, got {a.flat[err.argmax()]:.4e}, ' \
                               f'{(err > tol).mean() * 100:.2f}% out of tol)'


# ------------------------------------------------------------------------------------------ front-end
@pytest.mark.parametrize('n,b', [(160000, 3), (16000, 2), (5000, 1)])
def test_logmel_vs_oracle(n, b):
    from oracle import frontend as fe
    from pb_sed_amd import ops
    from pb_sed_amd.modules import get_fbanks, num_frames
    g = torch.Generator().manual_seed(1234)
    wav = torch.randn(b, n, generator=g)
    wav = wav / wav.abs().max(-1, keepdim=True)[0]
    t = num_frames(n)
    assert t == fe.num_frames(n)
    seq = np.array([t, max(t - 7, 1), max(t // 2, 1)][:b])
    ext = fe.LogMelExtractor().eval()              # fixed statistics (training mode would track them: one-frame clips -> variance 0)
    ext.mean.copy_(torch.linspace(-8, -4, 128))
    ext.inv_std.copy_(torch.linspace(.3, .6, 128))
    ref, _ = ext(fe.stft(wav), seq_len=seq)
    tables = ops.LogMelTables(get_fbanks(16000, 1024, 128), DEV)
    out = ops.logmel_fwd(wav.to(DEV), tables, ext.mean.to(DEV), ext.inv_std.to(DEV), t,
                         torch.as_tensor(seq, dtype=torch.int32).to(DEV))
    close(out, ref, atol=2e-4, name='logmel')


@pytest.mark.parametrize('n_filters', [40, 64, 200])
def test_logmel_other_filter_counts_and_statistics_vs_oracle(n_filters):
    """Filter counts that are no multiple of the mel stage's 16-filter groups / leave waves without a group / need more than
    one group pair per wave, and the per-mel sums the same launch accumulates (masked sum and sum of squares of what it wrote)."""
    from oracle import frontend as fe
    from pb_sed_amd import ops
    from pb_sed_amd.modules import get_fbanks, num_frames
    g = torch.Generator().manual_seed(99)
    b, n = 3, 40000
    wav = torch.randn(b, n, generator=g)
    wav = wav / wav.abs().max(-1, keepdim=True)[0]
    t = num_frames(n)
    seq = np.array([t, t - 5, t // 3])
    ext = fe.LogMelExtractor(number_of_filters=n_filters).eval()
    ext.mean.copy_(torch.linspace(-8, -4, n_filters))
    ext.inv_std.copy_(torch.linspace(.3, .6, n_filters))
    ref, _ = ext(fe.stft(wav), seq_len=seq)
    tables = ops.LogMelTables(get_fbanks(16000, 1024, n_filters), DEV)
    stats = torch.zeros(32 * n_filters * 2, dtype=torch.float64, device=DEV)
    out = ops.logmel_fwd(wav.to(DEV), tables, ext.mean.to(DEV), ext.inv_std.to(DEV), t,
                         torch.as_tensor(seq, dtype=torch.int32).to(DEV), stats=stats)
    close(out, ref, atol=2e-4, name=f'logmel {n_filters} filters')
    sums = stats.view(32, n_filters, 2).sum(0).cpu()
    o = out.double().cpu()[:, 0]
    assert torch.allclose(sums[:, 0], o.sum((0, 2)), rtol=1e-6, atol=1e-3) and torch.allclose(sums[:, 1], (o * o).sum((0, 2)), rtol=1e-6, atol=1e-3)


def test_logmel_time_warped_frames_vs_oracle():
    """pbsed_logmel_fwd_frames: the front-end at explicit frame positions (time-warped STFT, data.TimeWarp).  The regular
    grid reproduces pbsed_logmel_fwd bit for bit; warped positions (incl. windows hanging over both clip ends) match the
    oracle's STFT taken at the same positions; statistics tracking and mel warping compose with it."""
    from oracle import frontend as fe
    from pb_sed_amd import data, modules, ops
    from pb_sed_amd.modules import get_fbanks, num_frames
    g = torch.Generator().manual_seed(4321)
    b, n = 5, 48000
    wav = torch.randn(b, n, generator=g)
    wav = wav / wav.abs().max(-1, keepdim=True)[0]
    t = num_frames(n)
    seq = np.array([t, t - 3, t, t // 2, t - 1])
    seq_dev = torch.as_tensor(seq, dtype=torch.int32).to(DEV)
    ext = fe.LogMelExtractor().eval()
    ext.mean.copy_(torch.linspace(-8, -4, 128))
    ext.inv_std.copy_(torch.linspace(.3, .6, 128))
    tables = ops.LogMelTables(get_fbanks(16000, 1024, 128), DEV)
    mean, inv_std = ext.mean.to(DEV), ext.inv_std.to(DEV)
    tw = data.TimeWarp(modules.Uniform(.4, .6, seed=3), modules.Uniform(-.1, .1, seed=4))
    plain = ops.logmel_fwd(wav.to(DEV), tables, mean, inv_std, t, seq_dev)
    grid = tw.frame_positions(n, t, np.full(b, .5), np.zeros(b))
    same = ops.logmel_fwd(wav.to(DEV), tables, mean, inv_std, t, seq_dev, frame_pos=torch.from_numpy(grid).to(DEV))
    assert torch.equal(plain, same)
    a, s = tw.sample(b)
    pos = tw.frame_positions(n, t, a, s)
    pos[0, :3] -= 700                                                  # windows starting well before / ending after the clip
    pos[1, -3:] += 900
    ref, _ = ext(fe.stft(wav, frame_pos=pos), seq_len=seq)
    out = ops.logmel_fwd(wav.to(DEV), tables, mean, inv_std, t, seq_dev, frame_pos=torch.from_numpy(pos).to(DEV))
    close(out, ref, atol=2e-4, name='logmel warped frames')
    assert (out - plain).abs().max() > .5                              # it is a different signal


# ------------------------------------------------------------------------------------------ conv
CONV_CASES = [
    # cin, cout, F, T, k(2d? tuple), pool, prologue, ragged
    dict(cin=1, cout=16, f=12, t=70, k=(3, 3), pool=False, pro=False),
    dict(cin=16, cout=16, f=8, t=150, k=(3, 3), pool=True, pro=True),
    dict(cin=16, cout=32, f=8, t=65, k=(3, 3), pool=False, pro=True),
    dict(cin=32, cout=64, f=4, t=64, k=(3, 3), pool=True, pro=True),
    dict(cin=24, cout=128, f=6, t=50, k=(3, 3), pool=True, pro=True),
    dict(cin=20, cout=256, f=4, t=33, k=(3, 3), pool=False, pro=True),
    dict(cin=11, cout=16, f=6, t=40, k=(3, 3), pool=False, pro=False),
    dict(cin=1, cout=48, f=6, t=70, k=(3, 3), pool=True, pro=False),      # few input channels, wide output (width-2 first layer)
    dict(cin=3, cout=160, f=4, t=33, k=(3, 3), pool=False, pro=True),
    dict(cin=64, cout=256, f=1, t=130, k=(1, 1), pool=False, pro=True),
    dict(cin=40, cout=10, f=1, t=77, k=(1, 1), pool=False, pro=True),
    dict(cin=48, cout=96, f=1, t=100, k=(1, 3), pool=False, pro=True),
    dict(cin=266, cout=768, f=1, t=45, k=(1, 1), pool=False, pro=False),
    # Conv1d weight gradients of the producer / consumer bf16x3 kernel (>= 64 channels either side): 3 taps with rows off the
    # 16-byte alignment, the CNN1d shape, channel counts off the 128 / 64 tiles
    dict(cin=96, cout=160, f=1, t=203, k=(1, 3), pool=False, pro=True),
    dict(cin=256, cout=256, f=1, t=500, k=(1, 3), pool=False, pro=True),
    dict(cin=200, cout=72, f=1, t=64, k=(1, 3), pool=False, pro=False),
    dict(cin=520, cout=136, f=1, t=130, k=(1, 1), pool=False, pro=True),      # k = 1: wide inputs only (>= 512 channels)
    # 3x3 weight gradients of the 16- / 32-channel layers on the producer / consumer bf16x3 kernel with time-sliced consumer
    # waves (rows 16-byte aligned: T % 4 == 0), partial time tiles, channel counts off the tiles
    dict(cin=16, cout=16, f=8, t=152, k=(3, 3), pool=True, pro=True),
    dict(cin=16, cout=32, f=6, t=260, k=(3, 3), pool=False, pro=True),
    dict(cin=32, cout=32, f=8, t=132, k=(3, 3), pool=True, pro=True),
    dict(cin=24, cout=40, f=5, t=100, k=(3, 3), pool=False, pro=False),
    dict(cin=32, cout=128, f=4, t=64, k=(3, 3), pool=True, pro=True),
    # ... and of the register-resident column walk (conv_wgrad_s16.h): several 16-row segments per column with a short last
    # one, 32-t chunks cut by T and by the clip lengths, no prologue
    dict(cin=16, cout=16, f=36, t=100, k=(3, 3), pool=True, pro=True),
    dict(cin=32, cout=32, f=22, t=72, k=(3, 3), pool=False, pro=True),
    dict(cin=32, cout=16, f=17, t=36, k=(3, 3), pool=False, pro=False),
    dict(cin=16, cout=32, f=48, t=64, k=(3, 3), pool=True, pro=False),
    dict(cin=11, cout=16, f=20, t=64, k=(3, 3), pool=True, pro=True),       # fewer than 16 input channels: idle lanes, unwritten columns
    dict(cin=5, cout=32, f=6, t=36, k=(3, 3), pool=False, pro=True),
    # 1x1 conv2d layers (every second layer of net_config 'deep'; forward / data gradient in the bf16 tests below, the
    # weight gradient here): one row per tile, two under a (2,1) pool
    dict(cin=64, cout=96, f=6, t=72, k=(1, 1), pool=True, pro=True),
    dict(cin=40, cout=48, f=5, t=100, k=(1, 1), pool=False, pro=True),
    dict(cin=32, cout=64, f=4, t=50, k=(1, 1), pool=False, pro=False),
    dict(cin=136, cout=200, f=5, t=100, k=(1, 1), pool=False, pro=True),     # weight gradient: the Conv1d producer / consumer kernel over rows
    dict(cin=128, cout=128, f=3, t=36, k=(1, 1), pool=False, pro=False),
]


def _rbf(v):
    """Round to the nearest bf16 (ties to even) as the kernels do to an fp32 operand; float64 out."""
    return v.float().to(torch.bfloat16).double()


class _RoundedConv(torch.autograd.Function):
    """The bf16 mode of one conv launch restated (oracle/bf16emu.py::_ConvBf16 for a test-local reference): forward from
    rounded operands, each backward product from ITS rounded operands."""

    @staticmethod
    def forward(ctx, a, w, bias):
        ar, wr = _rbf(a), _rbf(w)
        ctx.save_for_backward(ar, wr)
        return F.conv2d(ar, wr, bias)

    @staticmethod
    def backward(ctx, gy):
        ar, wr = ctx.saved_tensors
        gr = _rbf(gy)
        return torch.nn.grad.conv2d_input(ar.shape, wr, gr), torch.nn.grad.conv2d_weight(ar, wr.shape, gr), gr.sum((0, 2, 3))


def _conv_ref(x, w, bias, scale, shift, seq, k, pool, rounded=False):
    """CPU reference of one layer: [BN-apply -> ReLU -> mask] -> pad -> conv -> pool.  ``rounded``: the bf16 mode - operands
    of every product rounded to bf16 where the kernels round them (the activation is formed in fp32 first, as fmaf does)."""
    a = x
    if scale is not None:
        a = x * scale.view(1, -1, 1, 1) + shift.view(1, -1, 1, 1)
        if rounded:
            a = a + (a.detach().float().double() - a.detach())          # the value the kernel holds before it rounds to bf16
        a = F.relu(a)
        m = (torch.arange(x.shape[-1])[None] < torch.as_tensor(seq)[:, None]).to(x.dtype)
        a = a * m[:, None, None, :]
    ph, pw = k[0] - 1, k[1] - 1
    a = F.pad(a, (pw // 2, pw - pw // 2, ph // 2, ph - ph // 2))
    y = _RoundedConv.apply(a, w, bias) if rounded else F.conv2d(a, w, bias)
    if pool:
        y, idx = F.max_pool2d(y, (2, 1), return_indices=True)
    return y


@pytest.mark.parametrize('case', CONV_CASES, ids=lambda c: f"{c['cin']}x{c['cout']}k{c['k'][0]}{c['k'][1]}p{int(c['pool'])}")
def test_conv_fwd_bwd_vs_torch(case):
    from pb_sed_amd import ops
    torch.manual_seed(0)
    b, cin, cout, f, t, k, pool, pro = 3, case['cin'], case['cout'], case['f'], case['t'], case['k'], case['pool'], case['pro']
    x = torch.randn(b, cin, f, t, dtype=torch.float64)
    w = (torch.randn(cout, cin, *k, dtype=torch.float64) / np.sqrt(cin * k[0] * k[1])).requires_grad_()
    bias = torch.randn(cout, dtype=torch.float64).requires_grad_()
    seq = np.array([t, max(t - 9, 1), max(t // 2, 1)])
    scale = (torch.rand(cin, dtype=torch.float64) + .5) if pro else None
    shift = torch.randn(cin, dtype=torch.float64) * .3 if pro else None
    xr = x.clone().requires_grad_()
    y_ref = _conv_ref(xr, w, bias, scale, shift, seq, k, pool)
    gy = torch.randn_like(y_ref)
    y_ref.backward(gy)

    dx = lambda a: None if a is None else a.float().to(DEV).contiguous()
    seq_dev = torch.as_tensor(seq, dtype=torch.int32).to(DEV)
    wd = w.detach().float().to(DEV)
    pc = ops.PackedConv(wd if k[0] > 1 else wd)
    xd = dx(x)
    y, idx, stats = ops.conv_fwd(xd, pc, pc.fwd(), bias=dx(bias.detach()), scale=dx(scale), shift=dx(shift),
                                 relu=True, seq_len=seq_dev, pool=pool, want_stats=True)
    close(y, y_ref, name='conv_fwd')
    # statistics epilogue: masked sums of the produced tensor
    m = (torch.arange(t)[None] < torch.as_tensor(seq)[:, None]).double()[:, None, None, :]
    yd = y_ref.detach()
    close(stats.sum(0)[:, 0], (yd * m).sum((0, 2, 3)), atol=1e-3, rtol=1e-4, name='stats_sum')
    close(stats.sum(0)[:, 1], (yd * yd * m).sum((0, 2, 3)), atol=1e-3, rtol=1e-4, name='stats_sumsq')
    # weight / bias gradient
    dw = torch.zeros_like(wd)
    db = torch.zeros(cout, device=DEV)
    ops.conv_bwd_weight(xd, dx(gy), pc, dw, db, scale=dx(scale), shift=dx(shift), relu=True, seq_len=seq_dev,
                        unpool_idx=idx)
    close(dw, w.grad, atol=2e-4, rtol=1e-3, name='conv_wgrad')
    close(db, bias.grad, atol=2e-4, rtol=1e-3, name='conv_bgrad')
    # data gradient (plain: only valid without prologue)
    if not pro:
        g, _ = ops.conv_bwd_data(dx(gy), pc, pc.dgrad(), xd.shape, idx, None)
        close(g, xr.grad, atol=2e-4, rtol=1e-3, name='conv_dgrad')


def test_conv_bn_relu_backward_chain_vs_autograd():
    """conv_i output -> Normalization(train) -> ReLU -> conv_{i+1}: statistics epilogue, bn_finalize,
    fused dgrad epilogue and bn_bwd_apply against autograd through the oracle layers."""
    from oracle import nn as onn
    from pb_sed_amd import ops
    torch.manual_seed(1)
    b, c0, c1, c2, f, t = 3, 8, 16, 24, 8, 90
    seq = np.array([90, 71, 40])
    l1 = onn._ConvLayer(2, c0, c1, 3, pool=(2, 1), pre=False).double()
    l2 = onn._ConvLayer(2, c1, c2, 3, pool=1, pre=True).double()
    with torch.no_grad():
        l2.norm.gamma.uniform_(.5, 1.5)
        l2.norm.beta.normal_(0, .2)
        l1.conv.bias.normal_(0, .1)
        l2.conv.bias.normal_(0, .1)
    x = torch.randn(b, c0, f, t, dtype=torch.float64)
    y1 = l1(x, seq)
    y2 = l2(y1, seq)
    gy = torch.randn_like(y2)
    y2.backward(gy)

    dd = lambda a: a.detach().float().to(DEV).contiguous()
    seq_dev = torch.as_tensor(seq, dtype=torch.int32).to(DEV)
    pc1, pc2 = ops.PackedConv(dd(l1.conv.weight)), ops.PackedConv(dd(l2.conv.weight))
    y1d, idx1, stats = ops.conv_fwd(dd(x), pc1, pc1.fwd(), bias=dd(l1.conv.bias), seq_len=seq_dev, pool=True,
                                    want_stats=True)
    close(y1d, y1, name='y1')

    class N:  # norm parameter holder on device
        gamma, beta = dd(l2.norm.gamma), dd(l2.norm.beta)
        eps, momentum = l2.norm.eps, l2.norm.momentum
        running_mean, running_power = torch.zeros(c1, device=DEV), torch.ones(c1, device=DEV)
    count = float(seq.sum() * (f // 2))
    st = ops.bn_finalize(stats, count, N)
    close(N.running_mean, l2.norm.running_mean, name='running_mean')
    close(N.running_power, l2.norm.running_power, name='running_power')
    y2d, _, _ = ops.conv_fwd(y1d, pc2, pc2.fwd(), bias=dd(l2.conv.bias), scale=st.scale, shift=st.shift,
                             seq_len=seq_dev)
    close(y2d, y2, name='y2')
    dz, bstats = ops.conv_bwd_data(dd(gy), pc2, pc2.dgrad(), y1d.shape, None, seq_dev,
                                   bn=(y1d, st.mean, st.invstd, st.scale, st.shift))
    dgamma, dbeta = torch.zeros(c1, device=DEV), torch.zeros(c1, device=DEV)
    g1 = ops.bn_backward(dz, y1d, st, bstats, count, dgamma, dbeta, seq_dev)
    close(dgamma, l2.norm.gamma.grad, atol=2e-4, rtol=1e-3, name='dgamma')
    close(dbeta, l2.norm.beta.grad, atol=2e-4, rtol=1e-3, name='dbeta')
    dw1 = torch.zeros_like(dd(l1.conv.weight))
    db1 = torch.zeros(c1, device=DEV)
    ops.conv_bwd_weight(dd(x), g1, pc1, dw1, db1, seq_len=seq_dev, unpool_idx=idx1)
    close(dw1, l1.conv.weight.grad, atol=2e-4, rtol=1e-3, name='dw1 (through pool + BN backward)')
    close(db1, l1.conv.bias.grad, atol=2e-4, rtol=1e-3, name='db1')


# ------------------------------------------------------------------------------------------ GRU
@pytest.mark.parametrize('b,h,t,ragged', [(5, 64, 23, True), (32, 256, 40, False), (17, 128, 31, True)])
def test_gru_scan_fwd_bwd_vs_torch(b, h, t, ragged):
    from oracle import nn as onn
    from pb_sed_amd import ops
    torch.manual_seed(2)
    cin = 24
    seq = np.sort(np.random.RandomState(0).randint(t // 2, t + 1, b))[::-1].copy() if ragged else np.full(b, t)
    seq[0] = t
    x = torch.randn(b, cin, t)
    grus = [onn.GRU(cin, h, 1, reverse=False), onn.GRU(cin, h, 1, reverse=True)]
    xs = [x.clone().requires_grad_() for _ in grus]
    ys = [g(xi, seq)[0] for g, xi in zip(grus, xs)]
    gys = [torch.randn_like(y) for y in ys]
    for y, gy in zip(ys, gys):
        y.backward(gy)
    seq_dev = torch.as_tensor(seq, dtype=torch.int32).to(DEV)
    dd = lambda a: a.detach().float().to(DEV).contiguous()
    gi = []
    for g in grus:
        gi_bct = torch.einsum('oc,bct->bot', g.rnn.weight_ih_l0, x) + g.rnn.bias_ih_l0[None, :, None]
        gi.append(ops.bct_to_tbc(dd(gi_bct)))
    hs, save = ops.gru_scan_fwd(gi, [dd(g.rnn.weight_hh_l0) for g in grus], [dd(g.rnn.bias_hh_l0) for g in grus],
                                [0, 1], seq_dev)
    for i in range(2):
        close(ops.tbc_to_bct(hs[i]), ys[i], name=f'gru fwd chain{i}')
    dy = [ops.bct_to_tbc(dd(gy)) for gy in gys]
    dgi, dgh = ops.gru_scan_bwd([ops.transpose2d(dd(g.rnn.weight_hh_l0)) for g in grus], hs, save, dy, [0, 1], seq_dev)
    for i, g in enumerate(grus):
        dgi_b, dgh_b = ops.tbc_to_bct(dgi[i]).cpu(), ops.tbc_to_bct(dgh[i]).cpu()
        hprev = ops.tbc_to_bct(hs[i], shift=1 if i else -1).cpu()
        close(torch.einsum('bot,bct->oc', dgi_b, x), g.rnn.weight_ih_l0.grad, atol=3e-4, rtol=1e-3, name=f'dW_ih{i}')
        close(dgi_b.sum((0, 2)), g.rnn.bias_ih_l0.grad, atol=3e-4, rtol=1e-3, name=f'db_ih{i}')
        close(torch.einsum('bot,bct->oc', dgh_b, hprev), g.rnn.weight_hh_l0.grad, atol=3e-4, rtol=1e-3, name=f'dW_hh{i}')
        close(dgh_b.sum((0, 2)), g.rnn.bias_hh_l0.grad, atol=3e-4, rtol=1e-3, name=f'db_hh{i}')
        close(torch.einsum('bot,oc->bct', dgi_b, g.rnn.weight_ih_l0.detach()), xs[i].grad, atol=3e-4, rtol=1e-3, name=f'dx{i}')


# ------------------------------------------------------------------------------------------ losses
@pytest.mark.parametrize('name', ['ragged_strong', 'full_len', 'no_bwd', 'slat', 'weak_only',
                                  'half_weight_smooth', 'class_weights'])
def test_fbcrnn_loss_kernel_vs_reference_golden(golden, name):
    import ast
    from pb_sed_amd import ops
    g = golden('ref_fbcrnn_loss.npz')
    kw = ast.literal_eval(str(g[f'{name}/kw']))
    dd = lambda a: torch.as_tensor(a).float().to(DEV).contiguous()
    yb = dd(g[f'{name}/y_bwd']) if f'{name}/y_bwd' in g else None
    cw = dd(np.array(kw['class_weights'], dtype=np.float32)) if 'class_weights' in kw else None
    loss, _, _, d_f, d_b = ops.fbcrnn_loss(
        dd(g[f'{name}/y_fwd']), yb, dd(g[f'{name}/weak_targets']), dd(g[f'{name}/boundary_targets']),
        torch.as_tensor(g[f'{name}/seq_len'], dtype=torch.int32).to(DEV), minimum_score=1e-5,
        strong_weight=kw.get('strong_fwd_bwd_loss_weight', 1.), slat=kw.get('slat', False),
        label_smoothing=kw.get('label_smoothing', 0.), class_weights=cw, inputs_are_scores=True)
    assert loss.item() == pytest.approx(float(g[f'{name}/loss']), rel=2e-5)
    close(d_f, g[f'{name}/grad_y_fwd'], atol=1e-6, rtol=2e-4, name='dL/dy_fwd')
    if yb is not None:
        close(d_b, g[f'{name}/grad_y_bwd'], atol=1e-6, rtol=2e-4, name='dL/dy_bwd')


@pytest.mark.parametrize('name', ['a', 'b'])
def test_bicrnn_loss_kernel_vs_reference_golden(golden, name):
    from pb_sed_amd import ops
    g = golden('ref_bicrnn_loss.npz')
    dd = lambda a: torch.as_tensor(a).float().to(DEV).contiguous()
    loss, _, d = ops.bicrnn_loss(dd(g[f'{name}/y']), dd(g[f'{name}/strong_targets']),
                                 torch.as_tensor(g[f'{name}/seq_len'], dtype=torch.int32).to(DEV),
                                 inputs_are_scores=True)
    assert loss.item() == pytest.approx(float(g[f'{name}/loss']), rel=2e-5)
    close(d, g[f'{name}/grad_y'], atol=1e-7, rtol=2e-4, name='dL/dy')


def test_squash_roundtrip():
    from pb_sed_amd import ops
    x = torch.randn(1000, device=DEV) * 4
    y = ops.squash_fwd(x, 1e-5)
    close(y, 1e-5 + (1 - 2e-5) * torch.sigmoid(x.cpu()), atol=1e-6)
    dy = torch.randn(1000, device=DEV)
    s = torch.sigmoid(x.cpu().double())
    close(ops.squash_bwd(y, dy, 1e-5), dy.cpu().double() * (1 - 2e-5) * s * (1 - s), atol=1e-5, rtol=1e-3)


# ------------------------------------------------------------------------------------------ optimiser
def test_adam_and_grad_norm_vs_torch():
    from pb_sed_amd import ops
    torch.manual_seed(3)
    n = 100003
    p0, grads = torch.randn(n), [torch.randn(n) * s for s in (1., .1, 3.)]
    pr = p0.clone().requires_grad_()
    opt = torch.optim.Adam([pr], lr=5e-4)
    p, m, v = p0.to(DEV), torch.zeros(n, device=DEV), torch.zeros(n, device=DEV)
    ss, norm = torch.zeros((), dtype=torch.float64, device=DEV), torch.zeros((), device=DEV)
    for step, g in enumerate(grads, 1):
        pr.grad = g.clone()
        ref_norm = torch.nn.utils.clip_grad_norm_([pr], 5.)
        opt.step()
        gd = g.to(DEV)
        ops.grad_sumsq(gd, ss)
        ops.adam_step(p, gd, m, v, lr=5e-4, step=step, max_norm=5., sumsq=ss, norm_out=norm)
        assert norm.item() == pytest.approx(ref_norm.item(), rel=1e-5)
    close(p, pr, atol=1e-6, rtol=1e-5, name='adam params after 3 steps')


def test_adam_skips_the_update_behind_a_raised_scan_flag():
    """A persistent GRU scan that timed out leaves an error word set; the fused Adam given those words must not touch
    parameters or moments (Trainer.step hands them over, the host raises once it has read them)."""
    from pb_sed_amd import ops
    n = 4099
    p, g = torch.randn(n, device=DEV), torch.randn(n, device=DEV)
    m, v = torch.zeros(n, device=DEV), torch.zeros(n, device=DEV)
    p0 = p.clone()
    flags = torch.zeros(ops.GRU_FLAG_WORDS, dtype=torch.int32, device=DEV)
    flags[5] = 1
    ops.adam_step(p, g, m, v, lr=1e-2, step=1, skip_flags=flags)
    assert torch.equal(p, p0) and not m.any() and not v.any()
    flags.zero_()
    ops.adam_step(p, g, m, v, lr=1e-2, step=1, skip_flags=flags)
    assert not torch.equal(p, p0) and m.any()
    with pytest.raises(RuntimeError, match='timed out'):
        ops.gru_flags_raise(np.array([0, 0, 1]))
    ops.gru_flags_raise(np.zeros(4))


@pytest.mark.parametrize('persist', ['2', '0'])
@pytest.mark.parametrize('b,h,t,ragged', [(5, 64, 23, True), (32, 256, 30, False), (19, 128, 17, True), (16, 512, 5, False),
                                          (1, 128, 40, True)])
def test_gru_stack_wavefront_vs_torch(b, h, t, ragged, persist, monkeypatch):
    """2-layer forward + time-reversed stacks vs the oracle wrapper around nn.GRU, through both scan
    implementations: persistent granule exchange ('2', default) and one launch per step ('0')."""
    from oracle import nn as onn
    from pb_sed_amd import ops
    monkeypatch.setenv('PBSED_GRU_PERSIST', persist)
    torch.manual_seed(4)
    cin, nl = 24, 2
    seq = np.sort(np.random.RandomState(1).randint(t // 2, t + 1, b))[::-1].copy() if ragged else np.full(b, t)
    seq[0] = t
    x = torch.randn(b, cin, t)
    grus = [onn.GRU(cin, h, nl, reverse=False), onn.GRU(cin, h, nl, reverse=True)]
    xs = [x.clone().requires_grad_() for _ in grus]
    ys = [g(xi, seq)[0] for g, xi in zip(grus, xs)]
    gys = [torch.randn_like(y) for y in ys]
    for y, gy in zip(ys, gys):
        y.backward(gy)
    seq_dev = torch.as_tensor(seq, dtype=torch.int32).to(DEV)
    dd = lambda a: a.detach().float().to(DEV).contiguous()
    P = lambda g, n, l: getattr(g.rnn, f'{n}_l{l}')
    gi0 = []
    for g in grus:
        gi_bct = torch.einsum('oc,bct->bot', g.rnn.weight_ih_l0, x) + g.rnn.bias_ih_l0[None, :, None]
        gi0.append(ops.bct_to_tbc(dd(gi_bct)))
    idx = [(g, l) for g in grus for l in range(nl)]
    hs, save = ops.gru_stack_fwd(gi0, [dd(P(g, 'weight_ih', l)) if l else None for g, l in idx],
                                 [dd(P(g, 'bias_ih', l)) if l else None for g, l in idx],
                                 [dd(P(g, 'weight_hh', l)) for g, l in idx], [dd(P(g, 'bias_hh', l)) for g, l in idx],
                                 [0, 1], seq_dev, nl)
    for c in range(2):
        close(ops.tbc_to_bct(hs[c * nl + nl - 1]), ys[c], name=f'stack fwd chain{c}')
    dy = [ops.bct_to_tbc(dd(gy)) for gy in gys]
    dgi, dgh = ops.gru_stack_bwd([ops.transpose2d(dd(P(g, 'weight_hh', l))) for g, l in idx],
                                 [ops.transpose2d(dd(P(g, 'weight_ih', l + 1))) if l + 1 < nl else None for g, l in idx],
                                 hs, save, dy, [0, 1], seq_dev, nl)
    for c, g in enumerate(grus):
        for l in range(nl):
            i = c * nl + l
            dgi_b, dgh_b = ops.tbc_to_bct(dgi[i]).cpu(), ops.tbc_to_bct(dgh[i]).cpu()
            hprev = ops.tbc_to_bct(hs[i], shift=1 if c else -1).cpu()
            xin = x if l == 0 else ops.tbc_to_bct(hs[i - 1]).cpu()
            tag = f'chain{c} layer{l}'
            close(torch.einsum('bot,bct->oc', dgi_b, xin), P(g, 'weight_ih', l).grad, atol=3e-4, rtol=1e-3, name='dW_ih ' + tag)
            close(dgi_b.sum((0, 2)), P(g, 'bias_ih', l).grad, atol=3e-4, rtol=1e-3, name='db_ih ' + tag)
            close(torch.einsum('bot,bct->oc', dgh_b, hprev), P(g, 'weight_hh', l).grad, atol=3e-4, rtol=1e-3, name='dW_hh ' + tag)
            close(dgh_b.sum((0, 2)), P(g, 'bias_hh', l).grad, atol=3e-4, rtol=1e-3, name='db_hh ' + tag)
        dgi0 = ops.tbc_to_bct(dgi[c * nl]).cpu()
        close(torch.einsum('bot,oc->bct', dgi0, g.rnn.weight_ih_l0.detach()), xs[c].grad, atol=3e-4, rtol=1e-3, name=f'dx chain{c}')
    ops.check_gru_sync()


BF16_CASES = [c for c in CONV_CASES if c['cin'] >= 20] + [
    dict(cin=64, cout=64, f=8, t=70, k=(3, 3), pool=False, pro=False),
    dict(cin=32, cout=40, f=1, t=200, k=(1, 3), pool=False, pro=False),
]



@pytest.mark.parametrize('precision', ['bf16', 'bf16x3'])
@pytest.mark.parametrize('case', BF16_CASES, ids=lambda c: f"{c['cin']}x{c['cout']}k{c['k'][0]}{c['k'][1]}p{int(c['pool'])}{'pro' if c['pro'] else ''}")
def test_conv_bf16_mfma_vs_torch(case, precision):
    """bf16-MFMA conv family: plain bf16 operands (config-3 compute dtype, bf16 tolerance) and the exact 3-way
    split (fp32-class accuracy, checked at the fp32 tolerance)."""
    from pb_sed_amd import ops
    torch.manual_seed(0)
    b, cin, cout, f, t, k, pool, pro = 3, case['cin'], case['cout'], case['f'], case['t'], case['k'], case['pool'], case['pro']
    x = torch.randn(b, cin, f, t, dtype=torch.float64)
    w = (torch.randn(cout, cin, *k, dtype=torch.float64) / np.sqrt(cin * k[0] * k[1]))
    bias = torch.randn(cout, dtype=torch.float64)
    seq = np.array([t, max(t - 9, 1), max(t // 2, 1)])
    scale = (torch.rand(cin, dtype=torch.float64) + .5) if pro else None
    shift = torch.randn(cin, dtype=torch.float64) * .3 if pro else None
    xr = x.clone().requires_grad_()
    y_ref = _conv_ref(xr, w, bias, scale, shift, seq, k, pool)
    gy = torch.randn_like(y_ref)
    y_ref.backward(gy)
    tol = dict(atol=1e-4, rtol=1e-4) if precision == 'bf16x3' else dict(atol=4e-2, rtol=4e-2)
    dx = lambda a: None if a is None else a.float().to(DEV).contiguous()
    seq_dev = torch.as_tensor(seq, dtype=torch.int32).to(DEV)
    pc = ops.PackedConv(dx(w))
    xd = dx(x)
    y, idx, stats = ops.conv_fwd(xd, pc, pc.fwd(precision), bias=dx(bias), scale=dx(scale), shift=dx(shift), relu=True,
                                 seq_len=seq_dev, pool=pool, want_stats=True, precision=precision)
    close(y, y_ref, name=f'conv_fwd {precision}', **tol)
    if not pro and not pool:
        g, _ = ops.conv_bwd_data(dx(gy), pc, pc.dgrad(precision), xd.shape, None, None, precision=precision)
        close(g, xr.grad, name=f'conv_dgrad {precision}', **tol)
    if k == (1, 1) and f > 1:
        # a residual connection ending at the layer (net_config 'deep'): added to the biased, pooled output in the epilogue
        res = torch.randn(y.shape, generator=torch.Generator().manual_seed(5)).to(DEV)
        y2, idx2, _ = ops.conv_fwd(xd, pc, pc.fwd(precision), bias=dx(bias), scale=dx(scale), shift=dx(shift), relu=True,
                                   seq_len=seq_dev, pool=pool, want_stats=True, precision=precision, residual=res)
        for i, n in enumerate(seq):
            close(y2[i, ..., :n], (y + res)[i, ..., :n], name=f'conv_fwd {precision} + residual', atol=1e-6, rtol=1e-6)
        assert idx is None or torch.equal(idx, idx2)
    if precision == 'bf16':
        # THE bf16 gate: against the same layer from operands rounded to bf16 where the kernel rounds them - what is left is
        # fp32 accumulation order (the comparison above only bounds the inherent effect of the rounding, 4e-2)
        x32 = x.float().double()                      # the kernel's inputs are these fp32 values
        xq = x32.clone().requires_grad_()
        w32 = w.float().double()
        y_q = _conv_ref(xq, w32, bias.float().double(), None if scale is None else scale.float().double(),
                        None if shift is None else shift.float().double(), seq, k, pool, rounded=True)
        close(y, y_q, name='conv_fwd bf16 vs rounded operands', atol=2e-5, rtol=2e-5)
        if not pro and not pool:
            y_q.backward(gy.float().double())
            close(g, xq.grad, name='conv_dgrad bf16 vs rounded operands', atol=2e-5, rtol=2e-5)


WGRAD_BF16_CASES = [
    dict(cin=32, cout=32, f=8, t=70, k=(3, 3), pool=False, pro=True),
    dict(cin=64, cout=128, f=6, t=133, k=(3, 3), pool=True, pro=True),          # un-pooled dY, odd T (scalar idx loads)
    dict(cin=40, cout=72, f=4, t=96, k=(3, 3), pool=False, pro=False),          # channel counts off the 32 / 64 tiles
    dict(cin=128, cout=128, f=4, t=64, k=(3, 3), pool=True, pro=True),
    dict(cin=64, cout=64, f=8, t=96, k=(3, 3), pool=False, pro=True),           # the one-part producer / consumer kernel (T % 4 == 0, >= 64 channels)
    dict(cin=96, cout=160, f=6, t=100, k=(3, 3), pool=True, pro=True),          # ... channel counts off its 64-wide tiles, ragged last column
    dict(cin=72, cout=64, f=3, t=36, k=(3, 3), pool=False, pro=False),          # ... odd row count, one short column
    dict(cin=64, cout=48, f=1, t=200, k=(1, 3), pool=False, pro=True),
    dict(cin=96, cout=256, f=1, t=130, k=(1, 1), pool=False, pro=True),
    dict(cin=266, cout=64, f=1, t=77, k=(1, 1), pool=False, pro=False),
]


@pytest.mark.parametrize('case', WGRAD_BF16_CASES, ids=lambda c: f"{c['cin']}x{c['cout']}k{c['k'][0]}{c['k'][1]}p{int(c['pool'])}{'pro' if c['pro'] else ''}")
def test_conv_wgrad_bf16_vs_torch(case):
    """Weight / bias gradient on bf16 MFMA (time as the contraction index, shifted kw operands built in registers):
    against the float64 reference at the bf16 tolerance, and against the fp32 kernel of the same entry point."""
    from pb_sed_amd import ops
    torch.manual_seed(0)
    b, cin, cout, f, t, k, pool, pro = 3, case['cin'], case['cout'], case['f'], case['t'], case['k'], case['pool'], case['pro']
    x = torch.randn(b, cin, f, t, dtype=torch.float64)
    w = (torch.randn(cout, cin, *k, dtype=torch.float64) / np.sqrt(cin * k[0] * k[1])).requires_grad_()
    bias = torch.randn(cout, dtype=torch.float64).requires_grad_()
    seq = np.array([t, max(t - 9, 1), max(t // 2, 1)])
    scale = (torch.rand(cin, dtype=torch.float64) + .5) if pro else None
    shift = torch.randn(cin, dtype=torch.float64) * .3 if pro else None
    dx = lambda a: None if a is None else a.float().to(DEV).contiguous()
    seq_dev = torch.as_tensor(seq, dtype=torch.int32).to(DEV)
    wd = w.detach().float().to(DEV)
    pc = ops.PackedConv(wd)
    xd = dx(x)
    y_full = _conv_ref(x, w, bias, scale, shift, seq, k, False)          # un-pooled conv output
    idx = None
    if pool:
        # the gradient is routed through the argmax the (fp32) forward pass took: a near-tie decided differently in
        # float64 would move one position's gradient to the other row (2 % of max |dW| in this size) without either
        # weight-gradient kernel being wrong
        _, idx, _ = ops.conv_fwd(xd, pc, pc.fwd(), bias=dx(bias.detach()), scale=dx(scale), shift=dx(shift), relu=True,
                                 seq_len=seq_dev, pool=True)
        gy = torch.randn(b, cout, f // 2, t, dtype=torch.float64)
        rows = torch.arange(f)[None, None, :, None]
        g_full = gy.repeat_interleave(2, dim=2) * (idx.cpu().long().repeat_interleave(2, dim=2) == rows % 2)
    else:
        gy = torch.randn_like(y_full)
        g_full = gy
    y_full.backward(g_full)
    out = {}
    for precision in ('bf16', 'f32'):
        dw, db = torch.zeros_like(wd), torch.zeros(cout, device=DEV)
        ops.conv_bwd_weight(xd, dx(gy), pc, dw, db, scale=dx(scale), shift=dx(shift), relu=True, seq_len=seq_dev,
                            unpool_idx=idx, precision=precision)
        out[precision] = (dw, db)
    scale_w = w.grad.abs().max().item()
    err = (out['bf16'][0].cpu().double() - w.grad).abs().max().item() / scale_w
    err32 = (out['f32'][0].cpu().double() - w.grad).abs().max().item() / scale_w
    assert err32 < 1e-3 and err < 2e-2, (err, err32)                 # bf16 operands: 2^-9 relative rounding, sqrt(N) growth
    l2 = ((out['bf16'][0].cpu().double() - w.grad).norm() / w.grad.norm()).item()
    assert l2 < 6e-3, l2
    close(out['bf16'][1], bias.grad, atol=.25, rtol=2e-2, name='conv_bgrad bf16')      # sums of bf16-rounded dY
    # THE bf16 gate: rounded-operand restatement (R(dY) x R(relu(fma(x)) mask), bias gradient = sum R(dY))
    x32, wq = x.float().double(), w.detach().float().double().requires_grad_()
    bq = bias.detach().float().double().requires_grad_()
    yq = _conv_ref(x32, wq, bq, None if scale is None else scale.float().double(), None if shift is None else shift.float().double(),
                   seq, k, False, rounded=True)
    yq.backward(g_full.float().double())
    errq = (out['bf16'][0].cpu().double() - wq.grad).abs().max().item() / wq.grad.abs().max().item()
    assert errq < 5e-5, errq
    close(out['bf16'][1], bq.grad, atol=2e-4, rtol=2e-5, name='conv_bgrad bf16 vs rounded operands')


@pytest.mark.parametrize('precision', ['f32', 'bf16'])
@pytest.mark.parametrize('t,b,g,k', [(23, 5, 192, 24), (30, 32, 768, 256), (17, 19, 384, 128), (9, 3, 12, 8), (500, 32, 768, 512)])
def test_gru_wgrad_vs_torch(t, b, g, k, precision):
    """Batched time-major weight/bias gradient GEMM (with the h_{t-1} / h_{t+1} row shift) vs fp64 einsum.  'f32' = exact
    bf16x3 operand splits on the bf16 MFMA: held to the fp32 tolerance; 'bf16': operands rounded to bf16 (2^-9 each)."""
    from pb_sed_amd import ops
    torch.manual_seed(7)
    shifts = [0, -1, 1]
    ks = [k, k, 2 * k]                                  # GEMMs of different input width share the launch
    dg = [torch.randn(t, b, g) for _ in shifts]
    x = [torch.randn(t, b, kk) for kk in ks]
    dw0 = [torch.randn(g, kk) for kk in ks]
    db0 = [torch.randn(g) for _ in shifts]
    dw = [w.to(DEV) for w in dw0]
    db = [v.to(DEV) for v in db0[:2]] + [None]
    ops.gru_wgrad([d.to(DEV) for d in dg], [v.to(DEV) for v in x], shifts, dw, db, precision=precision)
    tol = 2e-5 if precision == 'f32' else 2e-2
    for i, sh in enumerate(shifts):
        xs = torch.zeros(t, b, ks[i], dtype=torch.float64)
        if sh == 0:
            xs[:] = x[i]
        elif sh < 0:
            xs[1:] = x[i][:-1]
        else:
            xs[:-1] = x[i][1:]
        ref = dw0[i].double() + torch.einsum('tbg,tbk->gk', dg[i].double(), xs)
        close(dw[i], ref.float(), atol=tol * (t * b) ** .5, rtol=1e-5, name=f'gru_wgrad dW shift {sh}')
        if precision == 'bf16':                        # THE bf16 gate: both operands rounded to bf16, fp32 accumulation
            refq = dw0[i].double() + torch.einsum('tbg,tbk->gk', _rbf(dg[i]), _rbf(xs))
            close(dw[i], refq.float(), atol=2e-5 * (t * b) ** .5, rtol=1e-5, name=f'gru_wgrad dW shift {sh} vs rounded operands')
        if db[i] is not None:
            close(db[i], (db0[i].double() + dg[i].double().sum((0, 1))).float(), atol=2e-5 * (t * b) ** .5, rtol=1e-5,
                  name=f'gru_wgrad db shift {sh}')


@pytest.mark.parametrize('precision', ['f32', 'bf16'])
@pytest.mark.parametrize('t,b,n,ks', [(23, 5, 192, [24]), (500, 32, 768, [256]), (40, 19, 256, [768, 768]), (9, 3, 12, [8, 36, 4]),
                                      (130, 7, 768, [512]), (17, 4, 100, [260])])
def test_tm_gemm_vs_torch(t, b, n, ks, precision):
    """Time-major projection sum_i x_i @ w_i^T + bias against fp64 (row / output / k tails, several sources).  'f32' = exact
    bf16x3 operand splits: fp32 tolerance; 'bf16': operands rounded to bf16."""
    from pb_sed_amd import ops
    torch.manual_seed(11)
    xs = [torch.randn(t, b, k) for k in ks]
    ws = [torch.randn(n, k) * .2 for k in ks]
    bias = torch.randn(n)
    ref = bias.double() + sum(x.double() @ w.double().T for x, w in zip(xs, ws))
    out = ops.tm_gemm([x.to(DEV) for x in xs], [w.to(DEV) for w in ws], bias.to(DEV), precision=precision)
    scale = sum(ks) ** .5 * .2
    close(out, ref.float(), atol=(3e-6 if precision == 'f32' else 4e-2) * scale, rtol=1e-5, name=f'tm_gemm {precision}')
    if precision == 'bf16':                            # THE bf16 gate: rounded operands, fp32 accumulation
        refq = bias.double() + sum(_rbf(x) @ _rbf(w).T for x, w in zip(xs, ws))
        close(out, refq.float(), atol=3e-6 * scale, rtol=1e-5, name='tm_gemm bf16 vs rounded operands')
    out0 = ops.tm_gemm([xs[0].to(DEV)], [ws[0].to(DEV)], None, precision=precision)
    close(out0, (xs[0].double() @ ws[0].double().T).float(), atol=(3e-6 if precision == 'f32' else 4e-2) * ks[0] ** .5 * .2, rtol=1e-5,
          name='tm_gemm without bias')


WINO_CASES = [c for c in CONV_CASES if c['k'] == (3, 3) and c['cin'] >= 16] + [
    dict(cin=64, cout=64, f=8, t=70, k=(3, 3), pool=False, pro=False),
    dict(cin=32, cout=64, f=6, t=500, k=(3, 3), pool=True, pro=True),
    dict(cin=40, cout=72, f=5, t=131, k=(3, 3), pool=False, pro=False),
    # wider layers (two to four 64-cout tiles per spatial tile); odd row counts, a ragged last column
    dict(cin=64, cout=128, f=6, t=132, k=(3, 3), pool=True, pro=True),
    dict(cin=128, cout=256, f=5, t=100, k=(3, 3), pool=False, pro=True),
    dict(cin=128, cout=128, f=8, t=64, k=(3, 3), pool=True, pro=False),
    dict(cin=128, cout=64, f=3, t=36, k=(3, 3), pool=False, pro=False),
]


@pytest.mark.parametrize('prec', ['wino', 'winox3'])
@pytest.mark.parametrize('case', WINO_CASES, ids=lambda c: f"{c['cin']}x{c['cout']}f{c['f']}t{c['t']}p{int(c['pool'])}{'pro' if c['pro'] else ''}")
def test_conv_winograd_vs_torch(case, prec):
    """3x3 conv with the time axis in the Winograd F(4,3) domain: forward (prologue, bias, pool + argmax, statistics)
    and data gradient (plain and through the pool argmax) at the fp32 tolerances of the direct kernels."""
    from pb_sed_amd import ops
    torch.manual_seed(0)
    b, cin, cout, f, t, k, pool, pro = 3, case['cin'], case['cout'], case['f'], case['t'], case['k'], case['pool'], case['pro']
    x = torch.randn(b, cin, f, t, dtype=torch.float64)
    w = (torch.randn(cout, cin, *k, dtype=torch.float64) / np.sqrt(cin * 9))
    bias = torch.randn(cout, dtype=torch.float64)
    seq = np.array([t, max(t - 9, 1), max(t // 2, 1)])
    scale = (torch.rand(cin, dtype=torch.float64) + .5) if pro else None
    shift = torch.randn(cin, dtype=torch.float64) * .3 if pro else None
    xr = x.clone().requires_grad_()
    y_ref = _conv_ref(xr, w, bias, scale, shift, seq, k, pool)
    gy = torch.randn_like(y_ref)
    y_ref.backward(gy)
    dx = lambda a: None if a is None else a.float().to(DEV).contiguous()
    seq_dev = torch.as_tensor(seq, dtype=torch.int32).to(DEV)
    pc = ops.PackedConv(dx(w))
    xd = dx(x)
    y, idx, stats = ops.conv_fwd(xd, pc, pc.fwd(prec), bias=dx(bias), scale=dx(scale), shift=dx(shift), relu=True,
                                 seq_len=seq_dev, pool=pool, want_stats=True, precision=prec)
    close(y, y_ref, name='conv_fwd wino')
    m = (torch.arange(t)[None] < torch.as_tensor(seq)[:, None]).double()[:, None, None, :]
    yd = y_ref.detach()
    close(stats.sum(0)[:, 0], (yd * m).sum((0, 2, 3)), atol=1e-3, rtol=1e-4, name='stats_sum wino')
    close(stats.sum(0)[:, 1], (yd * yd * m).sum((0, 2, 3)), atol=1e-3, rtol=1e-4, name='stats_sumsq wino')
    if pool:        # same argmax as the direct kernel (ties aside) -> same un-pooling in the backward kernels
        _, idx_d, _ = ops.conv_fwd(xd, pc, pc.fwd(), bias=dx(bias), scale=dx(scale), shift=dx(shift), relu=True,
                                   seq_len=seq_dev, pool=True)
        valid = (torch.arange(t, device=DEV)[None] < seq_dev[:, None])[:, None, None, :]     # masked frames tie exactly
        assert ((idx != idx_d) & valid).float().mean().item() < 2e-3      # near-ties of the pooled rows may resolve differently
    if not pro:
        g, _ = ops.conv_bwd_data(dx(gy), pc, pc.dgrad(prec), xd.shape, idx, None, precision=prec)
        close(g, xr.grad, atol=2e-4, rtol=1e-3, name='conv_dgrad wino')


S16_CASES = [
    # cin, cout, F, T, B, pool, prologue
    (16, 16, 8, 132, 3, True, True),         # the 16->16 + pool layer; tiles with partial rows / columns
    (16, 32, 6, 64, 2, False, True),         # 16->32: two cout tiles per block (2 x 2 waves), F not a multiple of 4
    (11, 16, 4, 72, 3, False, False),        # the tag-conditioned first layer: 11 input channels, no prologue
    (16, 16, 128, 500, 2, True, True),       # real layer size: many tiles per persistent block
    (5, 7, 3, 4, 2, False, True),            # one quad of frames, tiny channel counts, F = 3
    (16, 24, 10, 200, 9, True, False),       # cout off the 16 / 32 tiles, 9 clips
    (16, 16, 4, 133, 2, True, True),         # rows that are not 16-byte aligned: ops falls back to the direct kernel
]


@pytest.mark.parametrize('cin,cout,f,t,b,pool,pro', S16_CASES)
def test_conv_few_channel_bf16x3_vs_torch(cin, cout, f, t, b, pool, pro):
    """csrc/conv_s16.hip: 3x3 conv over <= 16 channels on the bf16 MFMA (two taps x 16 channels per K = 32 step, exact
    three-way operand splits): forward (prologue, bias, pool + argmax, statistics) and data gradient (plain, through the pool
    argmax, and with the BN-ReLU-backward epilogue) at the fp32 tolerances of the direct kernel, and against that kernel."""
    from pb_sed_amd import ops
    torch.manual_seed(1)
    k = (3, 3)
    x = torch.randn(b, cin, f, t, dtype=torch.float64)
    w = (torch.randn(cout, cin, *k, dtype=torch.float64) / np.sqrt(cin * 9))
    bias = torch.randn(cout, dtype=torch.float64)
    seq = np.maximum(np.array([t] + [max(t - 7 * i, 1) for i in range(1, b)]), 1)
    scale = (torch.rand(cin, dtype=torch.float64) + .5) if pro else None
    shift = torch.randn(cin, dtype=torch.float64) * .3 if pro else None
    xr = x.clone().requires_grad_()
    y_ref = _conv_ref(xr, w, bias, scale, shift, seq, k, pool)
    gy = torch.randn_like(y_ref)
    y_ref.backward(gy)
    dx = lambda a: None if a is None else a.float().to(DEV).contiguous()
    seq_dev = torch.as_tensor(seq, dtype=torch.int32).to(DEV)
    pc = ops.PackedConv(dx(w))
    xd = dx(x)
    y, idx, stats = ops.conv_fwd(xd, pc, pc.fwd('s16x3'), bias=dx(bias), scale=dx(scale), shift=dx(shift), relu=True,
                                 seq_len=seq_dev, pool=pool, want_stats=True, precision='s16x3')
    close(y, y_ref, name='conv_fwd s16x3')
    m = (torch.arange(t)[None] < torch.as_tensor(seq)[:, None]).double()[:, None, None, :]
    yd = y_ref.detach()
    close(stats.sum(0)[:, 0], (yd * m).sum((0, 2, 3)), atol=1e-3, rtol=1e-4, name='stats_sum s16x3')
    close(stats.sum(0)[:, 1], (yd * yd * m).sum((0, 2, 3)), atol=1e-3, rtol=1e-4, name='stats_sumsq s16x3')
    yd_, idx_d, _ = ops.conv_fwd(xd, pc, pc.fwd(), bias=dx(bias), scale=dx(scale), shift=dx(shift), relu=True,
                                 seq_len=seq_dev, pool=pool)
    assert (y - yd_).abs().max().item() < 2e-5 * max(yd_.abs().max().item(), 1.)          # the direct fp32 kernel
    if pool:
        valid = (torch.arange(t, device=DEV)[None] < seq_dev[:, None])[:, None, None, :]
        assert ((idx != idx_d) & valid).float().mean().item() < 2e-3      # near-ties of the pooled rows may resolve differently
    # data gradient of a layer that CONTRACTS <= 16 channels: roles swapped (this layer's cout must be <= 16)
    if cout <= 16:
        if not pro:
            g, _ = ops.conv_bwd_data(dx(gy), pc, pc.dgrad('s16x3'), xd.shape, idx, None, precision='s16x3')
            close(g, xr.grad, atol=2e-4, rtol=1e-3, name='conv_dgrad s16x3')
        else:
            # BN-ReLU-backward epilogue: dz = conv^T(unpool(gy)) * relu' * mask and its (sum dz, sum dz xhat) statistics,
            # against the direct kernel's epilogue on the same inputs
            mean, invstd = dx(torch.randn(cin, dtype=torch.float64) * .1), dx(torch.rand(cin, dtype=torch.float64) + .5)
            bn = (xd, mean, invstd, dx(scale), dx(shift))
            g1, st1 = ops.conv_bwd_data(dx(gy), pc, pc.dgrad('s16x3'), xd.shape, idx, seq_dev, bn=bn, precision='s16x3')
            g0, st0 = ops.conv_bwd_data(dx(gy), pc, pc.dgrad(), xd.shape, idx, seq_dev, bn=bn)
            close(g1, g0.double().cpu(), atol=2e-4, rtol=1e-3, name='conv_dgrad + BN-ReLU backward s16x3')
            close(st1.sum(0), st0.sum(0).double().cpu(), atol=2e-3, rtol=1e-3, name='BN backward sums s16x3')


C1X3_CASES = [
    # cin, cout, kw, T, B, prologue
    (256, 256, 3, 500, 3, True),          # the CNN1d shape: 4 time tiles per clip, the last one partial
    (2048, 256, 1, 130, 2, True),         # 64 chunks of one tap
    (96, 128, 3, 77, 5, False),           # odd T, one partial tile
    (40, 100, 1, 129, 9, True),           # channel counts off the 32 / 128 tiles, 9 clips > 8 XCD slots
    (64, 384, 3, 256, 2, True),           # three cout tiles, exact time tiles
    (32, 96, 3, 1, 2, False),             # a single frame
]


@pytest.mark.parametrize('cin,cout,kw,t,b,pro', C1X3_CASES)
def test_conv1d_producer_consumer_x3_vs_torch(cin, cout, kw, t, b, pro):
    """csrc/conv1d_pc.hip (precision 'c1x3'): forward with prologue, bias and masked statistics, plain data gradient, and the
    data gradient with the fused BN-ReLU-mask backward epilogue against the direct fp32 kernel."""
    from pb_sed_amd import ops
    torch.manual_seed(5)
    x = torch.randn(b, cin, 1, t, dtype=torch.float64)
    w = torch.randn(cout, cin, 1, kw, dtype=torch.float64) / np.sqrt(cin * kw)
    bias = torch.randn(cout, dtype=torch.float64)
    seq = np.array([max(t - 7 * i, 1) for i in range(b)])
    scale = (torch.rand(cin, dtype=torch.float64) + .5) if pro else None
    shift = torch.randn(cin, dtype=torch.float64) * .3 if pro else None
    xr = x.clone().requires_grad_()
    y_ref = _conv_ref(xr, w, bias, scale, shift, seq, (1, kw), False)
    gy = torch.randn_like(y_ref)
    y_ref.backward(gy)
    dx = lambda a: None if a is None else a.float().to(DEV).contiguous()
    seq_dev = torch.as_tensor(seq, dtype=torch.int32).to(DEV)
    pc = ops.PackedConv(dx(w[:, :, 0]))
    xd = dx(x[:, :, 0])
    y, _, stats = ops.conv_fwd(xd, pc, pc.fwd('c1x3'), bias=dx(bias), scale=dx(scale), shift=dx(shift), relu=True,
                               seq_len=seq_dev, want_stats=True, precision='c1x3')
    close(y, y_ref[:, :, 0], atol=1e-4, rtol=1e-4, name='conv1d fwd c1x3')
    m = (torch.arange(t)[None] < torch.as_tensor(seq)[:, None]).to(torch.float64)[:, None]
    ym = y_ref[:, :, 0].detach() * m
    close(stats.sum(0), torch.stack([ym.sum((0, 2)), (ym * ym).sum((0, 2))], 1), atol=2e-3, rtol=1e-4, name='conv1d statistics c1x3')
    if not pro:
        g, _ = ops.conv_bwd_data(dx(gy[:, :, 0]), pc, pc.dgrad('c1x3'), xd.shape, None, None, precision='c1x3')
        close(g, xr.grad[:, :, 0], atol=1e-4, rtol=1e-4, name='conv1d dgrad c1x3')
    # fused BN-ReLU-mask backward epilogue: same dz and (sum dz, sum dz * xhat) as the direct kernel
    mean, invstd = torch.randn(cin, device=DEV) * .1, torch.rand(cin, device=DEV) + .5
    gamma, beta = torch.rand(cin, device=DEV) + .5, torch.randn(cin, device=DEV) * .3
    bsc, bsh = gamma * invstd, beta - mean * gamma * invstd
    gd = dx(gy[:, :, 0])
    dz_r, st_r = ops.conv_bwd_data(gd, pc, pc.dgrad('f32'), xd.shape, None, seq_dev, bn=(xd, mean, invstd, bsc, bsh), precision='f32')
    dz, st = ops.conv_bwd_data(gd, pc, pc.dgrad('c1x3'), xd.shape, None, seq_dev, bn=(xd, mean, invstd, bsc, bsh), precision='c1x3')
    close(dz, dz_r, atol=2e-5, rtol=1e-4, name='conv1d dz c1x3 vs direct')
    close(st.sum(0), st_r.sum(0), atol=2e-3, rtol=1e-4, name='conv1d bn-backward sums c1x3 vs direct')


@pytest.mark.parametrize('cin,cout,f,t,pool', [(64, 64, 8, 150, True), (32, 96, 6, 65, False), (128, 64, 4, 500, True),
                                              (32, 48, 6, 64, True), (32, 32, 8, 132, False),      # <= 32 channels produced: 32-cout blocks
                                              (128, 256, 5, 68, False), (256, 128, 6, 100, True)])      # two / four cout tiles
def test_conv_winograd_dgrad_bn_epilogue_matches_direct(cin, cout, f, t, pool):
    """Data gradient with the fused BN-ReLU-mask backward epilogue (and un-pooling): Winograd vs direct kernel on
    the same inputs - dz, and the (sum dz, sum dz*xhat) statistics BN backward needs."""
    from pb_sed_amd import ops
    torch.manual_seed(3)
    b = 3
    fo = f // 2 if pool else f
    x = torch.randn(b, cin, f, t, device=DEV)                       # the layer's raw (pre-BN) forward input
    w = torch.randn(cout, cin, 3, 3, device=DEV) / (cin * 9) ** .5
    g = torch.randn(b, cout, fo, t, device=DEV)
    idx = torch.randint(0, 2, (b, cout, fo, t), device=DEV, dtype=torch.uint8) if pool else None
    seq = torch.tensor([t, max(t - 9, 1), max(t // 2, 1)], dtype=torch.int32, device=DEV)
    mean, invstd = torch.randn(cin, device=DEV) * .1, torch.rand(cin, device=DEV) + .5
    gamma, beta = torch.rand(cin, device=DEV) + .5, torch.randn(cin, device=DEV) * .3
    scale, shift = gamma * invstd, beta - mean * gamma * invstd
    pc = ops.PackedConv(w)
    out = {}
    for prec in ('f32', 'wino', 'winox3'):
        dz, st = ops.conv_bwd_data(g, pc, pc.dgrad(prec), x.shape, idx, seq, bn=(x, mean, invstd, scale, shift),
                                   precision=prec)
        out[prec] = (dz, st.sum(0))
    for prec in ('wino', 'winox3'):
        close(out[prec][0], out['f32'][0], atol=2e-5, rtol=1e-4, name=f'dz {prec} vs direct')
        close(out[prec][1], out['f32'][1], atol=2e-3, rtol=1e-4, name=f'bn-backward sums {prec} vs direct')


def test_logmel_augmentation_vs_oracle():
    """Training-only front-end augmentation (noise + time mask + frequency mask + sequence mask) against the oracle's
    restatement on the same draws; and the sampler keeps to the reference configuration's bounds."""
    from oracle import frontend as ofe
    from pb_sed_amd import modules, ops
    torch.manual_seed(5)
    b, f, t = 6, 128, 200
    y = torch.randn(b, 1, f, t)
    seq = np.array([200, 190, 150, 100, 64, 10])
    fe = modules.NormalizedLogMelExtractor(n_time_masks=1, n_frequency_masks=1, max_noise_scale=.2, augmentation_seed=3)
    for _ in range(20):
        masks, scales = fe.sample_augmentation(seq)
        assert ((masks[:, 1] - masks[:, 0]) <= np.minimum(70, np.floor(.2 * seq))).all() and (masks[:, 1] <= seq).all()
        assert ((masks[:, 3] - masks[:, 2]) <= 20).all() and (masks[:, 3] <= f).all() and (masks[:, [0, 2]] >= 0).all()
        assert (scales >= 0).all() and (scales <= .2).all()
    noise = torch.randn(b, 1, f, t)
    ref = ofe.augment(y, seq, masks, noise, scales)
    out = ops.augment_logmel(y.to(DEV).contiguous(), torch.from_numpy(masks).to(DEV),
                             torch.as_tensor(seq, dtype=torch.int32).to(DEV), noise.to(DEV), torch.from_numpy(scales).to(DEV))
    close(out, ref, atol=1e-6, rtol=1e-6, name='augment_logmel')
    out2 = ops.augment_logmel(y.to(DEV).contiguous(), torch.from_numpy(masks).to(DEV),
                              torch.as_tensor(seq, dtype=torch.int32).to(DEV))
    close(out2, ofe.augment(y, seq, masks), atol=0, rtol=0, name='augment_logmel masks only')


@pytest.mark.parametrize('name', ['a', 'b', 'c', 'd'])
def test_bicrnn_review_buffers_vs_reference_golden(golden, name):
    """strong_label.CRNN.review (pb_sed/models/strong_label/crnn.py:95-138) executed by the reference on seeded inputs:
    loss, strong_label_rate and the y_strong / targets_strong validation buffers (segment-wise maxima over
    eval_segment_length = 1, 3, 4 frames of the strongly labelled clips)."""
    from pb_sed_amd.models import strong_label
    g = golden('ref_bicrnn_loss.npz')
    seg = int(g[f'{name}/eval_segment_length'])
    model = strong_label.CRNN(None, None, None, tag_conditioning=True, eval_segment_length=seg)
    y = torch.as_tensor(g[f'{name}/y']).to(DEV).requires_grad_(True)
    st = torch.as_tensor(g[f'{name}/strong_targets']).to(DEV)
    seq = g[f'{name}/seq_len']
    b, k, t = y.shape
    review = model.review({'seq_len': seq.tolist()}, (y, seq, torch.zeros(b, 1, 4, t, device=DEV), seq, (None, st)))
    assert review['loss'].item() == pytest.approx(float(g[f'{name}/loss']), rel=2e-5)
    assert review['scalars']['strong_label_rate'] == pytest.approx(float(g[f'{name}/strong_label_rate']), abs=1e-7)
    np.testing.assert_array_equal(review['buffers']['y_strong'], g[f'{name}/y_strong'])
    np.testing.assert_array_equal(review['buffers']['targets_strong'], g[f'{name}/targets_strong'])
    review['loss'].backward()
    close(y.grad, g[f'{name}/grad_y'], atol=1e-7, rtol=2e-4, name='dL/dy')
    # summary path: modify_summary turns the buffers into *_strong metrics (needs >= 1 strongly labelled clip)
    summary = dict(scalars={k_: [v] for k_, v in review['scalars'].items()}, images={},
                   buffers={k_: [v] for k_, v in review['buffers'].items()})
    model.modify_summary(summary)
    assert 'macro_fscore_strong' in summary['scalars'] and summary['scalars']['num_examples_strong'] == len(g[f'{name}/y_strong'])


@pytest.mark.parametrize('b,h,t,nl', [(7, 128, 130, 2), (32, 256, 500, 1), (40, 64, 33, 2)])
def test_gru_scans_with_bf16_operands_stay_close_to_the_fp32_scans(b, h, t, nl):
    """pbsed_gru_stack_{fwd,bwd}_granule_bf16 (plain bf16 operands of the recurrent / projection products, the bf16 training
    mode): states and BPTT gradients against the fp32-class scans on the same inputs - bf16 rounding of W and h (2^-9 each)
    through T dependent steps; the tolerance is that of the bf16 convolutions."""
    from pb_sed_amd import ops
    torch.manual_seed(9)
    nch = 2
    seq = torch.as_tensor(np.sort(np.random.RandomState(2).randint(t // 2, t + 1, b))[::-1].copy(), dtype=torch.int32).to(DEV)
    gi0 = [torch.randn(t, b, 3 * h, device=DEV) * .5 for _ in range(nch)]
    mk = lambda *s: torch.randn(*s, device=DEV) * h ** -.5
    idx = [(c, l) for c in range(nch) for l in range(nl)]
    w_ih = [mk(3 * h, h) if l else None for c, l in idx]
    b_ih = [mk(3 * h) if l else None for c, l in idx]
    w_hh = [mk(3 * h, h) for _ in idx]
    b_hh = [mk(3 * h) for _ in idx]
    dy = [torch.randn(t, b, h, device=DEV) * (torch.arange(t, device=DEV)[:, None, None] < seq[None, :, None]) for _ in range(nch)]
    out = {}
    for prec in ('f32', 'bf16'):
        hs, save = ops.gru_stack_fwd(gi0, w_ih, b_ih, w_hh, b_hh, [False, True], seq, nl, save=True, precision=prec)
        w_hh_t = [ops.transpose2d(w) for w in w_hh]
        w_up = [ops.transpose2d(w_ih[c * nl + l + 1]) if l + 1 < nl else None for c, l in idx]
        dgi, dgh = ops.gru_stack_bwd(w_hh_t, w_up, hs, save, dy, [False, True], seq, nl, precision=prec)
        ops.check_gru_sync()
        out[prec] = (hs, dgi, dgh)
    for a, r in zip(out['bf16'][0], out['f32'][0]):
        assert (a - r).abs().max().item() < 3e-2
    for k in (1, 2):
        for a, r in zip(out['bf16'][k], out['f32'][k]):
            assert ((a - r).norm() / r.norm().clamp_min(1e-6)).item() < 5e-2
    assert (out['bf16'][0][0] - out['f32'][0][0]).abs().max().item() > 0      # it is the other kernel
    # THE bf16 gate: the bf16-operand oracle scan (oracle/bf16emu.py: bf16(h_{t-1}) bf16(W_hh)^T in the recurrence, the layer
    # boundary projection and every BPTT product from ITS rounded operands) in float64 on the same inputs.  A rounding of h
    # that falls the other way (fp32 vs float64 pre-rounding values) perturbs later steps - the recurrence is contractive, so
    # the states agree to ~1e-4 where the fp32 scans are 3e-2 away.
    if t <= 130:
        from oracle import bf16emu
        seq_c = seq.cpu().numpy()
        for c in range(nch):
            rev = bool(c)
            x_in = gi0[c].cpu().double().transpose(0, 1)          # [B,T,3H] = W_ih x + b_ih of layer 0
            m = (torch.arange(t)[None] < torch.as_tensor(seq_c)[:, None]).double()
            from oracle.nn import reverse_sequence
            leaves, y = [], None
            gi = (reverse_sequence(x_in, seq_c) if rev else x_in).clone().requires_grad_()
            leaves.append(gi)
            cur = gi
            for l in range(nl):
                i = c * nl + l
                if l:
                    cur = bf16emu._LinearBf16.apply(y, w_ih[i].cpu().double()) + b_ih[i].cpu().double()
                y = bf16emu._scan(cur, w_hh[i].cpu().double(), b_hh[i].cpu().double(), m)
            y_out = reverse_sequence(y, seq_c) if rev else y
            got = out['bf16'][0][c * nl + nl - 1].cpu().double().transpose(0, 1)
            e = (got - y_out.detach()).abs().max().item()
            assert e < 1e-3, (c, e)
            (y_out * dy[c].cpu().double().transpose(0, 1)).sum().backward()
            dgi_ref = reverse_sequence(gi.grad, seq_c) if rev else gi.grad          # gradient wrt layer 0's projected input
            dgi_hip = out['bf16'][1][c * nl].cpu().double().transpose(0, 1)
            eg = ((dgi_hip - dgi_ref).norm() / dgi_ref.norm()).item()
            assert eg < 5e-3, (c, eg)


def test_persistent_scan_protocol_under_skewed_timing():
    """A timing stress of the scans' inter-workgroup protocol (the data is the flag: tagged words, polled): the same scans with
    the first-poll delay forced to 0 (every first look comes too early: the retry path runs at every step), to the built-in value
    and to 4x that (consumers late), and with a second stream keeping compute units busy beside them - every run has to produce
    BIT-IDENTICAL states, BPTT gradients and no time-out flag.  (SURVEY.md section 5 lists race tooling as optional; this is the
    check the hand-off protocol can be given without a sanitizer.)"""
    import ctypes as C
    from pb_sed_amd import _lib, ops
    torch.manual_seed(3)
    nch, nl, b, h, t = 2, 2, 32, 256, 160
    seq = torch.as_tensor(np.sort(np.random.RandomState(4).randint(t // 2, t + 1, b))[::-1].copy(), dtype=torch.int32).to(DEV)
    gi0 = [torch.randn(t, b, 3 * h, device=DEV) * .5 for _ in range(nch)]
    mk = lambda *s_: torch.randn(*s_, device=DEV) * h ** -.5
    idx = [(c, l) for c in range(nch) for l in range(nl)]
    w_ih, b_ih = [mk(3 * h, h) if l else None for c, l in idx], [mk(3 * h) if l else None for c, l in idx]
    w_hh, b_hh = [mk(3 * h, h) for _ in idx], [mk(3 * h) for _ in idx]
    w_hh_t = [ops.transpose2d(w) for w in w_hh]
    w_up = [ops.transpose2d(w_ih[c * nl + l + 1]) if l + 1 < nl else None for c, l in idx]
    dy = [torch.randn(t, b, h, device=DEV) * (torch.arange(t, device=DEV)[:, None, None] < seq[None, :, None]) for _ in range(nch)]

    def run():
        hs, save = ops.gru_stack_fwd(gi0, w_ih, b_ih, w_hh, b_hh, [False, True], seq, nl, save=True)
        dgi, dgh = ops.gru_stack_bwd(w_hh_t, w_up, hs, save, dy, [False, True], seq, nl)
        ops.check_gru_sync()
        return [v.clone() for v in hs + dgi + dgh]
    base = run()                                        # (tunes the delays of this shape in place)
    cur = (C.c_int * 4)()
    _lib.call('pbsed_gru_get_poll_delays', 0, cur)
    tuned = list(cur)
    default = ops._POLL_DEFAULT[(torch.cuda.current_device(), 0)]
    saved = dict(ops._POLL_TUNED)
    try:
        for fwd_d, bwd_d, busy in ((0, 0, False), (default[0], default[2], False), (4 * default[0], 4 * default[2], False), (tuned[0], tuned[2], True)):
            for k in [k for k in ops._POLL_TUNED if k[3][:5] == (nch, nl, b, h, t)]:
                ops._POLL_TUNED[k] = fwd_d if k[2] == 0 else bwd_d
            side = torch.cuda.Stream()
            if busy:                                    # something else holds compute units while the scans run
                with torch.cuda.stream(side):
                    junk = torch.randn(2048, 2048, device=DEV)
                    for _ in range(20):
                        junk = junk @ junk * 1e-3
            got = run()
            side.synchronize()
            for i, (a, r) in enumerate(zip(got, base)):
                assert torch.equal(a, r), f'delays ({fwd_d}, {bwd_d}) busy={busy}: tensor {i} differs'
    finally:
        ops._POLL_TUNED.clear()
        ops._POLL_TUNED.update(saved)


def test_scratch_registrations_are_bounded_over_many_streams():
    """ops.ensure_scratch keeps at most SCRATCH_MAX_STREAMS caller-owned scratch buffers per device (least recently
    used first out, registration withdrawn) - a program cycling through streams neither pins 160 MB per stream for ever nor
    fills the library's registration table; weight gradients on every stream stay correct."""
    from pb_sed_amd import ops
    torch.manual_seed(0)
    x = torch.randn(2, 64, 4, 32, device=DEV)
    g = torch.randn(2, 64, 4, 32, device=DEV)
    w = torch.randn(64, 64, 3, 3, device=DEV) * .05
    pc = ops.PackedConv(w)
    ref = torch.zeros_like(w)
    ops.conv_bwd_weight(x, g, pc, ref, None, relu=False)
    torch.cuda.synchronize()
    for i in range(ops.SCRATCH_MAX_STREAMS + 70):            # more streams than the library's 64-entry table
        st = torch.cuda.Stream()
        with torch.cuda.stream(st):
            dw = torch.zeros_like(w)
            ops.conv_bwd_weight(x, g, pc, dw, None, relu=False)
        st.synchronize()
        assert (dw - ref).abs().max().item() <= 2e-4 * ref.abs().max().item(), i
        assert sum(1 for k in ops._SCRATCH if k[0] == torch.cuda.current_device()) <= ops.SCRATCH_MAX_STREAMS


def test_weight_gradients_on_two_streams_with_caller_owned_scratch():
    """include/pbsed.h: the only memory the library would own is the partial-sum scratch of the weight-gradient kernels;
    with pbsed_set_scratch it is the caller's per (device, stream), and two streams of one device run the slotted weight
    gradient of the same layer concurrently with independent results (ops.ensure_scratch registers one buffer per stream)."""
    from pb_sed_amd import ops
    torch.manual_seed(5)
    b, cin, cout, f, t = 4, 16, 16, 32, 500                         # few channels: the slotted (scratch) reduction
    xs = [torch.randn(b, cin, f, t, device=DEV) for _ in range(2)]
    gs = [torch.randn(b, cout, f, t, device=DEV) for _ in range(2)]
    w = torch.randn(cout, cin, 3, 3, device=DEV) * .1
    pc = ops.PackedConv(w)
    ref = []
    for x, g in zip(xs, gs):
        dw, db = torch.zeros_like(w), torch.zeros(cout, device=DEV)
        ops.conv_bwd_weight(x, g, pc, dw, db, relu=False)
        ref.append((dw.clone(), db.clone()))
    torch.cuda.synchronize()
    streams = [torch.cuda.Stream(), torch.cuda.Stream()]
    out = [(torch.zeros_like(w), torch.zeros(cout, device=DEV)) for _ in range(2)]
    for rep in range(8):                                            # interleaved enqueues on both streams
        for k, st in enumerate(streams):
            with torch.cuda.stream(st):
                out[k][0].zero_(), out[k][1].zero_()
                ops.conv_bwd_weight(xs[k], gs[k], pc, out[k][0], out[k][1], relu=False)
    torch.cuda.synchronize()
    assert len({key for key in ops._SCRATCH if key[1] in [st.cuda_stream for st in streams]}) == 2
    for k in range(2):
        close(out[k][0], ref[k][0], atol=2e-3, rtol=1e-4, name=f'dw stream {k}')
        close(out[k][1], ref[k][1], atol=2e-3, rtol=1e-4, name=f'db stream {k}')


# ------------------------------------------------------------------------------------------ BN backward in the dY loaders
@pytest.mark.parametrize('cin,cout,f,t,b,pool,per_cf,pro', [
    (64, 64, 8, 96, 3, False, False, True),          # conv_wgrad_pc_kernel<2,2>, per-channel coefficients
    (64, 128, 8, 100, 3, True, False, True),         # ... through a (2,1) pool (dz / gx are the pooled tensors), two cout tiles
    (128, 256, 4, 64, 2, False, True, True),         # ... coefficients per (channel, row): the last conv2d in front of the conv1d stack
    (32, 32, 8, 200, 3, True, False, True),          # conv_wgrad_pc_kernel<1,2,2,2,1> (time-sliced consumer waves, NB = 4)
    (32, 32, 6, 68, 2, False, False, False),         # ... no prologue on x
    (16, 16, 8, 96, 3, True, False, True),           # conv_wgrad_s16
    (16, 32, 8, 100, 3, False, False, True),
    (11, 16, 8, 64, 2, False, False, False),
    (32, 64, 8, 96, 3, False, False, True),          # conv_wgrad_wino_kernel
    (32, 64, 8, 100, 2, True, False, True),
    (1, 16, 8, 96, 3, False, False, False),          # the direct kernel (first layer)
])
def test_bn_backward_in_the_weight_gradient_loader_matches_the_standalone_pass(cin, cout, f, t, b, pool, per_cf, pro):
    """pbsed_conv_bwd_weight_bng (dY = k1 dz + k2 gx + k3 formed while dY is staged, written out for the data gradient) against
    pbsed_bn_bwd followed by pbsed_conv_bwd_weight on the same tensors: the formed gradient, dW, db, dgamma, dbeta.  Ragged
    sequence lengths (dY is zero beyond them), pooled and un-pooled layers, per-channel and per-(channel, row) norms."""
    from pb_sed_amd import ops
    torch.manual_seed(cin * 7 + cout + t)
    w = torch.randn(cout, cin, 3, 3, device=DEV) * .1
    pc = ops.PackedConv(w)
    if not ops.conv_bwd_weight_bng_supported(pc, cin, f, t, per_cf):
        pytest.skip('no weight-gradient kernel with the BN-backward loader for this shape (the engine takes the stand-alone pass)')
    seq = np.array(([t, max(t - 29, 1), max(t // 2 - 3, 1)] * 2)[:b])
    seq_dev = torch.as_tensor(seq, dtype=torch.int32).to(DEV)
    fo = f // 2 if pool else f
    x = torch.randn(b, cin, f, t, device=DEV)
    gx = torch.randn(b, cout, fo, t, device=DEV) * 1.5 + .3           # the conv's raw output = the next norm's input
    inside = (torch.arange(t, device=DEV)[None] < seq_dev[:, None])[:, None, None, :]
    dz = torch.randn(b, cout, fo, t, device=DEV) * (torch.rand(b, cout, fo, t, device=DEV) > .4) * inside     # ReLU-masked, zero beyond seq
    idx = (torch.rand(b, cout, fo, t, device=DEV) > .5).to(torch.uint8) if pool else None
    c_n = cout * fo if per_cf else cout                                # channels of the norm
    st = ops.BNState(c_n, DEV)
    st.mean.normal_(0, .3), st.invstd.uniform_(.5, 2.), st.scale.uniform_(.3, 1.7), st.shift.normal_(0, .2)
    # the (sum dz, sum dz * xhat) partial sums as the data-gradient epilogue leaves them (spread over the slots)
    xs = gx.reshape(b, c_n, -1, t) if per_cf else gx
    dzs = dz.reshape(b, c_n, -1, t) if per_cf else dz
    xhat = (xs - st.mean[None, :, None, None]) * st.invstd[None, :, None, None]
    stats = torch.zeros(ops.STAT_SLOTS, c_n, 2, device=DEV, dtype=torch.float64)
    stats[3, :, 0] = dzs.double().sum((0, 2, 3))
    stats[5, :, 1] = (dzs.double() * xhat.double()).sum((0, 2, 3))
    count = float(seq.sum() * (1 if per_cf else fo))
    xsc, xsh = (torch.rand(cin, device=DEV) + .5, torch.randn(cin, device=DEV) * .3) if pro else (None, None)

    dgamma_r, dbeta_r = torch.zeros(c_n, device=DEV), torch.zeros(c_n, device=DEV)
    g_ref = ops.bn_backward(dzs.clone().contiguous(), xs.contiguous(), st, stats.clone(), count, dgamma_r, dbeta_r, seq_dev)
    dw_r, db_r = torch.zeros_like(w), torch.zeros(cout, device=DEV)
    ops.conv_bwd_weight(x, g_ref.reshape(gx.shape), pc, dw_r, db_r, scale=xsc, shift=xsh, relu=True, seq_len=seq_dev, unpool_idx=idx)

    dgamma, dbeta = torch.zeros(c_n, device=DEV), torch.zeros(c_n, device=DEV)
    lazy = ops.LazyBNGrad(dzs.clone().contiguous(), xs.contiguous(), st, stats.clone(), count, dgamma, dbeta, seq_dev)
    dw, db = torch.zeros_like(w), torch.zeros(cout, device=DEV)
    g = ops.conv_bwd_weight(x, None, pc, dw, db, scale=xsc, shift=xsh, relu=True, seq_len=seq_dev, unpool_idx=idx, bng=lazy, per_cf=per_cf)
    torch.cuda.synchronize()
    assert torch.equal(dgamma, dgamma_r) and torch.equal(dbeta, dbeta_r)
    scale = g_ref.abs().max().item()
    assert (g.reshape(g_ref.shape) - g_ref).abs().max().item() <= 2e-6 * scale, ((g.reshape(g_ref.shape) - g_ref).abs().max().item(), scale)
    assert (g.reshape(g_ref.shape)[~inside.expand_as(dz).reshape(g_ref.shape)] == 0).all()      # written as zeros beyond the sequences
    close(dw, dw_r, atol=3e-6 * dw_r.abs().max().item(), rtol=0, name='dW')
    close(db, db_r, atol=3e-6 * db_r.abs().max().item(), rtol=0, name='db')
