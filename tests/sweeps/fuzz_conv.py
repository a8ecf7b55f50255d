"""Randomised shape sweep of the conv entry points (forward + statistics, data gradient with and without the fused
BN-backward epilogue, weight / bias gradient) against torch on the CPU in float64: odd T and F, channel counts that are
not multiples of the tiles, pooled and un-pooled layers, direct and Winograd kernels.  Usage: fuzz_conv.py [cases] [seed]"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np, torch
import torch.nn.functional as F
from pb_sed_amd import ops

DEV = 'cuda'
n_cases = int(sys.argv[1]) if len(sys.argv) > 1 else 40
rng = np.random.default_rng(int(sys.argv[2]) if len(sys.argv) > 2 else 0)


def ref_layer(x, w, bias, scale, shift, seq, k, pool):
    a = x
    if scale is not None:
        a = F.relu(x * scale.view(1, -1, 1, 1) + shift.view(1, -1, 1, 1))
        m = (torch.arange(x.shape[-1])[None] < torch.as_tensor(seq)[:, None]).to(x.dtype)
        a = a * m[:, None, None, :]
    ph, pw = k[0] - 1, k[1] - 1
    a = F.pad(a, (pw // 2, pw - pw // 2, ph // 2, ph - ph // 2))
    y = F.conv2d(a, w, bias)
    idx = None
    if pool:
        y, idx = F.max_pool2d(y, (2, 1), return_indices=True)
    return y


def err(a, b):
    a, b = a.detach().cpu().double(), b.detach().cpu().double()
    return ((a - b).abs().max() / b.abs().max().clamp_min(1e-6)).item()


worst = {}
for case in range(n_cases):
    k = [(3, 3), (3, 3), (1, 1), (1, 3)][rng.integers(4)]
    two_d = k == (3, 3) or (k == (1, 1) and rng.random() < .4)       # 1x1 conv2d layers (net_config 'deep'): rows, pools
    cin = int(rng.choice([1, 3, 8, 11, 16, 24, 32, 40, 64, 96, 128, 200]))
    cout = int(rng.choice([5, 10, 16, 24, 32, 48, 64, 96, 128, 160, 256]))
    f = int(rng.choice([2, 4, 6, 8, 10, 16, 22])) if two_d else 1
    t = int(rng.choice([7, 33, 50, 64, 65, 100, 127, 128, 150, 260, 500]))
    pool = bool(two_d and f % 2 == 0 and rng.random() < .5)
    pro = bool(cin > 1 and rng.random() < .7)
    b = int(rng.integers(1, 4))
    prec = 'wino' if (k == (3, 3) and cin >= 16 and rng.random() < .6) else 'f32'
    if cin >= 16 and rng.random() < .2:
        prec = 'bf16x3'                            # 3-way bf16 split: fp32-class accuracy through the bf16 MFMA kernels
    if k == (3, 3) and cin >= 16 and rng.random() < .4:
        prec = 'winox3'                            # round 3: producer / consumer bf16x3 Winograd (64- and 32-cout blocks; T % 4 != 0
    if not two_d and rng.random() < .5:            # falls back to the fp32 Winograd kernel), producer / consumer Conv1d
        prec = 'c1x3'
    torch.manual_seed(case)
    x = torch.randn(b, cin, f, t, dtype=torch.float64)
    w = (torch.randn(cout, cin, *k, dtype=torch.float64) / np.sqrt(cin * k[0] * k[1])).requires_grad_()
    bias = torch.randn(cout, dtype=torch.float64).requires_grad_()
    seq = np.sort(rng.integers(max(t // 3, 1), t + 1, b))[::-1].copy(); seq[0] = t
    scale = (torch.rand(cin, dtype=torch.float64) + .5) if pro else None
    shift = torch.randn(cin, dtype=torch.float64) * .3 if pro else None
    if os.environ.get('FUZZ_ONLY') and case != int(os.environ['FUZZ_ONLY']):      # same random stream, one case evaluated
        continue
    xr = x.clone().requires_grad_()
    y_ref = ref_layer(xr, w, bias, scale, shift, seq, k, pool)
    gy = torch.randn_like(y_ref)
    y_ref.backward(gy)
    dx = lambda a: None if a is None else a.float().to(DEV).contiguous()
    seq_dev = torch.as_tensor(seq, dtype=torch.int32).to(DEV)
    pc = ops.PackedConv(w.detach().float().to(DEV))
    xd = dx(x) if two_d else dx(x)[:, :, 0]
    gyd = dx(gy) if two_d else dx(gy)[:, :, 0]
    tag = f'case {case}: {cin}->{cout} k{k[0]}x{k[1]} F{f} T{t} B{b} pool={int(pool)} pro={int(pro)} {prec}'
    try:
        y, idx, stats = ops.conv_fwd(xd, pc, pc.fwd(prec), bias=dx(bias.detach()), scale=dx(scale), shift=dx(shift), relu=True,
                                     seq_len=seq_dev, pool=pool, want_stats=True, precision=prec)
        e = {'fwd': err(y.reshape(y_ref.shape), y_ref)}
        m = (torch.arange(t)[None] < torch.as_tensor(seq)[:, None]).double()[:, None, None, :]
        e['stats'] = err(stats.sum(0)[:, 0], (y_ref.detach() * m).sum((0, 2, 3)))
        dw = torch.zeros_like(pc.weight); db = torch.zeros(cout, device=DEV)
        ops.conv_bwd_weight(xd, gyd, pc, dw, db, scale=dx(scale), shift=dx(shift), relu=True, seq_len=seq_dev, unpool_idx=idx)
        w_grad, x_grad = w.grad, xr.grad
        if pool:
            # a pool window whose two rows agree to fp32 rounding may pick the other row here than the float64 reference does
            # (found by this sweep: 40->96 F22 T500, one such window moved the weight gradient by 4e-3): the gradients are then
            # compared for the argmax THIS run took - the reference is re-differentiated through the kernel's own pool indices
            a = xr.detach().clone().requires_grad_()
            w2 = w.detach().clone().requires_grad_()
            pre = ref_layer(a, w2, bias.detach(), scale, shift, seq, k, False)
            sel = idx.cpu().long()                                                    # 0 / 1: row of the window
            ref_sel = (pre[:, :, 1::2] > pre[:, :, 0::2]).long()
            n_flip = int((sel != ref_sel).sum())
            if n_flip:
                e['pool_argmax_flips'] = float(n_flip)
                if os.environ.get('FUZZ_DEBUG'):
                    gap = (pre[:, :, 1::2] - pre[:, :, 0::2]).abs()[sel != ref_sel]
                    where = (sel != ref_sel).nonzero()
                    print('   flipped windows: |row1 - row0| max', gap.max().item(), 'median', gap.median().item(),
                          ' t range', where[:, 3].min().item(), where[:, 3].max().item(), ' seq', seq.tolist(),
                          ' clips', sorted(set(where[:, 0].tolist())))
                picked = torch.where(sel.bool(), pre[:, :, 1::2], pre[:, :, 0::2])
                picked.backward(gy)
                w_grad, x_grad = w2.grad, a.grad
        e['wgrad'] = err(dw, w_grad); e['bgrad'] = err(db, bias.grad)
        if not pro:
            dz, _ = ops.conv_bwd_data(gyd, pc, pc.dgrad(prec), xd.shape, idx, None, precision=prec)
            e['dgrad'] = err(dz.reshape(x.shape), x_grad)
        torch.cuda.synchronize()
    except Exception as ex:                      # an argument error is a finding too
        print(tag, 'EXCEPTION', type(ex).__name__, str(ex)[:120]); continue
    bad = {k_: v for k_, v in e.items() if k_ != 'pool_argmax_flips' and v > (1e-3 if prec == 'bf16x3' else 2e-4)}
    n_bad = globals().get('n_bad', 0) + bool(bad)
    for k_, v in e.items(): worst[k_] = max(worst.get(k_, 0.), v)
    print(tag, {k_: f'{v:.1e}' for k_, v in e.items()}, 'BAD' if bad else '')
print('worst relative deviations:', {k_: f'{v:.1e}' for k_, v in worst.items()}, 'cases over the bar:', globals().get('n_bad', 0))
