"""Debug aid (GPU box): CNN stack alone, HIP engine vs float64 oracle, per-layer gradient errors."""
import copy
import os
import sys

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from oracle import nn as onn, models as om          # noqa: E402
from pb_sed_amd import engine, modules               # noqa: E402

DEV = 'cuda:0'
NET = dict(out_channels_2d=[16, 16, 32], pool_sizes_2d=[1, (2, 1), (2, 1)], kernel_size_2d=3,
           out_channels_1d=[64, 64, 64], kernel_size_1d=[1, 3, 1])


def run(b, f, t, seq, seed=0):
    torch.manual_seed(seed)
    ref = om.build_cnn(1, input_height=f, **NET).double().train()
    with torch.no_grad():
        for n, p in ref.named_parameters():
            if n.endswith('gamma'):
                p.uniform_(.7, 1.3)
            elif n.endswith('beta') or n.endswith('bias'):
                p.normal_(0, .1)
    prod = modules.build_cnn(1, input_height=f, **NET)
    prod.load_state_dict({k: v.float() for k, v in ref.state_dict().items()})
    prod.to(DEV).train()
    x = torch.randn(b, 1, f, t, dtype=torch.float64)
    m = (torch.arange(t)[None] < torch.as_tensor(seq)[:, None]).double()[:, None, None, :]
    x = x * m
    h_ref, _ = ref(x, np.asarray(seq))
    gh = torch.randn_like(h_ref) * m[:, :, 0]
    h_ref.backward(gh)
    fp, fg = engine.flatten_parameters(prod)
    fg.zero_()
    layers = engine.describe_stack([prod.cnn_2d, prod.cnn_1d])
    seq_dev = torch.as_tensor(np.asarray(seq), dtype=torch.int32).to(DEV)
    h, ctx = engine.stack_forward(layers, x.float().to(DEV), seq_dev, np.asarray(seq), True)
    print(f'B={b} F={f} T={t} seq={list(seq)}: fwd max err {(h.cpu().double() - h_ref).abs().max():.2e}')
    engine.stack_backward(layers, ctx, gh.float().to(DEV), seq_dev, np.asarray(seq), False)
    torch.cuda.synchronize()
    refp = dict(ref.named_parameters())
    for n, p in prod.named_parameters():
        q = refp[n].grad
        sc = q.abs().max().item()
        if sc < 1e-7:
            continue
        err = (p.grad.cpu().double() - q).abs().max().item()
        flag = '  <<<<' if err > 2e-3 * sc else ''
        print(f'   {n:36s} rel err {err / sc:.2e}{flag}')


if __name__ == '__main__':
    run(5, 128, 100, [100] * 5)
    run(5, 128, 100, [100, 93, 85, 77, 70])
    run(2, 128, 500, [500, 420])
    run(1, 128, 100, [100])
