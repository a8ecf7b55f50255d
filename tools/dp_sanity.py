"""Two data-parallel ranks on ONE GPU (gloo backend, CUDA tensors): exercises Trainer + GradSync end to end with the HIP
path (bucket hooks fired from autograd's device thread, async all-reduce of flat-gradient slices, averaged Adam).
Both ranks must hold identical parameters after every step.  usage: python tools/dp_sanity.py"""
import os
import sys

import numpy as np
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))


def worker(rank, world, port, q):
    os.environ.update(MASTER_ADDR='127.0.0.1', MASTER_PORT=str(port))
    dist.init_process_group('gloo', rank=rank, world_size=world)
    torch.cuda.set_device(0)
    from pb_sed_amd.models import weak_label
    from pb_sed_amd.trainer import Trainer, shard_batch
    torch.manual_seed(0)
    net = dict(out_channels_2d=[16, 32, 64], pool_sizes_2d=[1, (2, 1), (2, 1)], kernel_size_2d=3,
               out_channels_1d=[64, 64], kernel_size_1d=[3, 1])
    model = weak_label.CRNN.build(num_events=10, hidden_size=64, num_layers=2, net=net).to('cuda:0')
    trainer = Trainer(model, lr=1e-3, gradient_clipping=5.)
    g = torch.Generator().manual_seed(7)
    b, n, t = 8, 32000, 100
    batch = {'audio_data': torch.randn(b, n, generator=g).cuda(), 'seq_len': [t] * b,
             'weak_targets': (torch.rand(b, 10, generator=g) < .3).float().cuda(),
             'boundary_targets': torch.zeros(b, 10, t).cuda()}
    batch['weak_targets'][:, 0] = 1
    mine = shard_batch(batch, rank, world)
    sums = []
    for _ in range(3):
        rev = trainer.step(mine)
        torch.cuda.synchronize()
        sums.append((float(rev['loss'].item()), float(trainer.flat_param.double().sum().item()),
                     float(trainer.flat_param.double().abs().sum().item())))
    q.put((rank, sums))
    dist.destroy_process_group()


if __name__ == '__main__':
    mp.set_start_method('spawn')
    q = mp.Queue()
    port = 29000 + os.getpid() % 1000
    procs = [mp.Process(target=worker, args=(r, 2, port, q)) for r in range(2)]
    [p.start() for p in procs]
    res = dict(q.get(timeout=600) for _ in procs)
    [p.join() for p in procs]
    for step in range(3):
        (l0, s0, a0), (l1, s1, a1) = res[0][step], res[1][step]
        print(f'step {step}: loss rank0 {l0:.6f} rank1 {l1:.6f}; param checksum rank0 {s0:.9f} rank1 {s1:.9f}')
        assert np.isfinite(l0) and np.isfinite(l1)
        assert s0 == s1 and a0 == a1, 'ranks diverged: the gradient all-reduce / averaged Adam is not symmetric'
    print('dp sanity ok: ranks bit-identical after every step')
