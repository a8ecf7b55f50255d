"""Does a small host -> device copy make the host wait for the stream?  Ten ~1 ms matmuls are queued, then the copy is issued and
the host time of that call is measured.  Usage (GPU box): python tools/micro/h2d_block_probe.py"""
import time
import numpy as np
import torch

dev = torch.device('cuda:0')
work = torch.randn(4096, 4096, device=dev)
pinned = torch.arange(64, dtype=torch.int32).pin_memory()
dst = torch.empty(64, dtype=torch.int32, device=dev)
dflags = torch.zeros(64, dtype=torch.int32, device=dev)
hflags = torch.empty(64, dtype=torch.int32, pin_memory=True)


def host_time(fn, label):
    torch.cuda.synchronize()
    for _ in range(10):
        work @ work
    t0 = time.perf_counter()
    fn()
    t1 = time.perf_counter()
    torch.cuda.synchronize()
    t2 = time.perf_counter()
    print(f'{label:62s} host {1e6 * (t1 - t0):9.1f} us   (queue drained {1e3 * (t2 - t1):6.2f} ms later)')


for _ in range(2):
    host_time(lambda: None, 'nothing')
    host_time(lambda: torch.as_tensor(np.arange(64, dtype=np.int32)).to(dev), 'pageable numpy -> .to(dev)')
    host_time(lambda: torch.tensor([1, 2, 3], device=dev), 'torch.tensor(list, device=dev)')
    host_time(lambda: pinned.to(dev, non_blocking=True), 'pinned .to(dev, non_blocking=True)')
    host_time(lambda: dst.copy_(pinned, non_blocking=True), 'dst.copy_(pinned, non_blocking=True)')
    host_time(lambda: pinned.to(dev), 'pinned .to(dev) blocking flag')
    host_time(lambda: hflags.copy_(dflags, non_blocking=True), 'D2H into pinned, non_blocking')
    host_time(lambda: torch.empty(64, dtype=torch.int32, pin_memory=True), 'torch.empty(pin_memory=True)')
    host_time(lambda: torch.empty(3, 1000, 1000, pin_memory=True), 'torch.empty(12 MB, pin_memory=True), new size')
    host_time(lambda: torch.cuda.Event().record(), 'Event().record()')
    host_time(lambda: torch.zeros(64, device=dev), 'torch.zeros on device')
    host_time(lambda: torch.full((64,), 3, device=dev), 'torch.full on device')
