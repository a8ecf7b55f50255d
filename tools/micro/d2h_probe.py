"""How long a device -> pinned-host copy of one batch's scores takes, and whether a side stream hides it behind kernels.
Usage (GPU box): python tools/micro/d2h_probe.py"""
import time
import torch

dev = torch.device('cuda:0')
x = torch.randn(64, 3, 10, 500, device=dev)
host = torch.empty(x.shape, dtype=x.dtype, pin_memory=True)
work = torch.randn(4096, 4096, device=dev)


def timed(fn, n=20):
    fn()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(n):
        fn()
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) / n * 1e6


print(f'{x.numel() * 4 / 1e6:.2f} MB')
print(f'pinned, non_blocking, current stream: {timed(lambda: host.copy_(x, non_blocking=True)):8.1f} us')
print(f'pageable .cpu():                      {timed(lambda: x.cpu()):8.1f} us')
print(f'matmul alone:                         {timed(lambda: work @ work):8.1f} us')
side = torch.cuda.Stream()


def both_same():
    work @ work
    host.copy_(x, non_blocking=True)


def both_side():
    ev = torch.cuda.Event()
    ev.record()
    with torch.cuda.stream(side):
        side.wait_event(ev)
        host.copy_(x, non_blocking=True)
    work @ work


print(f'matmul + copy, one stream:            {timed(both_same):8.1f} us')
print(f'matmul + copy on a side stream:       {timed(both_side):8.1f} us')
for mb in (0.25, 1, 4, 16, 64):
    y = torch.empty(int(mb * 250000), device=dev)
    h = torch.empty(y.shape, pin_memory=True)
    us = timed(lambda: h.copy_(y, non_blocking=True))
    print(f'{mb:6.2f} MB pinned D2H: {us:8.1f} us  {mb * 1e6 / us / 1e3:6.2f} GB/s')
