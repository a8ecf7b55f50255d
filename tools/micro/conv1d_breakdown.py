"""Where does a 1-D (k=1 / k=3) conv launch spend its time?  Times the same 256->256, B=32, T=500 layer with and
without the BN/ReLU prologue, the statistics epilogue, and as data gradient (gpurun: python tools/micro/conv1d_breakdown.py)."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
from pb_sed_amd import ops

dev = 'cuda'
B, T = 32, 500
seq = torch.full((B,), T, dtype=torch.int32, device=dev)


def timed(fn, n=50):
    for _ in range(5):
        fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n):
        fn()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n * 1e3


for cin, cout, kw in ((256, 256, 1), (256, 256, 3), (2048, 256, 1), (256, 768, 1), (256, 10, 1)):
    w = torch.randn(cout, cin, kw, device=dev) * 0.05
    pc = ops.PackedConv(w)
    wp, wd = pc._pack(0), pc._pack(1)
    x = torch.randn(B, cin, T, device=dev)
    g = torch.randn(B, cout, T, device=dev)
    sc, sh = torch.rand(cin, device=dev) + 0.5, torch.randn(cin, device=dev)
    bias = torch.randn(cout, device=dev)
    gf = 2 * B * T * cin * cout * kw / 1e9
    rows = [
        ('plain', lambda: ops.conv_fwd(x, pc, wp, bias, seq_len=seq)),
        ('+prologue', lambda: ops.conv_fwd(x, pc, wp, bias, sc, sh, True, seq_len=seq)),
        ('+stats', lambda: ops.conv_fwd(x, pc, wp, bias, seq_len=seq, want_stats=True)),
        ('+both', lambda: ops.conv_fwd(x, pc, wp, bias, sc, sh, True, seq_len=seq, want_stats=True)),
        ('dgrad', lambda: ops.conv_bwd_data(g, pc, wd, x.shape, seq_len=seq)),
        ('dgrad+bn', lambda: ops.conv_bwd_data(g, pc, wd, x.shape, seq_len=seq, bn=(x, sh, sc, sc, sh))),
    ]
    print(f'--- {cin}->{cout} k{kw}: {gf:.2f} GFLOP, ideal {gf / 157.3 * 1e3:.1f} us MFMA, '
          f'{(x.numel() + g.numel()) * 4 / 5e6:.1f} us HBM@5TB/s')
    if cin >= 32 and cout >= 32:
        for prec in ('bf16', 'bf16x3'):
            wpb, wdb = pc._pack_bf16(0, ops.NSPLIT[prec]), pc._pack_bf16(1, ops.NSPLIT[prec])
            rows += [(prec + ' +both', lambda wpb=wpb, prec=prec: ops.conv_fwd(x, pc, wpb, bias, sc, sh, True, seq_len=seq, want_stats=True, precision=prec)),
                     (prec + ' dgrad+bn', lambda wdb=wdb, prec=prec: ops.conv_bwd_data(g, pc, wdb, x.shape, seq_len=seq, bn=(x, sh, sc, sc, sh), precision=prec))]
    for name, fn in rows:
        us = timed(fn)
        print(f'{name:16s} {us:7.1f} us  {gf / us * 1e3:6.1f} TFLOP/s')
