"""Stand-alone timing of pbsed_bn_bwd (the BN backward apply pass) at the shapes of a C2 step.  Usage (GPU box): python tools/micro/bn_bwd_bench.py"""
import sys, torch, time
sys.path.insert(0,'/root/repo')
from pb_sed_amd import ops, _lib
from pb_sed_amd._lib import call, ptr
dev='cuda:0'
for (B,C,S,T) in ((32,128,16,500),(32,64,32,500),(32,16,128,500),(32,2048,1,500),(32,256,1,500)):
    dz=torch.randn(B,C,S,T,device=dev); x=torch.randn(B,C,S,T,device=dev)
    sums=torch.zeros(32,C,2,dtype=torch.float64,device=dev)
    mean=torch.zeros(C,device=dev); inv=torch.ones(C,device=dev); sc=torch.ones(C,device=dev)
    dg=torch.zeros(C,device=dev); db=torch.zeros(C,device=dev)
    seq=torch.full((B,),T,dtype=torch.int32,device=dev)
    import ctypes
    def run():
        call('pbsed_bn_bwd', ptr(dz), ptr(x), ptr(sums), float(B*S*T), ptr(mean), ptr(inv), ptr(sc), ptr(dg), ptr(db), ptr(seq), B, C, S, T, ops.stream())
    run(); torch.cuda.synchronize()
    e0,e1=torch.cuda.Event(enable_timing=True),torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(20): run()
    e1.record(); torch.cuda.synchronize()
    us=e0.elapsed_time(e1)/20*1e3
    print(f'B{B} C{C} S{S} T{T}: {us:7.1f} us  {dz.numel()*12/us/1e6:6.2f} TB/s')
