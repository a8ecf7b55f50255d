import sys, torch, os
sys.path.insert(0, '/root/repo' if os.path.exists('/root/repo/pb_sed_amd') else os.getcwd())
from pb_sed_amd import ops
dev='cuda'
def tm(f, n=20):
    for _ in range(3): f()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n): f()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n
b,t=32,500
for cin,cout,kw in [(2048,256,1),(256,2048,1),(256,256,1),(256,256,3)]:
    x=torch.randn(b,cin,t,device=dev); w=torch.randn(cout,cin,kw,device=dev)/(cin*kw)**.5
    pc=ops.PackedConv(w); sc,sh=torch.rand(cin,device=dev)+.5,torch.randn(cin,device=dev)*.3
    mean,invstd=torch.randn(cin,device=dev)*.1,torch.rand(cin,device=dev)+.5
    seq=torch.full((b,),t,dtype=torch.int32,device=dev); g=torch.randn(b,cout,t,device=dev)
    wp=pc.fwd('c1x3'); wd=pc.dgrad('c1x3')
    f0=tm(lambda: ops.conv_fwd(x,pc,wp,precision='c1x3'))
    f1=tm(lambda: ops.conv_fwd(x,pc,wp,scale=sc,shift=sh,seq_len=seq,want_stats=True,precision='c1x3'))
    d0=tm(lambda: ops.conv_bwd_data(g,pc,wd,x.shape,None,None,precision='c1x3'))
    d1=tm(lambda: ops.conv_bwd_data(g,pc,wd,x.shape,None,seq,bn=(x,mean,invstd,sc,sh),precision='c1x3'))
    fl=2*b*cout*cin*kw*t/1e9
    print(f'{cin}->{cout} k{kw}: fwd plain {f0:.3f} fwd pro+stats {f1:.3f} | dgrad plain {d0:.3f} dgrad bn {d1:.3f}   ({fl/f0:.0f} / {fl/d0:.0f} TF plain)')
