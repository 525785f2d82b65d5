import os, sys, statistics
sys.path.insert(0, '/root/repo/pasta-gan-plusplus_amd')
import torch
from torch_utils import custom_ops
custom_ops.verbosity = 'none'
from torch_utils.ops import conv2d_mfma
dev='cuda'
for (N,H,cin,cout) in [(8,16,512,512),(8,8,512,512),(8,32,512,512),(16,16,512,512)]:
    x=torch.randn(N,cin,H,H,device=dev); w=torch.randn(cout,cin,3,3,device=dev)/(3*cin**0.5)
    kw=dict(in_scale=torch.rand(N,cin,device=dev)+0.5,out_scale=torch.rand(N,cout,device=dev)+0.5,noise=torch.randn(H,H,device=dev),noise_gain=0.1,bias=torch.randn(cout,device=dev),act='lrelu',alpha=0.2,gain=1.4,clamp=256.0)
    res={}
    ref=None
    for f in (0,1,2,3):
        pk=conv2d_mfma.pack_weight(w,winograd=f)
        try:
            run=lambda: conv2d_mfma.conv2d_forward(x,pk,cout,3,3,pad=(1,1),winograd=f,**kw)
            y=run()
        except Exception as e:
            res[f]=f'n/a ({type(e).__name__})'; continue
        if ref is None: ref=y
        ts=[]
        for r in range(6):
            e0,e1=torch.cuda.Event(enable_timing=True),torch.cuda.Event(enable_timing=True)
            e0.record()
            for _ in range(8): run()
            e1.record(); torch.cuda.synchronize()
            if r: ts.append(e0.elapsed_time(e1)/8*1e3)
        res[f]=f'{statistics.median(ts):6.1f} us (|d| {float((y-ref).abs().max()):.1e})'
    print(f'N{N} {H}x{H} {cin}->{cout} mod: direct(+splitK n/a here) {res[0]} | F(2x2) {res[1]} | F(4x4) one-wg {res[2]} | F(4x4) two-wg {res[3]}', flush=True)
