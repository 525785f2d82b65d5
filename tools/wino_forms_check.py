import sys
sys.path.insert(0, '/root/repo/pasta-gan-plusplus_amd')
import torch
from torch_utils import custom_ops
custom_ops.verbosity = 'none'
from torch_utils.ops import conv2d_mfma
dev='cuda'
torch.manual_seed(0)
for (N,H,W,cin,cout) in [(1,32,32,512,512),(1,16,16,512,512),(1,64,64,512,256),(1,32,32,1024,512),(2,32,32,512,512),(1,16,16,64,64),(1,20,32,64,128)]:
    x=torch.randn(N,cin,H,W,device=dev); w=torch.randn(cout,cin,3,3,device=dev)/(3*cin**0.5)
    ref=torch.nn.functional.conv2d(x.double().cpu(), w.double().cpu(), padding=1)
    outs={}
    for f in (0,1,2,3):
        for flip,tr in ((False,False),(True,True)):
            if tr:
                wt=w.transpose(0,1).contiguous()   # IOHW view of the same kernel... pack(transpose_oi) reads w as [Cin',Cout'] so pass the transposed tensor
                pk=conv2d_mfma.pack_weight(wt, flip=flip, transpose_oi=True, winograd=f)
                r=torch.nn.functional.conv2d(x.double().cpu(), w.double().cpu().flip([2,3]), padding=1)
            else:
                pk=conv2d_mfma.pack_weight(w, winograd=f); r=ref
            try:
                y=conv2d_mfma.conv2d_forward(x,pk,cout,3,3,pad=(1,1),winograd=f)
            except Exception as e:
                outs[(f,tr)]=None; continue
            outs[(f,tr)]=(float((y.double().cpu()-r).abs().max()), y)
    s=f'N{N} {H}x{W} {cin}->{cout}: '
    for tr in (False,True):
        s+=('dgrad-pack ' if tr else 'plain ')
        for f in (0,1,2,3):
            o=outs[(f,tr)]
            s+=f'[{f}] '+('n/a ' if o is None else f'{o[0]:.1e} ')
        if outs[(2,tr)] and outs[(3,tr)]:
            s+=f'|3-2| {float((outs[(2,tr)][1]-outs[(3,tr)][1]).abs().max()):.1e}  '
    print(s, flush=True)
