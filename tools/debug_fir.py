import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (ROOT, os.path.join(ROOT, 'pasta-gan-plusplus_amd')):
    sys.path.insert(0, p)
import torch
from torch_utils import custom_ops
custom_ops.verbosity = 'none'
from torch_utils.ops import upfirdn2d
for dt in (torch.float32, torch.bfloat16):
    x = torch.arange(2 * 16 * 5 * 6, dtype=torch.float32).reshape(2, 16, 5, 6).cuda().to(dt) % 97
    xcl = x.contiguous(memory_format=torch.channels_last)
    for fshape, pad in (((1, 1), [0, 0, 0, 0]), ((2, 2), [1, 0, 1, 0]), ((4, 4), [1, 1, 1, 1])):
        for pos in [(0, 0), (fshape[0] - 1, fshape[1] - 1)]:
            f = torch.zeros(fshape).cuda(); f[pos] = 1
            a = upfirdn2d.upfirdn2d(xcl, f, padding=pad)
            b = upfirdn2d.upfirdn2d(x.float(), f, padding=pad)
            d = (a.float() - b).abs()
            print(dt, fshape, pos, 'max err', float(d.max()), 'a[0,0]', a[0, 0].float().cpu().flatten()[:8].tolist(), 'b[0,0]', b[0, 0].cpu().flatten()[:8].tolist())
x = (torch.arange(2 * 16 * 5 * 6, dtype=torch.float32).reshape(2, 16, 5, 6) % 97).cuda()
xcl = x.contiguous(memory_format=torch.channels_last)
f = torch.ones([1, 1]).cuda()
a = upfirdn2d.upfirdn2d(xcl, f); b = upfirdn2d.upfirdn2d(x, f)
d = (a - b).abs()
print('per (n,c) max err:', d.amax(dim=(2, 3)).cpu().tolist())
print('a[0,5]', a[0, 5].cpu().tolist()); print('b[0,5]', b[0, 5].cpu().tolist())
