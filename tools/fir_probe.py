"""Dev tool (GPU box): the blur after an up-sampling convolution, [8,64,513,513] (rows pitched to 516) -> [8,64,512,512], with and
without the fused tail; PG_FIR_BLUR4=0 selects the generic tiled kernel."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, 'pasta-gan-plusplus_amd'))
import torch
from torch_utils import custom_ops
custom_ops.verbosity = 'none'
from torch_utils.ops import upfirdn2d

def timeit(fn, n=20):
    for _ in range(3): fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n * 1e-3

f = upfirdn2d.setup_filter([1, 3, 3, 1]).cuda()
buf = torch.randn(8, 64, 513, 516, device='cuda')
x = buf[:, :, :, :513]
nbytes = 4 * (8 * 64 * 513 * 513 + 8 * 64 * 512 * 512)
b = torch.randn(64, device='cuda'); nz = torch.randn(512, 512, device='cuda')
t = timeit(lambda: upfirdn2d.upfirdn2d(x, f, padding=[1, 1, 1, 1], gain=4))
print(f'blur pitched input: {t*1e6:.1f} us  {nbytes/t/1e9:.0f} GB/s')
t = timeit(lambda: upfirdn2d.upfirdn2d_bias_act(x, f, padding=[1, 1, 1, 1], gain=4, noise=nz, b=b, act='lrelu', act_gain=1.4, clamp=256))
print(f'blur + tail pitched input: {t*1e6:.1f} us  {nbytes/t/1e9:.0f} GB/s')
