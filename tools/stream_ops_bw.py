"""Dev tool (GPU box): achieved HBM GB/s of the streaming ops at BASELINE config-2 sizes (algorithmic bytes / time)."""
import sys, os
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, 'pasta-gan-plusplus_amd'))
import torch
from torch_utils import custom_ops
custom_ops.verbosity = 'none'
from torch_utils.ops import upfirdn2d, bias_act, conv2d_mfma

def timeit(fn, n=20):
    for _ in range(3): fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n * 1e-3

dev = 'cuda'
f = upfirdn2d.setup_filter([1, 3, 3, 1]).to(dev)
def report(name, nbytes, t):
    print(f'{name:58s} {t*1e6:8.1f} us  {nbytes/t/1e9:7.0f} GB/s  ({nbytes/t/8e12*100:4.1f} % of 8 TB/s, {nbytes/t/6.3e12*100:4.1f} % of the 6.3 TB/s copy rate)', flush=True)

x = torch.randn(8, 64, 513, 513, device=dev)
y = upfirdn2d.upfirdn2d(x, f, padding=[1, 1, 1, 1], gain=4)
report('upfirdn2d blur [8,64,513,513]->[8,64,512,512]', 4 * (x.numel() + y.numel()), timeit(lambda: upfirdn2d.upfirdn2d(x, f, padding=[1, 1, 1, 1], gain=4)))
b = torch.randn(64, device=dev); nz = torch.randn(512, 512, device=dev)
report('  + fused noise/bias/lrelu/clamp tail (same bytes)', 4 * (x.numel() + y.numel()), timeit(lambda: upfirdn2d.upfirdn2d_bias_act(x, f, padding=[1, 1, 1, 1], gain=4, noise=nz, b=b, act='lrelu', act_gain=1.4, clamp=256)))
x2 = torch.randn(8, 64, 512, 512, device=dev)
y2 = upfirdn2d.upfirdn2d(x2, f, down=2, padding=[1, 1, 1, 1])
report('upfirdn2d down2 [8,64,512,512]->[8,64,256,256]', 4 * (x2.numel() + y2.numel()), timeit(lambda: upfirdn2d.upfirdn2d(x2, f, down=2, padding=[1, 1, 1, 1])))
x3 = torch.randn(8, 3, 256, 256, device=dev)
y3 = upfirdn2d.upsample2d(x3, f)
report('upfirdn2d skip-image up2 [8,3,256,256]->[8,3,512,512]', 4 * (x3.numel() + y3.numel()), timeit(lambda: upfirdn2d.upsample2d(x3, f)))
report('bias_act lrelu+clamp fp32 [8,64,512,512]', 8 * x2.numel(), timeit(lambda: bias_act.bias_act(x2, b, act='lrelu', clamp=256)))
xh = x2.half(); bh = b.half()
report('bias_act lrelu+clamp fp16 [8,64,512,512]', 4 * xh.numel(), timeit(lambda: bias_act.bias_act(xh, bh, act='lrelu', clamp=256)))
x4 = torch.randn(8, 128, 256, 256, device=dev)
report('instance_norm_stats [8,128,256,256] (read once)', 4 * x4.numel(), timeit(lambda: conv2d_mfma.instance_norm_stats(x4)))
report('torch copy_ (same size, reference point)', 8 * x2.numel(), timeit(lambda: torch.empty_like(x2).copy_(x2)))
