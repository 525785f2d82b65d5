"""Dev tool (GPU box): host-side cost per launch of the Python wrappers (tiny tensors: the GPU is never the bottleneck)."""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, 'pasta-gan-plusplus_amd'))
import torch
from torch_utils import custom_ops
custom_ops.verbosity = 'none'
from torch_utils.ops import bias_act, upfirdn2d, conv2d_mfma, conv2d_mfma16, conv2d_gradfix

def per_call(fn, n=2000):
    for _ in range(50): fn()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(n): fn()
    dt = time.perf_counter() - t0
    torch.cuda.synchronize()
    return dt / n * 1e6

x = torch.randn(1, 16, 8, 8, device='cuda'); b = torch.randn(16, device='cuda'); f = upfirdn2d.setup_filter([1, 3, 3, 1]).cuda()
w = torch.randn(16, 16, 3, 3, device='cuda'); pk = conv2d_mfma.pack_weight(w)
xh = x.to(torch.bfloat16).contiguous(memory_format=torch.channels_last); pk16, _, _ = conv2d_mfma16.pack_weight(w, torch.bfloat16)
with torch.no_grad():
    print(f'torch add                    {per_call(lambda: x + x):6.1f} us/call')
    print(f'bias_act                     {per_call(lambda: bias_act.bias_act(x, b, act="lrelu")):6.1f}')
    print(f'upfirdn2d                    {per_call(lambda: upfirdn2d.upfirdn2d(x, f, padding=1)):6.1f}')
    print(f'conv2d_mfma.conv2d_forward   {per_call(lambda: conv2d_mfma.conv2d_forward(x, pk, 16, 3, 3, pad=(1, 1), bias=b, act="lrelu")):6.1f}')
    print(f'conv2d_mfma16.conv2d_forward {per_call(lambda: conv2d_mfma16.conv2d_forward(xh, pk16, 16, 3, 3, pad=(1, 1))):6.1f}')
    print(f'conv2d_gradfix.conv2d        {per_call(lambda: conv2d_gradfix.conv2d(x, w, padding=1)):6.1f}')
    print(f'F.conv2d (aten)              {per_call(lambda: torch.nn.functional.conv2d(x, w, padding=1)):6.1f}')
