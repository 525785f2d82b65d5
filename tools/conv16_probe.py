"""Dev tool (GPU box): launches of one 16-bit conv shape, timed with events; also the target of rocprofv3 --pmc passes.
usage: conv16_probe.py N CIN COUT H [K] [reps]"""
import sys, os
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, 'pasta-gan-plusplus_amd'))
import torch
from torch_utils import custom_ops
custom_ops.verbosity = 'none'
from torch_utils.ops import conv2d_mfma16 as M
a = [int(v) for v in sys.argv[1:]]
N, cin, cout, H = a[:4]
K = a[4] if len(a) > 4 else 3
reps = a[5] if len(a) > 5 else 10
dt = torch.bfloat16
x = torch.randn(N, cin, H, H, device='cuda').to(dt).contiguous(memory_format=torch.channels_last)
w = torch.randn(cout, cin, K, K, device='cuda') / (K * cin ** 0.5)
pk, _, _ = M.pack_weight(w, dt)
bias = torch.randn(cout, device='cuda')
y = M.conv2d_forward(x, pk, cout, K, K, pad=(K // 2, K // 2), bias=bias, act='lrelu', alpha=0.2, gain=1.4, clamp=256)
torch.cuda.synchronize()
e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
torch.cuda._sleep(2000000)
e0.record()
for _ in range(reps):
    y = M.conv2d_forward(x, pk, cout, K, K, pad=(K // 2, K // 2), bias=bias, act='lrelu', alpha=0.2, gain=1.4, clamp=256)
e1.record()
torch.cuda.synchronize()
ms = e0.elapsed_time(e1) / reps
fl = 2.0 * N * cout * H * H * cin * K * K
by = 2.0 * N * H * H * (cin + cout)
print(f'conv16 N{N} {cin}->{cout} {H}x{H} k{K}: {ms * 1e3:.1f} us  {fl / ms / 1e9:.0f} TFLOP/s  {by / ms / 1e6:.0f} GB/s')
