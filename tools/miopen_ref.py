"""Dev tool (GPU box): same-hardware reference point -- PyTorch-ROCm (MIOpen) fp32 conv2d on the hot shapes vs this package's kernel."""
import sys, os
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, 'pasta-gan-plusplus_amd'))
import torch
import torch.nn.functional as F
from torch_utils import custom_ops
custom_ops.verbosity = 'none'
from torch_utils.ops import conv2d_mfma

def timeit(fn, n=10):
    for _ in range(3): fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n * 1e3

dev = 'cuda'
for bench_mode in (False, True):
    torch.backends.cudnn.benchmark = bench_mode
    for (N, H, cin, cout, k) in [(8, 256, 128, 128, 3), (8, 512, 64, 64, 3), (8, 64, 512, 512, 3), (8, 512, 128, 64, 1)]:
        x = torch.randn(N, cin, H, H, device=dev)
        w = torch.randn(cout, cin, k, k, device=dev) / (k * cin ** 0.5)
        fl = 2.0 * N * cout * H * H * cin * k * k
        t_ref = timeit(lambda: F.conv2d(x, w, padding=k // 2))
        pk = conv2d_mfma.pack_weight(w)
        t_own = timeit(lambda: conv2d_mfma.conv2d_forward(x, pk, cout, k, k, pad=(k // 2, k // 2)))
        err = float((F.conv2d(x, w, padding=k // 2) - conv2d_mfma.conv2d_forward(x, pk, cout, k, k, pad=(k // 2, k // 2))).abs().max())
        print(f'benchmark={bench_mode} N{N} H{H} cin{cin} cout{cout} k{k}: MIOpen {t_ref:8.1f} us ({fl/t_ref/1e6:6.1f} TF)   own {t_own:8.1f} us ({fl/t_own/1e6:6.1f} TF)   max|diff| {err:.2e}', flush=True)
