"""Dev tool (GPU box): native weight-gradient kernel vs aten (MIOpen) on the training step's hot shapes."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, 'pasta-gan-plusplus_amd'))
import torch
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
    return e0.elapsed_time(e1) / n * 1e-3

for (n, cin, cout, hw, k) in [(4, 128, 128, 256, 3), (4, 64, 64, 512, 3), (4, 128, 256, 256, 3), (4, 64, 128, 512, 3), (4, 512, 512, 32, 3), (4, 256, 256, 64, 3),
                              (4, 64, 64, 512, 1), (4, 128, 64, 512, 1), (4, 192, 128, 256, 1)]:
    x = torch.randn(n, cin, hw, hw, device='cuda'); dy = torch.randn(n, cout, hw, hw, device='cuda'); w = torch.randn(cout, cin, k, k, device='cuda')
    fl = 2.0 * n * hw * hw * cin * cout * k * k
    t_nat = timeit(lambda: conv2d_mfma.weight_gradient(x, dy, w.shape, (k // 2, k // 2)))
    t_at = timeit(lambda: torch.ops.aten.convolution_backward(dy, x, w, None, [1, 1], [k // 2, k // 2], [1, 1], False, [0, 0], 1, [False, True, False]))
    print(f'N{n} {cin}->{cout} {hw}^2 k{k}: native {t_nat*1e6:8.1f} us {fl/t_nat/1e12:6.1f} TF | aten {t_at*1e6:8.1f} us {fl/t_at/1e12:6.1f} TF', flush=True)
