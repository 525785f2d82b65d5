"""Dev tool (GPU box): time the 3x3 MFMA conv at several Cin to split fixed per-block cost from per-chunk cost."""
import sys, os, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, 'pasta-gan-plusplus_amd'))
import torch
from torch_utils import custom_ops
custom_ops.verbosity = 'none'
from torch_utils.ops import conv2d_mfma

def timeit(fn, n=10):
    fn(); torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n

dev = 'cuda'
for (N, H, cout) in [(8, 256, 128), (8, 512, 64), (8, 64, 512)]:
    for cin in (8, 64, 128, 256, 512):
        if H == 512 and cin > 128: continue
        x = torch.randn(N, cin, H, H, device=dev)
        w = torch.randn(cout, cin, 3, 3, device=dev) / (3 * cin ** 0.5)
        pk = conv2d_mfma.pack_weight(w)
        ms = timeit(lambda: conv2d_mfma.conv2d_forward(x, pk, cout, 3, 3, pad=(1, 1)))
        fl = 2.0 * N * cout * H * H * cin * 9
        print(f'N{N} H{H} cin{cin:4d} cout{cout:4d}: {ms*1e3:9.1f} us  {fl/ms/1e9:7.1f} TF', flush=True)
