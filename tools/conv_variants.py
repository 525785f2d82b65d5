"""Dev tool (GPU box): time the epilogue/prologue variants of the 3x3 conv relative to the plain launch of the same shape."""
import sys, os
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
    return e0.elapsed_time(e1) / n * 1e3

dev = 'cuda'
for (N, H, cin, cout) in [(8, 256, 128, 128), (8, 512, 64, 64)]:
    x = torch.randn(N, cin, H, H, device=dev)
    w = torch.randn(cout, cin, 3, 3, device=dev) / (3 * cin ** 0.5)
    pk = conv2d_mfma.pack_weight(w)
    res = torch.randn(N, cout, H, H, device=dev)
    b = torch.randn(cout, device=dev)
    s = torch.rand(N, cin, device=dev) + 0.5
    d = torch.rand(N, cout, device=dev) + 0.5
    noise = torch.randn(H, H, device=dev)
    pk2 = conv2d_mfma.pack_spade_gamma_beta(w, w)
    mean, rstd = torch.randn(N * cout, device=dev), torch.rand(N * cout, device=dev) + 0.5
    base = timeit(lambda: conv2d_mfma.conv2d_forward(x, pk, cout, 3, 3, pad=(1, 1)))
    rows = [('plain', base)]
    rows.append(('bias+lrelu+clamp', timeit(lambda: conv2d_mfma.conv2d_forward(x, pk, cout, 3, 3, pad=(1, 1), bias=b, act='lrelu', alpha=0.2, gain=1.4, clamp=256))))
    rows.append(('residual', timeit(lambda: conv2d_mfma.conv2d_forward(x, pk, cout, 3, 3, pad=(1, 1), residual=res))))
    rows.append(('modulated+noise (SynthesisLayer)', timeit(lambda: conv2d_mfma.conv2d_forward(x, pk, cout, 3, 3, pad=(1, 1), in_scale=s, out_scale=d, noise=noise, bias=b, act='lrelu', alpha=0.2, gain=1.4, clamp=256))))
    rows.append(('XF relu prologue', timeit(lambda: conv2d_mfma.conv2d_forward(x, pk, cout, 3, 3, pad=(1, 1), in_act='relu', in_gain=1.4))))
    rows.append(('XF + residual', timeit(lambda: conv2d_mfma.conv2d_forward(x, pk, cout, 3, 3, pad=(1, 1), in_act='relu', in_gain=1.4, residual=res))))
    rows.append(('spade gamma/beta (2x FLOPs)', timeit(lambda: conv2d_mfma.conv2d_forward(x, pk2, 2 * cout, 3, 3, pad=(1, 1), spade=(res, mean, rstd))) / 2))
    for name, t in rows:
        print(f'N{N} H{H} cin{cin} cout{cout} {name:36s} {t:9.1f} us  x{t / base:5.3f}', flush=True)
