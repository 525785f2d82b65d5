"""Dev tool (GPU box): time pg_conv2d_forward_splitk at explicit ksplit values on the low-resolution phase-conv shapes."""
import sys, os, ctypes
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, 'pasta-gan-plusplus_amd'))
import torch
from torch_utils import custom_ops
custom_ops.verbosity = 'none'
from torch_utils.ops import conv2d_mfma
from torch_utils.ops import _native as nat
lib = conv2d_mfma._init().lib

def timeit(fn, n=20):
    fn(); torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n * 1e3

for (N, H, cin, cout, kh, kw) in [(8, 32, 512, 512, 2, 2), (8, 32, 512, 512, 1, 1), (8, 64, 512, 256, 2, 2), (8, 16, 512, 512, 2, 2)]:
    x = torch.randn(N, cin, H, H, device='cuda')
    w = torch.randn(cout, cin, kh, kw, device='cuda') / (cin * kh * kw) ** 0.5
    pk = conv2d_mfma.pack_weight(w)
    oh, ow = H - kh + 1 + 2 * (kh - 1), H - kw + 1 + 2 * (kw - 1)
    y = torch.empty(N, cout, oh, ow, device='cuda')
    fz = conv2d_mfma.Fusion(); fz.in_clamp = -1.0; fz.clamp = -1.0
    out = []
    for ks in (1, 2, 4, 8, 16):
        ws = torch.empty(ks * y.numel(), device='cuda')
        def run():
            st = lib.pg_conv2d_forward_splitk(nat.ptr(x), nat.ptr(pk), nat.ptr(y), N, cin, H, H, cout, kh, kw, 1, kh - 1, kw - 1, oh, ow, nat.i64arr(y.stride()),
                                              1, 1, 0, 0, ctypes.byref(fz), nat.ptr(ws), ks, nat.stream_of(x))
            assert st == 0, st
        out.append(f'ks{ks}: {timeit(run):7.1f}us')
    fl = 2.0 * N * cout * oh * ow * cin * kh * kw
    print(f'N{N} {cin}->{cout} {H}x{H} k{kh}x{kw}  ' + '  '.join(out) + f'   ({fl/1e9:.1f} GF)', flush=True)
