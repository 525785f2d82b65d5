"""Dev tool (GPU box): Winograd F(2x2,3x3) kernel vs the direct MFMA kernel -- max error against an fp64 reference and time."""
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
    return e0.elapsed_time(e1) / n


dev = 'cuda'
torch.manual_seed(0)
# ---- correctness on small / ragged shapes, with fused stages
for (N, cin, cout, H, W, pad) in [(1, 16, 64, 8, 64, 1), (2, 20, 70, 9, 71, 1), (1, 3, 3, 5, 7, 1), (2, 48, 128, 33, 130, 0), (1, 128, 64, 16, 16, 2)]:
    x = torch.randn(N, cin, H, W, device=dev)
    w = torch.randn(cout, cin, 3, 3, device=dev) / (3 * cin ** 0.5)
    ref = torch.nn.functional.conv2d(x.double().cpu(), w.double().cpu(), padding=pad)
    yd = conv2d_mfma.conv2d_forward(x, conv2d_mfma.pack_weight(w), cout, 3, 3, pad=(pad, pad))
    yw = conv2d_mfma.conv2d_forward(x, conv2d_mfma.pack_weight(w, winograd=True), cout, 3, 3, pad=(pad, pad), winograd=True)
    ed = (yd.double().cpu() - ref).abs().max().item(); ew = (yw.double().cpu() - ref).abs().max().item()
    print(f'plain N{N} cin{cin} cout{cout} {H}x{W} pad{pad}: direct err {ed:.2e}  winograd err {ew:.2e}  scale {ref.abs().max():.2f}', flush=True)
    # fused: in_scale, out_scale, noise, bias, lrelu, gain, clamp, residual
    OH, OW = ref.shape[2:]
    ins = torch.rand(N, cin, device=dev) + 0.5; outs = torch.rand(N, cout, device=dev) + 0.5
    nz = torch.randn(OH, OW, device=dev); b = torch.randn(cout, device=dev); res = torch.randn(N, cout, OH, OW, device=dev)
    kw = dict(in_scale=ins, out_scale=outs, noise=nz, noise_gain=0.3, bias=b, act='lrelu', alpha=0.2, gain=1.4, clamp=2.0, residual=res)
    yd = conv2d_mfma.conv2d_forward(x, conv2d_mfma.pack_weight(w), cout, 3, 3, pad=(pad, pad), **kw)
    yw = conv2d_mfma.conv2d_forward(x, conv2d_mfma.pack_weight(w, winograd=True), cout, 3, 3, pad=(pad, pad), winograd=True, **kw)
    print(f'   fused: |winograd - direct| {(yd - yw).abs().max().item():.2e}', flush=True)
    kw = dict(in_act='lrelu', in_alpha=0.2, in_gain=1.3, in_clamp=1.5, bias=b)
    yd = conv2d_mfma.conv2d_forward(x, conv2d_mfma.pack_weight(w), cout, 3, 3, pad=(pad, pad), **kw)
    yw = conv2d_mfma.conv2d_forward(x, conv2d_mfma.pack_weight(w, winograd=True), cout, 3, 3, pad=(pad, pad), winograd=True, **kw)
    print(f'   prologue act: |winograd - direct| {(yd - yw).abs().max().item():.2e}', flush=True)

# ---- time
for (N, H, cin, cout) in [(8, 256, 128, 128), (8, 256, 64, 128), (8, 512, 64, 64), (8, 128, 256, 256), (8, 64, 512, 512), (8, 32, 512, 512), (8, 16, 512, 512), (8, 256, 128, 256)]:
    x = torch.randn(N, cin, H, H, device=dev)
    w = torch.randn(cout, cin, 3, 3, device=dev) / (3 * cin ** 0.5)
    pk, pw = conv2d_mfma.pack_weight(w), conv2d_mfma.pack_weight(w, winograd=True)
    md = timeit(lambda: conv2d_mfma.conv2d_forward(x, pk, cout, 3, 3, pad=(1, 1)))
    mw = timeit(lambda: conv2d_mfma.conv2d_forward(x, pw, cout, 3, 3, pad=(1, 1), winograd=True))
    fl = 2.0 * N * cout * H * H * cin * 9
    print(f'N{N} H{H} cin{cin:4d} cout{cout:4d}: direct {md*1e3:9.1f} us {fl/md/1e9:7.1f} TF | winograd {mw*1e3:9.1f} us {fl/mw/1e9:7.1f} TF(alg)  x{md/mw:.2f}', flush=True)
