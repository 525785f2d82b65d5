"""Dev tool (GPU box): the Winograd F(4x4,3x3) kernel (csrc/conv2d_wino4.h) against the direct MFMA kernel and an fp64 reference --
plain, every fused stage, SPADE mode, edge / ragged tiles -- and its time beside F(2x2,3x3) and the direct kernel.
    python tools/wino4_probe.py [check|time|all]"""
import os
import sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, 'pasta-gan-plusplus_amd'))
import torch
from torch_utils import custom_ops
custom_ops.verbosity = 'none'
from torch_utils.ops import conv2d_mfma

what = sys.argv[1] if len(sys.argv) > 1 else 'all'
dev = 'cuda'
torch.manual_seed(0)


def timeit(fn, n=10):
    fn(); torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n):
        fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n


def run(x, w, cout, pad, algo, **kw):
    return conv2d_mfma.conv2d_forward(x, conv2d_mfma.pack_weight(w, winograd=algo), cout, 3, 3, pad=(pad, pad), winograd=algo, **kw)


if what in ('check', 'all'):
    for (N, cin, cout, H, W, pad) in [(1, 32, 64, 8, 64, 1), (2, 64, 64, 16, 128, 1), (2, 32, 128, 24, 64, 1), (1, 48, 70, 9, 72, 1), (2, 20, 40, 13, 100, 1),
                                      (1, 128, 128, 40, 192, 1), (3, 64, 64, 64, 64, 2), (3, 64, 64, 40, 64, 3), (2, 40, 64, 7, 8, 1)]:
        if (W + 2 * pad - 2) % 4 != 0:
            print(f'N{N} cin{cin} cout{cout} {H}x{W} pad{pad}: output width {W + 2 * pad - 2} is no multiple of 4 -> the kernel declines (F(2x2) serves it)')
            continue
        x = torch.randn(N, cin, H, W, device=dev)
        w = torch.randn(cout, cin, 3, 3, device=dev) / (3 * cin ** 0.5)
        ref = torch.nn.functional.conv2d(x.double().cpu(), w.double().cpu(), padding=pad)
        yd, y4 = run(x, w, cout, pad, 0), run(x, w, cout, pad, 2)
        ed, e4 = (yd.double().cpu() - ref).abs().max().item(), (y4.double().cpu() - ref).abs().max().item()
        print(f'plain N{N} cin{cin} cout{cout} {H}x{W} pad{pad}: direct err {ed:.2e}  F(4x4) err {e4:.2e}  scale {ref.abs().max():.2f}', flush=True)
        OH, OW = ref.shape[2:]
        ins = torch.rand(N, cin, device=dev) + 0.5; outs = torch.rand(N, cout, device=dev) + 0.5
        nz = torch.randn(OH, OW, device=dev); b = torch.randn(cout, device=dev); res = torch.randn(N, cout, OH, OW, device=dev)
        kw = dict(in_scale=ins, out_scale=outs, noise=nz, noise_gain=0.3, bias=b, act='lrelu', alpha=0.2, gain=1.4, clamp=2.0, residual=res)
        print(f'   fused: |F(4x4) - direct| {(run(x, w, cout, pad, 0, **kw) - run(x, w, cout, pad, 2, **kw)).abs().max().item():.2e}', flush=True)
        nzb = torch.randn(N, OH, OW, device=dev)
        kw = dict(noise=nzb, bias=b, act='relu', gain=1.0)
        print(f'   per-sample noise + relu: |F(4x4) - direct| {(run(x, w, cout, pad, 0, **kw) - run(x, w, cout, pad, 2, **kw)).abs().max().item():.2e}', flush=True)
        if cout % 64 == 0:
            c = cout // 2
            wg_, wb_ = w[:c].contiguous(), w[c:].contiguous()
            sx = torch.randn(N, c, OH, OW, device=dev); mean = torch.randn(N, c, device=dev); rstd = torch.rand(N, c, device=dev) + 0.5
            outs_ = []
            for algo in (0, 2):
                pk = conv2d_mfma.pack_spade_gamma_beta(wg_, wb_, winograd=algo)
                outs_.append(conv2d_mfma.conv2d_forward(x, pk, cout, 3, 3, pad=(pad, pad), spade=(sx, mean, rstd), winograd=algo, act='lrelu', alpha=0.2, gain=1.4, clamp=3.0))
            print(f'   spade: |F(4x4) - direct| {(outs_[0] - outs_[1]).abs().max().item():.2e}  scale {outs_[0].abs().max().item():.2f}', flush=True)

if what in ('time', 'all'):
    for (N, H, cin, cout) in [(8, 256, 128, 128), (8, 256, 128, 256), (8, 512, 64, 64), (8, 512, 64, 128), (8, 128, 256, 256), (8, 64, 512, 512), (8, 32, 512, 512), (8, 16, 512, 512), (8, 8, 512, 512), (16, 256, 128, 128), (16, 512, 64, 64)]:
        x = torch.randn(N, cin, H, H, device=dev)
        w = torch.randn(cout, cin, 3, 3, device=dev) / (3 * cin ** 0.5)
        pk = [conv2d_mfma.pack_weight(w, winograd=a) for a in (0, 1, 2)]
        ms = [timeit(lambda a=a: conv2d_mfma.conv2d_forward(x, pk[a], cout, 3, 3, pad=(1, 1), winograd=a)) for a in (0, 1, 2)]
        fl = 2.0 * N * cout * H * H * cin * 9
        err = (conv2d_mfma.conv2d_forward(x, pk[2], cout, 3, 3, pad=(1, 1), winograd=2) - conv2d_mfma.conv2d_forward(x, pk[0], cout, 3, 3, pad=(1, 1))).abs().max().item()
        print(f'N{N} H{H} cin{cin:4d} cout{cout:4d}: direct {ms[0]*1e3:8.1f} us | F(2x2) {ms[1]*1e3:8.1f} us = {fl*4/9/ms[1]/1e9:6.1f} TF executed | '
              f'F(4x4) {ms[2]*1e3:8.1f} us = {fl/4/ms[2]/1e9:6.1f} TF executed ({fl/4/ms[2]/1e9/157.3:.3f} of peak)  x{ms[1]/ms[2]:.2f} vs F(2x2)  |F4-direct| {err:.1e}', flush=True)
