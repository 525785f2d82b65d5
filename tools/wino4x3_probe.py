"""Dev tool (GPU box): the X3 form of the F(4x4,3x3) kernel (transform-domain GEMM as six bf16 products of exact three-term splits, csrc/conv2d_wino4.h) against the
fp32 form: error of both against float64 (max |y - ref| / max |ref|), bit-identical repeats, every tail, and time per launch.
    python tools/wino4x3_probe.py [check|time|all]"""
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
    fn(); fn(); torch.cuda.synchronize()
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
                                      (1, 128, 128, 40, 192, 1), (3, 64, 64, 40, 64, 3), (2, 128, 128, 128, 128, 1), (1, 512, 512, 64, 64, 1)]:
        x = torch.randn(N, cin, H, W, device=dev)
        w = torch.randn(cout, cin, 3, 3, device=dev) / (3 * cin ** 0.5)
        ref = torch.nn.functional.conv2d(x.double(), w.double(), padding=pad)
        sc = ref.abs().max().item()
        y2, y4 = run(x, w, cout, pad, 2), run(x, w, cout, pad, 4)
        same = all(torch.equal(y4, run(x, w, cout, pad, 4)) for _ in range(3))
        print(f'plain N{N} cin{cin} cout{cout} {H}x{W} pad{pad}: fp32 form err {(y2.double() - ref).abs().max().item() / sc:.2e}  X3 err {(y4.double() - ref).abs().max().item() / sc:.2e}'
              f'  |X3 - fp32 form| {(y4 - y2).abs().max().item() / sc:.2e}  repeats identical: {same}', flush=True)
        OH, OW = ref.shape[2:]
        ins = torch.rand(N, cin, device=dev) + 0.5; outs = torch.rand(N, cout, device=dev) + 0.5
        nz = torch.randn(OH, OW, device=dev); b = torch.randn(cout, device=dev); res = torch.randn(N, cout, OH, OW, device=dev)
        kw = dict(in_scale=ins, out_scale=outs, noise=nz, noise_gain=0.3, bias=b, act='lrelu', alpha=0.2, gain=1.4, clamp=2.0, residual=res)
        print(f'   fused: |X3 - fp32 form| {(run(x, w, cout, pad, 2, **kw) - run(x, w, cout, pad, 4, **kw)).abs().max().item():.2e}', flush=True)
        if cout % 64 == 0:
            c = cout // 2
            wg_, wb_ = w[:c].contiguous(), w[c:].contiguous()
            sx = torch.randn(N, c, OH, OW, device=dev); mean = torch.randn(N, c, device=dev); rstd = torch.rand(N, c, device=dev) + 0.5
            outs_ = []
            for algo in (2, 4):
                pk = conv2d_mfma.pack_spade_gamma_beta(wg_, wb_, winograd=algo)
                outs_.append(conv2d_mfma.conv2d_forward(x, pk, cout, 3, 3, pad=(pad, pad), spade=(sx, mean, rstd), winograd=algo, act='lrelu', alpha=0.2, gain=1.4, clamp=3.0))
            print(f'   spade: |X3 - fp32 form| {(outs_[0] - outs_[1]).abs().max().item():.2e}  scale {outs_[0].abs().max().item():.2f}', flush=True)
    # integers: both forms exact
    x = torch.randint(-3, 4, (2, 64, 32, 64), device=dev).float(); w = torch.randint(-2, 3, (64, 64, 3, 3), device=dev).float()
    ref = torch.nn.functional.conv2d(x.double(), w.double(), padding=1)
    print('integer data: fp32 form max |err|', (run(x, w, 64, 1, 2).double() - ref).abs().max().item(), ' X3', (run(x, w, 64, 1, 4).double() - ref).abs().max().item(), flush=True)

if what in ('time', 'all'):
    for (N, H, cin, cout) in [(8, 256, 128, 128), (8, 256, 128, 256), (8, 512, 64, 64), (8, 512, 64, 128), (8, 128, 256, 256), (8, 64, 512, 512), (8, 32, 512, 512), (4, 256, 128, 128), (4, 512, 64, 64)]:
        x = torch.randn(N, cin, H, H, device=dev)
        w = torch.randn(cout, cin, 3, 3, device=dev) / (3 * cin ** 0.5)
        pk = {a: conv2d_mfma.pack_weight(w, winograd=a) for a in (2, 4)}
        ms = {a: timeit(lambda a=a: conv2d_mfma.conv2d_forward(x, pk[a], cout, 3, 3, pad=(1, 1), winograd=a)) for a in (2, 4)}
        fl = 2.0 * N * cout * H * H * cin * 9 / 4
        print(f'N{N} H{H} cin{cin:4d} cout{cout:4d}: fp32 form {ms[2]*1e3:8.1f} us = {fl/ms[2]/1e9:6.1f} TF ({fl/ms[2]/1e9/157.3:.3f}) | X3 {ms[4]*1e3:8.1f} us = {fl/ms[4]/1e9:6.1f} TF fp32-equivalent '
              f'({fl/ms[4]/1e9/157.3:.3f} of the fp32 peak; executed 6x: {6*fl/ms[4]/1e9/2500:.3f} of the bf16 peak)  x{ms[2]/ms[4]:.2f}', flush=True)
