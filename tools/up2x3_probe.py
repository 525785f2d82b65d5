"""Dev tool (GPU box): the fp32 `up = 2` layer with its multiplies on the bf16 pipe (csrc/conv2d_up2x3.h) against the fp32-MFMA kernel (csrc/conv2d_up2.h):
error of both against a float64 transposed convolution, bit-identical repeats, time per launch.
    python tools/up2x3_probe.py [check|time|all]"""
import os
import sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, 'pasta-gan-plusplus_amd'))
import torch
from torch_utils import custom_ops
custom_ops.verbosity = 'none'
from torch_utils.ops import conv2d_mfma

what = sys.argv[1] if len(sys.argv) > 1 else 'all'
torch.manual_seed(0)


def timeit(fn, n=10):
    fn(); fn(); torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n):
        fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n


def run(x, w, packs, x3, **kw):
    conv2d_mfma.UP2_X3 = x3
    try:
        return conv2d_mfma.conv_up2_forward(x, packs, int(w.shape[0]), **kw)
    finally:
        conv2d_mfma.UP2_X3 = True


if what in ('check', 'all'):
    for (N, cin, cout, H, W) in [(1, 32, 32, 8, 32), (2, 32, 32, 20, 36), (1, 64, 40, 33, 64), (2, 128, 64, 64, 64), (1, 512, 256, 64, 64), (3, 48, 96, 17, 100)]:
        x = torch.randn(N, cin, H, W, device='cuda')
        w = torch.randn(cout, cin, 3, 3, device='cuda') / (3 * cin ** 0.5)
        ins, outs = torch.rand(N, cin, device='cuda') + 0.5, torch.rand(N, cout, device='cuda') + 0.5
        packs = conv2d_mfma.pack_up2(w)
        assert 'x3' in packs
        for kw, tag in ((dict(), 'plain'), (dict(in_scale=ins, out_scale=outs), 'modulated')):
            xs = x.double() * (kw['in_scale'].double()[:, :, None, None] if kw else 1.0)
            ref = torch.nn.functional.conv_transpose2d(xs, w.double().transpose(0, 1), stride=2)
            if kw:
                ref = ref * kw['out_scale'].double()[:, :, None, None]
            sc = ref.abs().max().item()
            a, b = run(x, w, packs, False, **kw), run(x, w, packs, True, **kw)
            same = all(torch.equal(b, run(x, w, packs, True, **kw)) for _ in range(3))
            print(f'{tag} N{N} {cin}->{cout} {H}x{W}: fp32 kernel err {(a.double() - ref).abs().max().item() / sc:.2e}  bf16x3 err {(b.double() - ref).abs().max().item() / sc:.2e}  '
                  f'|x3 - fp32| {(a - b).abs().max().item() / sc:.2e}  repeats identical: {same}', flush=True)

if what in ('time', 'all'):
    for (N, cin, cout, H) in [(8, 128, 64, 256), (8, 256, 128, 128), (8, 512, 256, 64), (8, 512, 512, 32), (4, 128, 64, 256), (4, 512, 256, 64)]:
        x = torch.randn(N, cin, H, H, device='cuda')
        w = torch.randn(cout, cin, 3, 3, device='cuda') / (3 * cin ** 0.5)
        ins, outs = torch.rand(N, cin, device='cuda') + 0.5, torch.rand(N, cout, device='cuda') + 0.5
        packs = conv2d_mfma.pack_up2(w)
        t0 = timeit(lambda: run(x, w, packs, False, in_scale=ins, out_scale=outs))
        t1 = timeit(lambda: run(x, w, packs, True, in_scale=ins, out_scale=outs))
        fl = 2.0 * N * cout * cin * 9 * H * H
        print(f'N{N} {cin}->{cout} {H}^2: fp32 kernel {t0 * 1e3:7.1f} us = {fl / t0 / 1e9:6.1f} TF ({fl / t0 / 1e9 / 157.3:.3f}) | bf16x3 {t1 * 1e3:7.1f} us = {fl / t1 / 1e9:6.1f} TF fp32-equivalent '
              f'({fl / t1 / 1e9 / 157.3:.3f} of the fp32 peak; executed 6x: {6 * fl / t1 / 1e9 / 2500:.3f} of the bf16 peak)  x{t0 / t1:.2f}', flush=True)
