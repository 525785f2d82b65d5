"""conv2d16_wgrad vs aten::convolution_backward (MIOpen) on the discriminator's half-precision shapes of config 4 (batch 4).  GPU box:
    python tools/wgrad16_probe.py > gpurun_out/wgrad16_probe.txt"""
import os
import sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, 'pasta-gan-plusplus_amd'))
import torch
from torch_utils.ops import conv2d_mfma16

DEV = torch.device('cuda', 0)
CASES = [  # cin, cout, k, stride, pad, H
    (64, 64, 3, 1, 1, 512), (64, 128, 3, 2, 0, 513), (128, 128, 3, 1, 1, 256), (128, 256, 3, 2, 0, 257), (256, 256, 3, 1, 1, 128),
    (256, 512, 3, 2, 0, 129), (512, 512, 3, 1, 1, 64), (512, 512, 3, 2, 0, 65), (16, 64, 1, 1, 0, 512), (64, 128, 1, 1, 0, 256), (128, 256, 1, 1, 0, 128),
]


def timed(fn, reps=10):
    for _ in range(2):
        fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps):
        fn()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / reps * 1e3


def main():
    n = 4
    print(f'{"shape":38s} {"native us":>10s} {"TF":>7s} {"aten us":>10s} {"TF":>7s}  max|d|/scale')
    for cin, cout, k, s, pad, h in CASES:
        x = torch.randn([n, cin, h, h], device=DEV).half().contiguous(memory_format=torch.channels_last)
        oh = (h + 2 * pad - k) // s + 1
        dy = torch.randn([n, cout, oh, oh], device=DEV).half().contiguous(memory_format=torch.channels_last)
        w = torch.zeros([cout, cin, k, k], device=DEV).half().contiguous(memory_format=torch.channels_last)
        nat_fn = lambda: conv2d_mfma16.weight_gradient(x, dy, w.shape, (pad, pad), stride=s)
        aten_fn = lambda: torch.ops.aten.convolution_backward(dy, x, w, None, [s, s], [pad, pad], [1, 1], False, [0, 0], 1, [False, True, False])[1]
        a, b = nat_fn(), aten_fn().float()
        err = float((a - b).abs().max() / b.abs().max())
        fl = 2.0 * n * cout * cin * k * k * oh * oh
        tn, ta = timed(nat_fn), timed(aten_fn)
        print(f'N{n} {cin:3d}->{cout:3d} k{k} s{s} {h:3d}^2'.ljust(38) + f' {tn:10.1f} {fl / tn / 1e6:7.1f} {ta:10.1f} {fl / ta / 1e6:7.1f}  {err:.1e}')


if __name__ == '__main__':
    main()
