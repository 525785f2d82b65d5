"""GPU box: the 7x7 stem (N 3->64 at 512^2) -- conv2d_stem7x3 (bf16 pipe, three-term split) against conv2d_mfma<7,7,1> (fp32 MFMA): time and error vs float64 on a crop."""
import os, sys, statistics, math
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, 'pasta-gan-plusplus_amd'))
import torch
from torch_utils import custom_ops
custom_ops.verbosity = 'none'
from torch_utils.ops import conv2d_mfma
for n, h in ((8, 512), (16, 512), (4, 256)):
    x = torch.rand(n, 3, h, h, device='cuda') * 2 - 1
    w = torch.randn(64, 3, 7, 7, device='cuda')
    b = torch.randn(64, device='cuda')
    sc = 1 / math.sqrt(147)
    px, p32 = conv2d_mfma.pack_stem7(w, scale=sc), conv2d_mfma.pack_weight(w, scale=sc)
    fx = lambda: conv2d_mfma.conv_stem7_forward(x, px, 64, bias=b, act='relu', gain=math.sqrt(2))
    f32 = lambda: conv2d_mfma.conv2d_forward(x, p32, 64, 7, 7, pad=(3, 3), bias=b, act='relu', gain=math.sqrt(2))
    t = {}
    for name, f in (('x3', fx), ('fp32', f32)):
        ts = []
        for r in range(6):
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            for _ in range(4):
                y = f()
            e1.record(); torch.cuda.synchronize()
            if r:
                ts.append(e0.elapsed_time(e1) / 4 * 1e3)
        t[name] = statistics.median(ts)
    ref = (torch.nn.functional.conv2d(x[:1].double(), w.double() * sc, padding=3) + b.double()[None, :, None, None]).clamp(min=0) * math.sqrt(2)
    ex, e32 = float((fx()[:1].double() - ref).abs().max()), float((f32()[:1].double() - ref).abs().max())
    fl = 2.0 * n * 64 * h * h * 147
    print(f'N{n} 3->64 {h}^2: x3 {t["x3"]:7.1f} us ({fl / t["x3"] / 1e6:6.1f} TF)  fp32 {t["fp32"]:7.1f} us ({fl / t["fp32"] / 1e6:6.1f} TF)  err x3 {ex:.2e} fp32 {e32:.2e} (scale {float(ref.abs().max()):.2f})', flush=True)
