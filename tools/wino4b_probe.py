"""Dev tool (GPU box): the two F(4x4,3x3) kernels side by side -- csrc/conv2d_wino4.h (form 2: one 12-wave workgroup per CU, 32x32x2 MFMA) and
csrc/conv2d_wino4b.h (form 3: two 8-wave workgroups per CU, 16x16x4 MFMA) -- per launch, round-robin, median of several rounds, on the config-2 shapes
and tail kinds; plus |form 3 - form 2| on each.   python tools/wino4b_probe.py [rounds]"""
import os
import statistics
import sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, 'pasta-gan-plusplus_amd'))
import torch
from torch_utils import custom_ops
custom_ops.verbosity = 'none'
from torch_utils.ops import conv2d_mfma

rounds = int(sys.argv[1]) if len(sys.argv) > 1 else 7
dev = 'cuda'
torch.manual_seed(0)
SHAPES = [(8, 256, 128, 128, 'plain'), (8, 256, 128, 128, 'res'), (8, 256, 128, 256, 'spade'), (8, 512, 64, 64, 'plain'), (8, 512, 64, 128, 'spade'), (8, 512, 64, 64, 'mod'),
          (8, 256, 64, 128, 'plain'), (8, 128, 256, 256, 'mod'), (8, 64, 512, 512, 'mod'), (8, 32, 512, 512, 'mod')]
for (N, H, cin, cout, kind) in SHAPES:
    x = torch.randn(N, cin, H, H, device=dev)
    w = torch.randn(cout, cin, 3, 3, device=dev) / (3 * cin ** 0.5)
    kw = dict(bias=torch.randn(cout, device=dev), act='lrelu', alpha=0.2, gain=1.4, clamp=256.0)
    packs = {}
    if kind == 'spade':
        c = cout // 2
        sx = torch.randn(N, c, H, H, device=dev)
        kw = dict(spade=(sx, torch.randn(N, c, device=dev), torch.rand(N, c, device=dev) + 0.5), act='relu', gain=1.4)
        for f in (2, 3):
            packs[f] = conv2d_mfma.pack_spade_gamma_beta(w[:c].contiguous(), w[c:].contiguous(), winograd=f)
    else:
        for f in (2, 3):
            packs[f] = conv2d_mfma.pack_weight(w, winograd=f)
        if kind == 'res':
            kw['residual'] = torch.randn(N, cout, H, H, device=dev)
        if kind == 'mod':
            kw.update(in_scale=torch.rand(N, cin, device=dev) + 0.5, out_scale=torch.rand(N, cout, device=dev) + 0.5, noise=torch.randn(H, H, device=dev), noise_gain=0.1)
    run = lambda f: conv2d_mfma.conv2d_forward(x, packs[f], cout, 3, 3, pad=(1, 1), winograd=f, **kw)
    y2, y3 = run(2), run(3)
    err = float((y2 - y3).abs().max())
    times = {2: [], 3: []}
    for r in range(rounds + 1):
        for f in (2, 3):
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            for _ in range(4):
                run(f)
            e1.record(); torch.cuda.synchronize()
            if r > 0:
                times[f].append(e0.elapsed_time(e1) / 4 * 1e3)
    fl = 2.0 * N * cout * H * H * cin * 9 / 4
    t2, t3 = statistics.median(times[2]), statistics.median(times[3])
    print(f'N{N} H{H} {cin:3d}->{cout:3d} {kind:5s}: form 2 {t2:7.1f} us ({fl / t2 / 1e6 / 157.3:.3f} of peak)  form 3 {t3:7.1f} us ({fl / t3 / 1e6 / 157.3:.3f})  x{t2 / t3:.3f}   |3 - 2| {err:.2e} (scale {float(y2.abs().max()):.1f})', flush=True)
