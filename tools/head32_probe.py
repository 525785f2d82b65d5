"""Dev tool (GPU box): the fp32 streaming 1x1 head (ToRGB of the config-2 network, pg_conv1x1_small) alone on the config-2 shapes.
    python tools/head32_probe.py [N,cin,H ...]"""
import sys, os
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, 'pasta-gan-plusplus_amd'))
import torch
from torch_utils import custom_ops
custom_ops.verbosity = 'none'
from torch_utils.ops import conv2d_mfma as M

shapes = [(8, 512, 4), (8, 512, 8), (8, 512, 16), (8, 512, 32), (8, 512, 64), (8, 256, 128), (8, 128, 256), (8, 64, 512)]
if len(sys.argv) > 1:
    shapes = [tuple(int(v) for v in a.split(',')) for a in sys.argv[1:]]


def timed(run, reps=20):
    run(); torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    torch.cuda._sleep(1000000)
    e0.record()
    for _ in range(reps):
        run()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) * 1e3 / reps


for (N, cin, H) in shapes:
    for cout in (3, 7):
        x = torch.randn(N, cin, H, H, device='cuda')
        w = torch.randn(cout, cin, 1, 1, device='cuda') / cin ** 0.5
        styles = torch.rand(N, cin, device='cuda') + 0.5
        bias = torch.randn(cout, device='cuda')
        skip = torch.randn(N, cout, H, H, device='cuda')
        by = N * H * H * (4.0 * cin + 4 * cout)
        t0 = timed(lambda: M.conv1x1_small(x, w, styles=styles, bias=bias, clamp=256))
        t1 = timed(lambda: M.conv1x1_small(x, w, styles=styles, bias=bias, clamp=256, skip=skip))
        print(f'N{N} cin {cin:4d} {H:4d}^2 cout {cout}: no skip {t0:6.1f} us ({by / t0 * 1e-3:5.0f} GB/s) | skip {t1:6.1f} us ({(by + 4.0 * cout * N * H * H) / t1 * 1e-3:5.0f} GB/s)', flush=True)
