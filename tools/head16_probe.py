"""Dev tool (GPU box): the streaming 1x1 head (ToRGB of the 16-bit stack, pg_conv1x1_small16) alone on the config-5 shapes: microseconds per launch and the
HBM rate of its algorithmic bytes (2 Cin per pixel in, 4 Cout out, 4 Cout / 4 of the half-resolution skip).
    python tools/head16_probe.py [N,cin,H ...]"""
import sys, os
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, 'pasta-gan-plusplus_amd'))
import torch
from torch_utils import custom_ops
custom_ops.verbosity = 'none'
from torch_utils.ops import conv2d_mfma16 as M

dt = torch.bfloat16
shapes = [(4, 32, 1024), (4, 64, 512), (4, 128, 256), (4, 256, 128), (4, 512, 64), (4, 1024, 32), (4, 1024, 16), (4, 1024, 8)]
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
    x = torch.randn(N, cin, H, H, device='cuda').to(dt).contiguous(memory_format=torch.channels_last)
    w = torch.randn(3, cin, device='cuda') / cin ** 0.5
    styles = torch.rand(N, cin, device='cuda') + 0.5
    bias = torch.randn(3, device='cuda')
    skip = torch.randn(N, 3, H // 2, H // 2, device='cuda')
    full = torch.randn(N, 3, H, H, device='cuda')
    by = N * H * H * (2.0 * cin + 12)
    t0 = timed(lambda: M.conv1x1_small(x, w, styles=styles, bias=bias, clamp=256))
    t1 = timed(lambda: M.conv1x1_small(x, w, styles=styles, bias=bias, clamp=256, skip=skip, skip_up2=True))
    t2 = timed(lambda: M.conv1x1_small(x, w, styles=styles, bias=bias, clamp=256, skip=full))
    print(f'N{N} cin {cin:4d} {H:4d}^2: no skip {t0:6.1f} us ({by / t0 * 1e-3:5.0f} GB/s) | up2 skip {t1:6.1f} us ({(by + 3.0 * N * H * H) / t1 * 1e-3:5.0f} GB/s) | '
          f'same-size skip {t2:6.1f} us ({(by + 12.0 * N * H * H) / t2 * 1e-3:5.0f} GB/s)', flush=True)
