"""Dev tool (GPU box): the low-resolution 3x3 layers of the 16-bit stack (8^2 ... 64^2, 512 / 1024 channels, shared weights) under different split-K
factors: microseconds per layer (main launch + finish pass), against the time the packed weights alone need at 5 TB/s.
    python tools/small16_probe.py [N,cin,cout,H[,phases] ...]"""
import sys, os
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, 'pasta-gan-plusplus_amd'))
import torch
from torch_utils import custom_ops
custom_ops.verbosity = 'none'
from torch_utils.ops import conv2d_mfma16 as M

dt = torch.bfloat16
shapes = [(4, 1024, 1024, 8, 0), (4, 1024, 1024, 16, 0), (4, 1024, 1024, 32, 0), (4, 512, 512, 64, 0), (4, 1024, 1024, 8, 1), (4, 1024, 1024, 16, 1)]
if len(sys.argv) > 1:
    shapes = [tuple(int(v) for v in a.split(',')) for a in sys.argv[1:]]
    shapes = [s if len(s) == 5 else s + (0,) for s in shapes]


def timed(run, reps=20):
    run(); torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    torch.cuda._sleep(1000000)
    e0.record()
    for _ in range(reps):
        run()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) * 1e3 / reps


lib = M._init()
plan = lib.pg_conv2d16_splitk_plan
for (N, cin, cout, H, phases) in shapes:
    x = torch.randn(N, cin, H, H, device='cuda').to(dt).contiguous(memory_format=torch.channels_last)
    ct = 4 * cout if phases else cout
    wt = (torch.randn(ct, cin, 3, 3, device='cuda') / (3 * cin ** 0.5)).contiguous()
    pk, per, _ = M.pack_weight(wt, dt)
    dco = torch.rand(N, cout, device='cuda') + 0.5
    bias = torch.randn(cout, device='cuda')
    ep = dict(bias=bias, act='lrelu', alpha=0.2, gain=1.4, clamp=256, out_scale=dco)
    kw = dict(phases=True, out_hw=(H, H)) if phases else {}
    y = torch.empty([N, cout, 2 * H, 2 * H] if phases else [N, cout, H, H], dtype=dt, device='cuda', memory_format=torch.channels_last)
    default = plan(N, cin, H, H, ct, 3, 3, 1)
    line = f'N{N} {cin}->{ct} {H}^2: weights {pk.numel() * 2 / 5e6:5.1f} us at 5 TB/s | plan {default} |'
    for k in (1, 2, 4, 8, 16, 32):
        if (cin // 16) % k:
            continue
        lib.pg_conv2d16_splitk_plan = lambda *a, k=k: k
        try:
            line += f' k{k}={timed(lambda: M.conv2d_forward(x, pk, cout, 3, 3, pad=(1, 1), y=y, sample_stride=per, **kw, **ep)):.0f}'
        except Exception as e:
            line += f' k{k}=ERR({str(e)[:40]})'
    lib.pg_conv2d16_splitk_plan = plan
    print(line, flush=True)
