"""Dev tool (GPU box): the fused-x up = 2 kernel (csrc/conv2d_up2f16.h) on the config-5 shapes beside the composite four-phase launch, and under its
ablation switches (diagnostic build, PG_CONV16_DBG bits: 1 no stores, 4 no MFMA, 8 no epilogue, 64 no x filter, 128 no halo DMA).
    python tools/up2f_probe.py [N,cin,cout,H ...]"""
import sys, os
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, 'pasta-gan-plusplus_amd'))
import torch
from torch_utils import custom_ops
custom_ops.verbosity = 'none'
SRC = custom_ops.PLUGIN_SOURCES['conv2d_plugin']
FLAGS = ['-DPG_CONV16_STAMPS=1'] + [a for a in os.environ.get('UP2F_FLAGS', '').split() if a]
custom_ops.get_plugin('conv16_stamps', sources=SRC, extra_hipcc_flags=FLAGS, build_only=True)
from torch_utils.ops import conv2d_mfma, conv2d_mfma16 as M
custom_ops.PLUGIN_SOURCES['conv16_stamps'] = SRC
_orig = custom_ops.get_plugin
custom_ops.get_plugin = lambda name, **kw: _orig(name, extra_hipcc_flags=FLAGS, abi_name='conv2d_plugin', **kw)
conv2d_mfma._init('conv16_stamps')
custom_ops.get_plugin = _orig
from training import networks as PN
from torch_utils.ops import upfirdn2d
dt = torch.bfloat16
shapes = [(4, 64, 32, 512), (4, 128, 64, 256), (4, 256, 128, 128), (4, 512, 256, 64), (4, 1024, 512, 32)]
if len(sys.argv) > 1:
    shapes = [tuple(int(v) for v in a.split(',')) for a in sys.argv[1:]]
f = upfirdn2d.setup_filter([1, 3, 3, 1]).cuda()


def timed(run, reps=10):
    run(); torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    torch.cuda._sleep(1000000)
    e0.record()
    for _ in range(reps):
        run()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) * 1e3 / reps


for (N, cin, cout, H) in shapes:
    x = torch.randn(N, cin, H, H, device='cuda').to(dt).contiguous(memory_format=torch.channels_last)
    wt = (torch.randn(cin, cout, 3, 3, device='cuda') / (3 * cin ** 0.5)).contiguous()
    styles = torch.rand(N, cin, device='cuda') + 0.5
    shared = cout * 24 > 2 * H * H
    fy, fx = PN._separable_taps(f)
    stack = PN._up2_fused_weights(wt, fy)
    pk, per, _ = M.pack_weight(stack, dt, transpose_oi=True, styles=None if shared else styles)
    comp = torch.cat(list(PN._up2_composite_phases(wt, f).values()), dim=1).contiguous()
    pkc, perc, _ = M.pack_weight(comp, dt, transpose_oi=True, styles=None if shared else styles)
    bias = torch.randn(cout, device='cuda')
    noise = torch.randn(1, 4, H, H, device='cuda')
    ep = dict(bias=bias, act='lrelu', alpha=0.2, gain=1.4, clamp=256)
    y = torch.empty([N, cout, 2 * H, 2 * H], dtype=dt, device='cuda', memory_format=torch.channels_last)
    os.environ['PG_CONV16_DBG'] = '0'
    t_comp = timed(lambda: M.conv2d_forward(x, pkc, cout, 3, 3, pad=(1, 1), out_hw=(H, H), y=y, sample_stride=perc, noise=noise, phases=True, **ep))
    line = f'N{N} {cin}->{cout} {H}^2: composite {t_comp:.0f}us | fused-x'
    for dbg in [int(v) for v in os.environ.get("UP2F_DBGS", "0,1,8,4,12,128,136,140").split(",")]:
        os.environ['PG_CONV16_DBG'] = str(dbg)
        line += f'  dbg{dbg}={timed(lambda: M.conv_up2_fused(x, pk, cout, [2 * float(v) for v in fx], sample_stride=per, noise=noise, **ep)):.0f}'
    os.environ['PG_CONV16_DBG'] = '0'
    by = 2.0 * N * H * H * (cin + 4 * cout)
    print(line, f' [honest {2.0 * N * cout * H * H * cin * 9 / 1e9:.0f} GF, {by / 1e6:.0f} MB]', flush=True)
