"""Dev tool: build (here) and time (GPU box) variants of csrc/conv2d_up2x3.h compiled with -DUX_EXP=<mask> (1 no MFMAs, 2 no operand split, 4 no halo loads,
8 no weight DMA, 16 no output stores; results wrong by design).    UX_VARIANTS=0,1,2 python tools/up2x3_variants.py build|run"""
import os, sys, statistics
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, 'pasta-gan-plusplus_amd'))
from torch_utils import custom_ops
custom_ops.verbosity = 'none'
VARIANTS = os.environ.get('UX_VARIANTS', '0').split(',')       # "mask" or "mask:NAME=VALUE[:NAME=VALUE]" (extra -D defines, e.g. 0:UE_DEPTH_DEF=3)


def flags_of(v):
    parts = v.split(':')
    return [f'-DUX_EXP={int(parts[0])}'] + [f'-D{d}' for d in parts[1:]]


def name_of(v):
    return 'ux_exp' + ''.join(ch if ch.isalnum() else '_' for ch in v)


SRC = custom_ops.PLUGIN_SOURCES['conv2d_plugin']
for v in VARIANTS:
    custom_ops.get_plugin(name_of(v), sources=SRC, extra_hipcc_flags=flags_of(v), build_only=True)
if sys.argv[1] == 'build':
    sys.exit(0)
import torch
from torch_utils.ops import conv2d_mfma
libs = {}
for v in VARIANTS:
    conv2d_mfma._plugin = None
    custom_ops.PLUGIN_SOURCES[name_of(v)] = SRC
    _orig = custom_ops.get_plugin
    custom_ops.get_plugin = lambda name, _v=v, **kw: _orig(name, extra_hipcc_flags=flags_of(_v), abi_name='conv2d_plugin', **kw)
    libs[v] = conv2d_mfma._init(name_of(v))
    custom_ops.get_plugin = _orig
for (N, cin, cout, H) in [(8, 128, 64, 256), (8, 256, 128, 128), (8, 512, 256, 64), (8, 512, 512, 32)]:
    x = torch.randn(N, cin, H, H, device='cuda')
    w = torch.randn(cout, cin, 3, 3, device='cuda') / (3 * cin ** 0.5)
    ins, outs = torch.rand(N, cin, device='cuda') + 0.5, torch.rand(N, cout, device='cuda') + 0.5
    times = {v: [] for v in VARIANTS}
    packs = {}
    for v in VARIANTS:
        conv2d_mfma._plugin = libs[v]
        packs[v] = conv2d_mfma.pack_up2(w)
    for r in range(6):
        for v in VARIANTS:
            conv2d_mfma._plugin = libs[v]
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            for _ in range(4):
                conv2d_mfma.conv_up2_forward(x, packs[v], cout, in_scale=ins, out_scale=outs)
            e1.record(); torch.cuda.synchronize()
            if r > 0:
                times[v].append(e0.elapsed_time(e1) / 4 * 1e3)
    print(f'N{N} {cin}->{cout} {H}^2: ' + '  '.join(f'[{v}] {statistics.median(times[v]):6.1f}us' for v in VARIANTS), flush=True)
