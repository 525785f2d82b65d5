"""Dev tool: build (here) and time (GPU box) variants of csrc/conv2d_stem7x3.h compiled with -DS7_EXP=<mask> (1 no record updates, 2 no output stores, 4 no MFMAs;
results wrong by design) and extra defines.    S7_VARIANTS=0,1,2,0:NAME=VALUE python tools/stem7_variants.py build|run"""
import os, sys, statistics, math
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, 'pasta-gan-plusplus_amd'))
from torch_utils import custom_ops
custom_ops.verbosity = 'none'
VARIANTS = os.environ.get('S7_VARIANTS', '0').split(',')
SRC = custom_ops.PLUGIN_SOURCES['conv2d_plugin']
flags_of = lambda v: [f'-DS7_EXP={int(v.split(":")[0])}'] + [f'-D{d}' for d in v.split(':')[1:]]
name_of = lambda v: 's7_exp' + ''.join(ch if ch.isalnum() else '_' for ch in v)
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
for (N, H) in [(8, 512), (16, 512)]:
    x = torch.rand(N, 3, H, H, device='cuda') * 2 - 1
    w = torch.randn(64, 3, 7, 7, device='cuda')
    b = torch.randn(64, device='cuda')
    times = {v: [] for v in VARIANTS}
    packs = {}
    for v in VARIANTS:
        conv2d_mfma._plugin = libs[v]
        packs[v] = conv2d_mfma.pack_stem7(w, scale=1 / math.sqrt(147))
    for r in range(6):
        for v in VARIANTS:
            conv2d_mfma._plugin = libs[v]
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            for _ in range(4):
                conv2d_mfma.conv_stem7_forward(x, packs[v], 64, bias=b, act='relu', gain=math.sqrt(2))
            e1.record(); torch.cuda.synchronize()
            if r > 0:
                times[v].append(e0.elapsed_time(e1) / 4 * 1e3)
    print(f'N{N} 3->64 {H}^2: ' + '  '.join(f'[{v}] {statistics.median(times[v]):6.1f}us' for v in VARIANTS), flush=True)
