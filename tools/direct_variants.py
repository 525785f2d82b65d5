"""Dev tool: build (here) and time (GPU box, one process, round-robin, median) ablated builds of the direct conv kernel
(-DDIRECT_EXP=<mask>: 1 no output stores, 2 no halo DMA; results wrong by design)."""
import sys, os, statistics
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, 'pasta-gan-plusplus_amd'))
from torch_utils import custom_ops
custom_ops.verbosity = 'none'
VARIANTS = [int(v) for v in os.environ.get('DIRECT_VARIANTS', '0,1,2,3').split(',')]
SRC = custom_ops.PLUGIN_SOURCES['conv2d_plugin']
plugins = {v: custom_ops.get_plugin(f'direct_exp{v}', sources=SRC, extra_hipcc_flags=[f'-DDIRECT_EXP={v}'], build_only=True) for v in VARIANTS}
if sys.argv[1] == 'build':
    print(plugins); sys.exit(0)
import torch
from torch_utils.ops import conv2d_mfma
libs = {}
for v in VARIANTS:
    conv2d_mfma._plugin = None
    custom_ops.PLUGIN_SOURCES[f'direct_exp{v}'] = SRC
    _orig = custom_ops.get_plugin
    custom_ops.get_plugin = lambda name, _v=v, **kw: _orig(name, extra_hipcc_flags=[f'-DDIRECT_EXP={_v}'], **kw)
    libs[v] = conv2d_mfma._init(f'direct_exp{v}')
    custom_ops.get_plugin = _orig
SHAPES = [(8, 512, 64, 64, 1, 1), (8, 512, 128, 64, 1, 1), (8, 256, 128, 128, 1, 1), (8, 512, 3, 64, 7, 1), (8, 513, 64, 128, 3, 2), (8, 256, 128, 64, 2, 1)]
for (N, H, cin, cout, k, stride) in SHAPES:
    x = torch.randn(N, cin, H, H, device='cuda')
    w = torch.randn(cout, cin, k, k, device='cuda') / (k * cin ** 0.5)
    times = {v: [] for v in VARIANTS}
    packed = {}
    for v in VARIANTS:
        conv2d_mfma._plugin = libs[v]
        packed[v] = conv2d_mfma.pack_weight(w)
    pad = k // 2 if stride == 1 else 0
    for r in range(8):
        for v in VARIANTS:
            conv2d_mfma._plugin = libs[v]
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            for _ in range(4):
                y = conv2d_mfma.conv2d_forward(x, packed[v], cout, k, k, stride=stride, pad=(pad, pad))
            e1.record(); torch.cuda.synchronize()
            if r > 0:
                times[v].append(e0.elapsed_time(e1) / 4 * 1e3)
    print(f'N{N} {cin}->{cout} {H}x{H} k{k} s{stride}: ' + '  '.join(f'[{v}] {statistics.median(times[v]):7.1f}us' for v in VARIANTS), flush=True)
