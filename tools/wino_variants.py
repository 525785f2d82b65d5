"""Dev tool: build (here) / time (GPU box) ablated variants of the Winograd kernel (-DWINO_EXP bit mask:
1 no U loads in the K loop, 2 no LDS operand reads, 4 no epilogue, 8 no halo DMA).  Results are wrong by design."""
import sys, os
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, 'pasta-gan-plusplus_amd'))
from torch_utils import custom_ops
custom_ops.verbosity = 'none'
VARIANTS = [int(v) for v in os.environ.get('WINO_VARIANTS', '0,1,2,4,8,3,15').split(',')]
SRC = custom_ops.PLUGIN_SOURCES['conv2d_plugin']

if sys.argv[1] == 'build':
    for v in VARIANTS:
        print(custom_ops.get_plugin(f'wino_exp{v}', sources=SRC, extra_hipcc_flags=[f'-DWINO_EXP={v}'], build_only=True))
    sys.exit(0)

import torch
from torch_utils.ops import conv2d_mfma
v = int(sys.argv[2])
custom_ops.PLUGIN_SOURCES[f'wino_exp{v}'] = SRC
conv2d_mfma._plugin = None
conv2d_mfma._init.__defaults__ = (f'wino_exp{v}',)
_orig = custom_ops.get_plugin
custom_ops.get_plugin = lambda name, **kw: _orig(name, extra_hipcc_flags=[f'-DWINO_EXP={v}'], **kw)

def timeit(fn, n=10):
    fn(); torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n

out = []
for (N, H, cin, cout) in [(8, 256, 128, 128), (8, 256, 64, 128), (8, 64, 512, 512)]:
    x = torch.randn(N, cin, H, H, device='cuda')
    w = torch.randn(cout, cin, 3, 3, device='cuda') / (3 * cin ** 0.5)
    pw = conv2d_mfma.pack_weight(w, winograd=True)
    mw = timeit(lambda: conv2d_mfma.conv2d_forward(x, pw, cout, 3, 3, pad=(1, 1), winograd=True))
    out.append(f'{mw*1e3:8.1f}')
print(f'variant {v:2d}: ' + ' '.join(out) + '  us  (H256 c128->128 | H256 c64->128 | H64 c512->512)', flush=True)
