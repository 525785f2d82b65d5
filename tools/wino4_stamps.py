"""Dev tool (GPU box): in-kernel cycle stamps of conv2d_wino4 (built with -DWINO4_EXP=128): workgroup 0's second tile, per wave.
Stamps (cycles since the tile's start): 0 tile start | 1 chunk 2: after barrier A | 2 transform done | 3 after barrier B | 4 GEMM done |
5 K loop done | 6 column inverse done | 7 round 0 exchange written + operands requested | 8 after barrier | 9 after the vmcnt(0) |
10 round 0 finished | 11..14 after the closing barrier of rounds 0..3."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, 'pasta-gan-plusplus_amd'))
from torch_utils import custom_ops
custom_ops.verbosity = 'none'
SRC = custom_ops.PLUGIN_SOURCES['conv2d_plugin']
custom_ops.get_plugin('wino4_exp128', sources=SRC, extra_hipcc_flags=['-DWINO4_EXP=128'], build_only=True)
if len(sys.argv) > 1 and sys.argv[1] == 'build':
    sys.exit(0)
import torch
from torch_utils.ops import conv2d_mfma
ALGO = int(os.environ.get('WINO_ALGO', '2'))      # 4 = the X3 form
custom_ops.PLUGIN_SOURCES['wino4_exp128'] = SRC
_orig = custom_ops.get_plugin
custom_ops.get_plugin = lambda name, **kw: _orig(name, extra_hipcc_flags=['-DWINO4_EXP=128'], abi_name='conv2d_plugin', **kw)
conv2d_mfma._init('wino4_exp128')
custom_ops.get_plugin = _orig
for (N, H, cin, cout) in [(8, 256, 128, 128), (8, 512, 64, 64)]:
    x = torch.randn(N, cin, H, H, device='cuda')
    w = torch.randn(cout, cin, 3, 3, device='cuda') / (3 * cin ** 0.5)
    pk = conv2d_mfma.pack_weight(w, winograd=ALGO)
    for _ in range(3):
        y = conv2d_mfma.conv2d_forward(x, pk, cout, 3, 3, pad=(1, 1), winograd=ALGO)
    torch.cuda.synchronize()
    st = y.view(-1)[:12 * 16].view(torch.int32).cpu().reshape(12, 16)
    print(f'N{N} H{H} {cin}->{cout}: stamps per wave (cycles since tile start)')
    for wv in range(12):
        print(f'  wave {wv:2d}: ' + ' '.join(f'{int(v):7d}' for v in st[wv][:15]))
    if ALGO == 4:
        xs = y.view(-1)[192:192 + 12 * 20].view(torch.int32).cpu().reshape(12, 20)
        print('  X3 GEMM phase of chunk 2 (cycles since the phase start; per group: split done / A words home / MFMAs issued):')
        for wv in range(12):
            print(f'  wave {wv:2d}: ' + ' | '.join(' '.join(f'{int(xs[wv][1 + 3 * g + j] - xs[wv][0]):5d}' for j in range(3)) for g in range(6)))
