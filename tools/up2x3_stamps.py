"""Dev tool (GPU box): cycle totals of the multiplying waves of conv2d_up2x3 (built with -DUX_EXP=32): per round the cycles in the operand reads + MFMAs,
at the barrier, and per tile in the epilogue.   python tools/up2x3_stamps.py [build]"""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, 'pasta-gan-plusplus_amd'))
from torch_utils import custom_ops
custom_ops.verbosity = 'none'
SRC = custom_ops.PLUGIN_SOURCES['conv2d_plugin']
custom_ops.get_plugin('ux_exp32', sources=SRC, extra_hipcc_flags=['-DUX_EXP=32'], build_only=True)
if len(sys.argv) > 1 and sys.argv[1] == 'build':
    sys.exit(0)
import torch
from torch_utils.ops import conv2d_mfma
custom_ops.PLUGIN_SOURCES['ux_exp32'] = SRC
_orig = custom_ops.get_plugin
custom_ops.get_plugin = lambda name, **kw: _orig(name, extra_hipcc_flags=['-DUX_EXP=32'], abi_name='conv2d_plugin', **kw)
conv2d_mfma._init('ux_exp32')
custom_ops.get_plugin = _orig
for (N, cin, cout, H) in [(8, 128, 64, 256), (8, 512, 256, 64)]:
    x = torch.randn(N, cin, H, H, device='cuda')
    w = torch.randn(cout, cin, 3, 3, device='cuda') / (3 * cin ** 0.5)
    ins, outs = torch.rand(N, cin, device='cuda') + 0.5, torch.rand(N, cout, device='cuda') + 0.5
    packs = conv2d_mfma.pack_up2(w)
    for _ in range(3):
        y = conv2d_mfma.conv_up2_forward(x, packs, cout, in_scale=ins, out_scale=outs)
    torch.cuda.synchronize()
    base = y.storage_offset()
    raw = torch.as_strided(y, [64], [1], base).contiguous().view(torch.int64).cpu()
    print(f'N{N} {cin}->{cout} {H}^2:')
    for wv in range(4):
        mm, bar, ep, rounds = (int(v) for v in raw[wv * 4: wv * 4 + 4])
        tiles = max(1, rounds // (cin // 16))
        print(f'  wave {wv}: {rounds} rounds, {tiles} tiles: operand reads + MFMAs {mm / max(rounds, 1):8.0f} cycles per round, barrier wait {bar / max(rounds, 1):8.0f} per round, epilogue {ep / tiles:8.0f} per tile')
