"""Dev tool (GPU box): repeat launches of conv2d_wino4 on the hot shapes and compare every result with the first one and with the direct
kernel -- an intermittent wrong tile (a missed wait, a race on an LDS buffer) shows up as a non-zero count.
    python tools/wino4_stress.py [iterations]"""
import os
import sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, 'pasta-gan-plusplus_amd'))
import torch
from torch_utils import custom_ops
custom_ops.verbosity = 'none'
from torch_utils.ops import conv2d_mfma
EXP = os.environ.get('WINO4_STRESS_EXP')
if EXP:                                                      # a -DWINO4_EXP=<EXP> build of the plugin (debugging switches of conv2d_wino4.h)
    SRC = custom_ops.PLUGIN_SOURCES['conv2d_plugin']
    custom_ops.PLUGIN_SOURCES[f'wino4_exp{EXP}'] = SRC
    _orig = custom_ops.get_plugin
    custom_ops.get_plugin = lambda name, **kw: _orig(name, extra_hipcc_flags=[f'-DWINO4_EXP={EXP}'], abi_name='conv2d_plugin', **kw)
    conv2d_mfma._init(f'wino4_exp{EXP}')
    custom_ops.get_plugin = _orig
    if len(sys.argv) > 1 and sys.argv[1] == 'build':
        sys.exit(0)

iters = int(sys.argv[1]) if len(sys.argv) > 1 else 200
torch.manual_seed(1)
CASES = [(8, 512, 64, 64, 'plain'), (8, 256, 128, 128, 'plain'), (8, 512, 64, 64, 'mod'), (8, 256, 128, 256, 'spade'), (8, 256, 128, 128, 'res'),
         (16, 256, 128, 128, 'plain'), (1, 256, 128, 128, 'plain'), (8, 128, 256, 256, 'mod'), (8, 32, 512, 512, 'mod'), (3, 200, 64, 192, 'res')]
for (N, H, cin, cout, mode) in CASES:
    x = torch.randn(N, cin, H, H, device='cuda')
    w = torch.randn(cout, cin, 3, 3, device='cuda') / (3 * cin ** 0.5)
    kw = {}
    if mode == 'mod':
        kw = dict(in_scale=torch.rand(N, cin, device='cuda') + 0.5, out_scale=torch.rand(N, cout, device='cuda') + 0.5, noise=torch.randn(H, H, device='cuda'),
                  bias=torch.randn(cout, device='cuda'), act='lrelu', alpha=0.2, gain=1.4, clamp=256.0)
    elif mode == 'res':
        kw = dict(bias=torch.randn(cout, device='cuda'), residual=torch.randn(N, cout, H, H, device='cuda'), act='relu')
    if mode == 'spade':
        c = cout // 2
        sp = (torch.randn(N, c, H, H, device='cuda'), torch.randn(N, c, device='cuda'), torch.rand(N, c, device='cuda') + 0.5)
        pk4 = conv2d_mfma.pack_spade_gamma_beta(w[:c].contiguous(), w[c:].contiguous(), winograd=2)
        pk0 = conv2d_mfma.pack_spade_gamma_beta(w[:c].contiguous(), w[c:].contiguous(), winograd=0)
        kw = dict(spade=sp, act='lrelu', alpha=0.2, gain=1.4, clamp=3.0)
    else:
        pk4, pk0 = conv2d_mfma.pack_weight(w, winograd=2), conv2d_mfma.pack_weight(w)
    ref = conv2d_mfma.conv2d_forward(x, pk0, cout, 3, 3, pad=(1, 1), **kw)
    bad, worst = 0, 0.0
    other = torch.randn(N, cin, H, H, device='cuda')
    wo = conv2d_mfma.pack_weight(torch.randn(64, cin, 3, 3, device='cuda'))
    for it in range(iters):
        if it % 3 == 1:                                      # some other work in between: different cache / clock state
            conv2d_mfma.conv2d_forward(other, wo, 64, 3, 3, pad=(1, 1))
        y = conv2d_mfma.conv2d_forward(x, pk4, cout, 3, 3, pad=(1, 1), winograd=2, **kw)
        e = float((y - ref).abs().max())
        worst = max(worst, e)
        if e > 1e-3:
            bad += 1
            if bad <= 3:
                idx = torch.nonzero((y - ref).abs() > 1e-3)
                print(f'   iteration {it}: max |d| {e:.3e}, {idx.shape[0]} elements off; first {idx[0].tolist()}, last {idx[-1].tolist()}', flush=True)
    print(f'N{N} {cin}->{cout} {H}x{H} {mode}: {bad} bad launches of {iters}; worst |F(4x4) - direct| {worst:.3e}', flush=True)
