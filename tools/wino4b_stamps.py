"""Dev tool (GPU box): in-kernel s_memtime stamps of conv2d_wino4b (built with -DWINO4B_EXP=32768+...): every workgroup's third tile, per wave; workgroups are
paired by the CU they ran on (HW_ID / XCC_ID), and the phase intervals of both workgroups of one CU are printed on a common time axis.
Stamps per chunk k (< 8): 6k+0 chunk top | +1 after the halo wait | +2 after barrier A | +3 after transform + first A requests | +4 after barrier B | +5 GEMM issued;
48 K loop done + column half | 49 after the barrier | 50 tail done."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, 'pasta-gan-plusplus_amd'))
from torch_utils import custom_ops
custom_ops.verbosity = 'none'
EXP = int(os.environ.get('WINO4B_STAMP_EXP', '32768'))
SRC = custom_ops.PLUGIN_SOURCES['conv2d_plugin']
name = f'wino4b_exp{EXP}'
custom_ops.get_plugin(name, sources=SRC, extra_hipcc_flags=[f'-DWINO4B_EXP={EXP}'], build_only=True)
if len(sys.argv) > 1 and sys.argv[1] == 'build':
    sys.exit(0)
import torch
from torch_utils.ops import conv2d_mfma
custom_ops.PLUGIN_SOURCES[name] = SRC
_orig = custom_ops.get_plugin
custom_ops.get_plugin = lambda nm, **kw: _orig(nm, extra_hipcc_flags=[f'-DWINO4B_EXP={EXP}'], abi_name='conv2d_plugin', **kw)
conv2d_mfma._init(name)
custom_ops.get_plugin = _orig
for (N, H, cin, cout) in [(8, 256, 128, 128), (8, 512, 64, 64)]:
    x = torch.randn(N, cin, H, H, device='cuda')
    w = torch.randn(cout, cin, 3, 3, device='cuda') / (3 * cin ** 0.5)
    pk = conv2d_mfma.pack_weight(w, winograd=3)
    for _ in range(3):
        y = conv2d_mfma.conv2d_forward(x, pk, cout, 3, 3, pad=(1, 1), winograd=3)
    torch.cuda.synchronize()
    st = y.view(-1)[:512 * 8 * 64 * 2].view(torch.int64).cpu().reshape(512, 8, 64)
    nch = min(cin // 16, 8)
    cus = {}
    for b in range(512):
        hw, xcc = int(st[b, 0, 63]), int(st[b, 0, 62])
        key = (xcc & 15, (hw >> 13) & 7, (hw >> 12) & 1, (hw >> 8) & 15)
        cus.setdefault(key, []).append(b)
    pairs = [v for v in cus.values() if len(v) == 2]
    print(f'N{N} H{H} {cin}->{cout}: {len(cus)} CUs seen, {len(pairs)} with two workgroups', flush=True)
    for pr in pairs[:2]:
        t0 = min(int(st[b, w_, 0]) for b in pr for w_ in range(8))
        for b in pr:
            for w_ in (0, 4):
                r = [int(st[b, w_, i]) - t0 for i in range(51)]
                line = ' | '.join(f'k{k}: top {r[6*k]:6d} halo {r[6*k+1]-r[6*k]:5d} barA {r[6*k+2]-r[6*k+1]:5d} xform {r[6*k+3]-r[6*k+2]:5d} barB {r[6*k+4]-r[6*k+3]:5d} gemm {r[6*k+5]-r[6*k+4]:5d}' for k in range(nch))
                print(f'  wg {b:3d} wave {w_}: {line} || colhalf {r[48]-r[6*(nch-1)+5]:5d} bar {r[49]-r[48]:5d} tail {r[50]-r[49]:5d} end {r[50]:6d}')
