"""Dev tool (GPU box): where one wave of the 16-bit conv kernel spends its cycles (s_memtime stamps, PG_CONV16_DBG=32)."""
import sys, os, ctypes, collections
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, 'pasta-gan-plusplus_amd'))
import torch
from torch_utils import custom_ops
custom_ops.verbosity = 'none'
SRC = custom_ops.PLUGIN_SOURCES['conv2d_plugin']
custom_ops.get_plugin('conv16_stamps', sources=SRC, extra_hipcc_flags=['-DPG_CONV16_STAMPS=1'], build_only=True)      # the product build compiles the stamps out
if len(sys.argv) > 1 and sys.argv[1] == 'build':
    sys.exit(0)
from torch_utils.ops import conv2d_mfma, conv2d_mfma16 as M
custom_ops.PLUGIN_SOURCES['conv16_stamps'] = SRC
_orig = custom_ops.get_plugin
custom_ops.get_plugin = lambda name, **kw: _orig(name, extra_hipcc_flags=['-DPG_CONV16_STAMPS=1'], abi_name='conv2d_plugin', **kw)
conv2d_mfma._init('conv16_stamps')
custom_ops.get_plugin = _orig
lib = M._init()
buf = torch.zeros(4096, dtype=torch.int64, device='cuda')
lib.pg_conv2d16_debug_stamps.argtypes = [ctypes.c_void_p]
lib.pg_conv2d16_debug_stamps(ctypes.c_void_p(buf.data_ptr()))
dt = torch.bfloat16
names = {1: 'loop-top', 2: 'after-wait', 3: 'after-barrier', 4: 'after-issue', 5: 'after-compute', 6: 'after-epilogue'}
for arg in [a for a in sys.argv[1:] if a != 'build'] or ['4,32,32,1024,3']:
    N, cin, cout, H, K = (int(v) for v in arg.split(','))
    x = torch.randn(N, cin, H, H, device='cuda').to(dt).contiguous(memory_format=torch.channels_last)
    w = torch.randn(cout, cin, K, K, device='cuda') / (K * cin ** 0.5)
    pk, _, _ = M.pack_weight(w, dt)
    bias = torch.randn(cout, device='cuda')
    for wv in (0, 5, 8):
        os.environ['PG_CONV16_DBG'] = str(32 | (wv << 8))
        buf.zero_()
        M.conv2d_forward(x, pk, cout, K, K, pad=(K // 2, K // 2), bias=bias, act='lrelu', alpha=0.2, gain=1.4, clamp=256)
        torch.cuda.synchronize()
        st = [(int(v) >> 8, int(v) & 255) for v in buf.cpu().tolist() if v != 0]
        seg = collections.defaultdict(list)
        for (t0, a), (t1, b) in zip(st[:-1], st[1:]):
            seg[(a, b)].append(t1 - t0)
        print(f'shape {arg} wave {wv}: {len(st)} stamps, span {st[-1][0] - st[0][0]} ticks')
        for (a, b), v in sorted(seg.items()):
            v2 = v[len(v) // 4:]
            print(f'   {names[a]:>15} -> {names[b]:<15} n={len(v):4d} mean={sum(v2) / len(v2):8.0f} min={min(v2):6d} max={max(v2):7d}')
            if wv >= 8 and (a, b) == (3, 4):
                print('      per pass:', ' '.join(str(x) for x in v))
    os.environ['PG_CONV16_DBG'] = '0'
