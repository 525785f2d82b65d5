"""Dev tool (GPU box): time 16-bit conv shapes under the kernel's ablation switches (PG_CONV16_DBG), one process."""
import sys, os
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, 'pasta-gan-plusplus_amd'))
import torch
from torch_utils import custom_ops
custom_ops.verbosity = 'none'
SRC = custom_ops.PLUGIN_SOURCES['conv2d_plugin']
custom_ops.get_plugin('conv16_stamps', sources=SRC, extra_hipcc_flags=['-DPG_CONV16_STAMPS=1'], build_only=True)      # the product build has no dev switches
from torch_utils.ops import conv2d_mfma, conv2d_mfma16 as M
custom_ops.PLUGIN_SOURCES['conv16_stamps'] = SRC
_orig = custom_ops.get_plugin
custom_ops.get_plugin = lambda name, **kw: _orig(name, extra_hipcc_flags=['-DPG_CONV16_STAMPS=1'], abi_name='conv2d_plugin', **kw)
conv2d_mfma._init('conv16_stamps')
custom_ops.get_plugin = _orig
dt = torch.bfloat16
shapes = [(4, 32, 32, 1024, 3), (4, 64, 64, 512, 3), (4, 128, 128, 256, 3), (4, 64, 32, 512, 3), (4, 256, 256, 128, 3), (4, 64, 64, 512, 1)]
if len(sys.argv) > 1:
    shapes = [tuple(int(v) for v in a.split(',')) for a in sys.argv[1:]]
for (N, cin, cout, H, K) in shapes:
    x = torch.randn(N, cin, H, H, device='cuda').to(dt).contiguous(memory_format=torch.channels_last)
    w = torch.randn(cout, cin, K, K, device='cuda') / (K * cin ** 0.5)
    pk, _, _ = M.pack_weight(w, dt)
    bias = torch.randn(cout, device='cuda')
    line = f'N{N} {cin}->{cout} {H}^2 k{K}:'
    for dbg in (0, 8, 16, 24, 7, 15, 31):
        os.environ['PG_CONV16_DBG'] = str(dbg)
        run = lambda: M.conv2d_forward(x, pk, cout, K, K, pad=(K // 2, K // 2), bias=bias, act='lrelu', alpha=0.2, gain=1.4, clamp=256)
        run(); torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        torch.cuda._sleep(1000000)
        e0.record()
        for _ in range(10):
            run()
        e1.record(); torch.cuda.synchronize()
        line += f'  dbg{dbg}={e0.elapsed_time(e1) * 100:.0f}us'
    os.environ['PG_CONV16_DBG'] = '0'
    fl = 2.0 * N * cout * H * H * cin * K * K; by = 2.0 * N * H * H * (cin + cout)
    print(line, f' [{fl / 1e9:.0f} GF, {by / 1e6:.0f} MB]', flush=True)
