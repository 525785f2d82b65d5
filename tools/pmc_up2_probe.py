"""Dev tool (GPU box): a handful of launches of the `up = 2` layer (argv[1] = x3 | fp32) for rocprofv3 --pmc / --kernel-trace passes (tools/pmc_up2.sh)."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, 'pasta-gan-plusplus_amd'))
import torch
from torch_utils import custom_ops
custom_ops.verbosity = 'none'
from torch_utils.ops import conv2d_mfma
conv2d_mfma.UP2_X3 = (sys.argv[1] if len(sys.argv) > 1 else 'x3') == 'x3'
for (N, cin, cout, H) in [(8, 128, 64, 256), (8, 512, 256, 64)]:
    x = torch.randn(N, cin, H, H, device='cuda')
    w = torch.randn(cout, cin, 3, 3, device='cuda') / (3 * cin ** 0.5)
    ins, outs = torch.rand(N, cin, device='cuda') + 0.5, torch.rand(N, cout, device='cuda') + 0.5
    packs = conv2d_mfma.pack_up2(w)
    for _ in range(8):
        y = conv2d_mfma.conv_up2_forward(x, packs, cout, in_scale=ins, out_scale=outs)
    torch.cuda.synchronize()
print('done')
