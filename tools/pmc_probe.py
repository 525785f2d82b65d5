"""Dev tool (GPU box): a handful of launches of the two hottest conv shapes, for rocprofv3 --pmc passes (argv[1] = winograd4 | winograd | direct)."""
import sys, os
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, 'pasta-gan-plusplus_amd'))
import torch
from torch_utils import custom_ops
custom_ops.verbosity = 'none'
from torch_utils.ops import conv2d_mfma
dev = 'cuda'
wg = {'winograd4': conv2d_mfma.F4_WIDE, 'winograd4_fp32': 2, 'winograd': 1}.get(sys.argv[1] if len(sys.argv) > 1 else '', 0)      # winograd4: the form the policy hands out (4 = bf16x3 GEMM)
for (N, H, cin, cout) in [(8, 256, 128, 128), (8, 512, 64, 64)]:
    x = torch.randn(N, cin, H, H, device=dev)
    w = torch.randn(cout, cin, 3, 3, device=dev) / (3 * cin ** 0.5)
    pk = conv2d_mfma.pack_weight(w, winograd=wg)
    for _ in range(6):
        y = conv2d_mfma.conv2d_forward(x, pk, cout, 3, 3, pad=(1, 1), winograd=wg)
    torch.cuda.synchronize()
print('done')
