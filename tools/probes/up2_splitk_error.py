import os, sys
sys.path.insert(0, '/root/repo/pasta-gan-plusplus_amd')
import torch
from torch_utils import custom_ops
custom_ops.verbosity = 'none'
from torch_utils.ops import conv2d_mfma
lib = conv2d_mfma._init().lib
plan = lib.pg_conv2d_up2_splitk_plan
for (n, cin, cout, h) in [(4, 512, 512, 8), (2, 512, 512, 8), (1, 512, 512, 16), (2, 512, 512, 16), (1, 512, 512, 32)]:
    g = torch.Generator().manual_seed(1)
    x = torch.randn(n, cin, h, h, generator=g).cuda(); w = (torch.randn(cout, cin, 3, 3, generator=g) / (3 * cin ** 0.5)).cuda()
    s_in = (torch.randn(n, cin, generator=g)).cuda(); s_out = (torch.rand(n, cout, generator=g) + 0.5).cuda()
    packs = conv2d_mfma.pack_up2(w)
    ref = torch.nn.functional.conv_transpose2d(x.double() * s_in.double()[:, :, None, None], w.double().transpose(0, 1), stride=2) * s_out.double()[:, :, None, None]
    k = plan(n, cin, h, h, cout)
    a = conv2d_mfma.conv_up2_forward(x, packs, cout, in_scale=s_in, out_scale=s_out).double()
    lib.pg_conv2d_up2_splitk_plan = lambda *args: 1
    b = conv2d_mfma.conv_up2_forward(x, packs, cout, in_scale=s_in, out_scale=s_out).double()
    lib.pg_conv2d_up2_splitk_plan = plan
    sc = float(ref.abs().max())
    print(f'N{n} {cin}->{cout} {h}^2 plan {k}: split err {float((a - ref).abs().max()) / sc:.2e} rms {float((a - ref).pow(2).mean().sqrt()) / sc:.2e} | one share err {float((b - ref).abs().max()) / sc:.2e} rms {float((b - ref).pow(2).mean().sqrt()) / sc:.2e}')
