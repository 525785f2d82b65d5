import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (ROOT, os.path.join(ROOT, 'pasta-gan-plusplus_amd'), os.path.join(ROOT, 'tests', 'golden')):
    sys.path.insert(0, p)
import torch, torch.nn.functional as F
from torch_utils import custom_ops
custom_ops.verbosity = 'none'
from torch_utils.ops import conv2d_mfma16 as M, conv2d_mfma, upfirdn2d
dt = torch.bfloat16
g = torch.Generator().manual_seed(0)
n, cin, cout, h, w = 2, 32, 48, 8, 8
x = torch.randn([n, cin, h, w], generator=g).to(dt)
wt = torch.randn([cout, cin, 3, 3], generator=g)
styles = torch.randn([n, cin], generator=g) + 1
dco = conv2d_mfma.modconv_dcoefs(wt.cuda(), styles.cuda())
wio = wt.transpose(0, 1).contiguous()
out_hw = (2 * h + 1, 2 * w + 1)
ph = M.pack_transposed(wio.cuda(), dt, 2, (0, 0), (h, w), out_hw, styles=styles.cuda(), dcoefs=dco)
y = M.conv_transpose2d_forward(x.cuda(), ph, cout, out_hw)
w16 = ((wt[None] * styles[:, None, :, None, None]) * dco.cpu()[:, :, None, None, None]).to(dt).double()   # [n, cout, cin, 3, 3]
ref = torch.stack([F.conv_transpose2d(x[i:i + 1].double(), w16[i].transpose(0, 1), stride=2)[0] for i in range(n)])
print('modulated transposed rel err', float((y.double().cpu() - ref).abs().max() / ref.abs().max()))
# unmodulated, same shapes
ph0 = M.pack_transposed(wio.cuda(), dt, 2, (0, 0), (h, w), out_hw)
y0 = M.conv_transpose2d_forward(x.cuda(), ph0, cout, out_hw)
ref0 = F.conv_transpose2d(x.double(), wio.to(dt).double(), stride=2)
print('plain transposed rel err', float((y0.double().cpu() - ref0).abs().max() / ref0.abs().max()))
# FIR channels-last
f = upfirdn2d.setup_filter([1, 3, 3, 1]).cuda()
yc = y0
a = upfirdn2d.upfirdn2d(yc, f, padding=[1, 1, 1, 1], gain=4)
b = upfirdn2d.upfirdn2d(yc.float().contiguous(), f, padding=[1, 1, 1, 1], gain=4)
print('fir cl rel err', float((a.float() - b).abs().max() / b.abs().max()), a.shape, a.stride())
noise = torch.randn([2 * h, 2 * w], generator=g).cuda(); bias = torch.randn([cout], generator=g).cuda()
fu = upfirdn2d.upfirdn2d_bias_act(yc, f, padding=[1, 1, 1, 1], gain=4, noise=noise, b=bias, act='lrelu', alpha=0.2, act_gain=1.4, clamp=256)
r = b + noise + bias.reshape(1, -1, 1, 1); r = (torch.where(r > 0, r, r * 0.2) * 1.4).clamp(-256, 256)
print('fused fir rel err', None if fu is None else float((fu.float() - r).abs().max() / r.abs().max()))
