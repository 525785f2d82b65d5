"""Dev tool (GPU box): would the fp32 `up = 2` layers of config 2 pay on the bf16 matrix pipe by three-term operand splitting?  The transposed convolution's four
phases (2x2 / 2x1 / 1x2 / 1x1 taps) as launches of the 16-bit kernel over 6 Cin channels -- [x1,x2,x1,x2,x3,x1] against [w1,w1,w2,w2,w1,w3] -- with float32 output,
beside conv2d_up2 (fp32 MFMA).  Times the four launches alone (the operand split of x is priced at 16 bytes per element and 5 TB/s) and prints the error of both
against float64.
    python tools/up2_bf16x3_probe.py"""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, 'pasta-gan-plusplus_amd'))
import torch
from torch_utils import custom_ops
custom_ops.verbosity = 'none'
from torch_utils.ops import conv2d_mfma, conv2d_mfma16 as M


def timeit(fn, n=6):
    for _ in range(2): fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n * 1e3


def split3(t):
    trunc = lambda v: (v.view(torch.int32) & -65536).view(torch.float32)
    hi = trunc(t); r = t - hi; mid = trunc(r); lo = trunc(r - mid)
    return hi, mid, lo


for (n, cin, cout, h) in [(8, 512, 512, 32), (8, 512, 256, 64), (8, 256, 128, 128), (8, 128, 64, 256)]:
    x = torch.randn(n, cin, h, h, device='cuda'); w = torch.randn(cout, cin, 3, 3, device='cuda') / (3 * cin ** 0.5)
    s_in = torch.rand(n, cin, device='cuda') + 0.5; s_out = torch.rand(n, cout, device='cuda') + 0.5
    packs = conv2d_mfma.pack_up2(w)
    t32 = timeit(lambda: conv2d_mfma.conv_up2_forward(x, packs, cout, in_scale=s_in, out_scale=s_out))
    xs = split3(x * s_in[:, :, None, None]); ws = split3(w)
    x6 = torch.cat([xs[i] for i in (0, 1, 0, 1, 2, 0)], dim=1).to(torch.bfloat16).contiguous(memory_format=torch.channels_last)
    w6 = torch.cat([ws[i] for i in (0, 0, 1, 1, 0, 2)], dim=1)                       # [Cout, 6 Cin, 3, 3]
    oh = 2 * h + 1
    phases = M.pack_transposed(w6.transpose(0, 1).contiguous(), torch.bfloat16, 2, (0, 0), (h, h), (oh, oh))
    pitch = (oh + 3) // 4 * 4
    buf = torch.empty([n, cout, oh, pitch], device='cuda'); y = buf[:, :, :, :oh]

    def run():
        for ph, packed, per in phases:
            M.conv2d_forward(x6, packed, cout, len(ph['ky']), len(ph['kx']), stride=1, pad=ph['pad'], out_hw=ph['out_hw'], y=y, out_step=(2, 2), out_off=ph['off'],
                             sample_stride=per, out_dtype=torch.float32, out_scale=s_out)
    t16 = timeit(run)
    split_us = x.numel() * 16 / 5e12 * 1e6
    line = f'N{n} {cin}->{cout} {h}^2: conv2d_up2 (fp32) {t32:7.1f} us | four 16-bit phases over 6 Cin {t16:7.1f} us + split ~{split_us:5.1f} = {(t16 + split_us) / t32:4.2f} of fp32'
    if n * cin * h * h <= 8 * 512 * 64 * 64:
        ref = torch.nn.functional.conv_transpose2d(x.double() * s_in.double()[:, :, None, None], w.double().transpose(0, 1), stride=2) * s_out.double()[:, :, None, None]
        a = conv2d_mfma.conv_up2_forward(x, packs, cout, in_scale=s_in, out_scale=s_out).double(); run(); b = y.double()
        sc = float(ref.abs().max())
        line += f' | err / max|y|: fp32 kernel {float((a - ref).abs().max()) / sc:.2e}, bf16x3 {float((b - ref).abs().max()) / sc:.2e}'
    print(line, flush=True)
