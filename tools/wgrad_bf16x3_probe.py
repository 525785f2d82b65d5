"""Dev tool (GPU box), VERDICT r4 item 7 (exploratory): the fp32 weight gradient on the bf16 matrix pipe by three-term operand splitting.
x = x1 + x2 + x3, dy = d1 + d2 + d3 (bf16 terms by truncation: exact), dw = the six products (d1,x1) (d1,x2) (d2,x1) (d2,x2) (d1,x3) (d3,x1), fp32 accumulate.
Prototype form: the six products are ONE launch of the existing 16-bit weight-gradient kernel over a batch of 6 N images -- the batch axis is the GEMM's K axis,
so stacking [d1,d1,d2,d2,d1,d3] against [x1,x2,x1,x2,x3,x1] sums the products inside the kernel's own fp32 accumulation.  Prints, per shape: the fp32 kernel,
the 6N-batch 16-bit launch alone, the split passes as torch ops (what a fused splitting pass would have to beat: its bytes at 5 TB/s are printed too), and
the errors of both against float64.
    python tools/wgrad_bf16x3_probe.py"""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, 'pasta-gan-plusplus_amd'))
import torch
from torch_utils import custom_ops
custom_ops.verbosity = 'none'
from torch_utils.ops import conv2d_mfma, conv2d_mfma16


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


def stack6(t, order):
    parts = split3(t)
    return torch.cat([parts[i] for i in order], dim=0).to(torch.bfloat16).contiguous(memory_format=torch.channels_last)


for (n, cin, cout, hw) in [(4, 64, 64, 512), (4, 128, 128, 256), (4, 128, 256, 256), (4, 256, 256, 128), (4, 512, 512, 64), (4, 512, 512, 32)]:
    x = torch.randn(n, cin, hw, hw, device='cuda'); dy = torch.randn(n, cout, hw, hw, device='cuda') * 0.01
    shape = (cout, cin, 3, 3)
    fl = 2.0 * n * hw * hw * cin * cout * 9
    t32 = timeit(lambda: conv2d_mfma.weight_gradient(x, dy, shape, (1, 1)))
    xs, ds = stack6(x, (0, 1, 0, 1, 2, 0)), stack6(dy, (0, 0, 1, 1, 0, 2))
    t16 = timeit(lambda: conv2d_mfma16.weight_gradient(xs, ds, shape, (1, 1)))
    tsplit = timeit(lambda: (stack6(x, (0, 1, 0, 1, 2, 0)), stack6(dy, (0, 0, 1, 1, 0, 2))), n=3)
    x1 = x.to(torch.bfloat16).contiguous(memory_format=torch.channels_last); d1 = dy.to(torch.bfloat16).contiguous(memory_format=torch.channels_last)
    t16_1 = timeit(lambda: conv2d_mfma16.weight_gradient(x1, d1, shape, (1, 1)))
    os.environ['PG_WGRAD_BF16X3'] = '1'
    treal = timeit(lambda: conv2d_mfma.weight_gradient(x, dy, shape, (1, 1)))
    real = conv2d_mfma.weight_gradient(x, dy, shape, (1, 1))
    os.environ['PG_WGRAD_BF16X3'] = '0'
    lib = conv2d_mfma._init().lib
    from torch_utils.ops import _native as nat
    x3 = torch.empty([3, n, hw, hw, cin], dtype=torch.bfloat16, device='cuda')
    tsp = timeit(lambda: lib.pg_split3_bf16_cl(nat.ptr(x), nat.ptr(x3), n, cin, hw * hw, nat.stream_of(x)))
    ideal = (x.numel() + dy.numel()) * (4 + 6) / 5e12 * 1e6           # a fused pass writing THREE planes per operand (a kernel that reads planes by index needs no duplicates)
    line = f'N{n} {cin}->{cout} {hw}^2: fp32 {t32:7.1f} us ({fl / t32 * 1e-6:5.1f} TF) | 16-bit x1 {t16_1:6.1f} | 16-bit 6N {t16:7.1f} | split (torch ops) {tsplit:7.1f}, fused-pass bytes at 5 TB/s {ideal:6.1f} | PRODUCT PATH {treal:7.1f} us = {treal / t32:4.2f} of fp32 (split of x alone {tsp:6.1f} us = {x.numel() * 10 / tsp * 1e-3:5.0f} GB/s)'
    if n * cin * hw * hw <= 4 * 256 * 128 * 128:
        ref = torch.ops.aten.convolution_backward(dy.double(), x.double(), torch.zeros(shape, device='cuda', dtype=torch.float64), None, [1, 1], [1, 1], [1, 1], False, [0, 0], 1, [False, True, False])[1]
        a = conv2d_mfma.weight_gradient(x, dy, shape, (1, 1)).double(); b = real.double()
        sc = float(ref.abs().max())
        line += f' | max err / max|dw|: fp32 kernel {float((a - ref).abs().max()) / sc:.2e}, bf16x3 {float((b - ref).abs().max()) / sc:.2e}'
    print(line, flush=True)
