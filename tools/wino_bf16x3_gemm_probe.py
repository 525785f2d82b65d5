"""Dev tool (GPU box): an upper bound for DESIGN 7 item 1b -- the Winograd-domain GEMM of F(4x4,3x3) (36 positions x [Cout x Cin] x [Cin x tiles]) on the fp32 matrix
pipe against the same GEMM as six bf16 x bf16 products of three-term operand splits with fp32 accumulation, both through torch.bmm (hipBLASLt): what the vendor's
kernels make of the two arithmetic forms on this part, transforms and splitting not included.  Prints time, effective TFLOP/s and the error of both against float64.
    python tools/wino_bf16x3_gemm_probe.py"""
import torch


def timeit(fn, n=5):
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
    return [p.to(torch.bfloat16) for p in (hi, mid, lo)]


for (n, cin, cout, hw) in [(8, 128, 128, 256), (8, 64, 64, 512), (8, 256, 256, 128), (8, 512, 512, 64)]:
    tiles = n * (hw // 4) ** 2
    U = torch.randn(36, cout, cin, device='cuda') / cin ** 0.5
    V = torch.randn(36, cin, tiles, device='cuda')
    fl = 2.0 * 36 * cout * cin * tiles
    t32 = timeit(lambda: torch.bmm(U, V))
    Us, Vs = split3(U), split3(V)
    pairs = [(2, 0), (0, 2), (1, 1), (1, 0), (0, 1), (0, 0)]

    def six():
        acc = None
        for i, j in pairs:
            p = torch.bmm(Us[i], Vs[j]).float()
            acc = p if acc is None else acc + p
        return acc
    t16 = timeit(lambda: [torch.bmm(Us[i], Vs[j]) for i, j in pairs])
    t1 = timeit(lambda: torch.bmm(Us[0], Vs[0]))
    line = f'N{n} {cin}->{cout} {hw}^2 (36 x [{cout} x {cin}] x [{cin} x {tiles}]): fp32 bmm {t32:7.1f} us ({fl / t32 * 1e-6:5.1f} TF) | one bf16 bmm {t1:6.1f} us ({fl / t1 * 1e-6:6.1f} TF) | six bf16 bmm {t16:7.1f} us = {t16 / t32:4.2f} of fp32'
    if tiles <= 8 * 64 * 64:
        ref = torch.bmm(U.double(), V.double())
        a, b = torch.bmm(U, V).double(), six().double()
        sc = float(ref.abs().max())
        line += f' | err / max: fp32 {float((a - ref).abs().max()) / sc:.2e}, bf16x3 (bf16 outputs summed in fp32) {float((b - ref).abs().max()) / sc:.2e}'
    print(line, flush=True)
