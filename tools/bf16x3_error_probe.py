"""VERDICT r3 item 8, numerics first: what would float32 convolution on the bf16 matrix pipe by THREE-TERM operand splitting cost in accuracy on
this network?  CPU only.

Every float32 value v is split by truncation into three bf16 terms v = hi + mid + lo (8 + 8 + 8 = 24 significand bits: exact), and a product a*b is
replaced by the six partial products  ah*bh + ah*bm + am*bh + ah*bl + al*bh + am*bm  (the three dropped ones are <= 2^-24 |ab| each), each a bf16 x bf16
product (exact in float32) accumulated in float32 -- the arithmetic of six v_mfma_f32_32x32x16_bf16 per operand pair.  The float32 CPU oracle
network (oracle/network_ref.py, config 2, N = 1) is run as is, then with the chosen convolution families emulated that way, then in float64; printed:
max-abs deltas of the three outputs (a) split vs float32, (b) float32 vs float64, (c) split vs float64 -- (b) is the noise floor the split must stay near.

    python tools/bf16x3_error_probe.py [--families direct|all] [--seed 0]
        direct = what the direct MFMA kernels run in config 2 (stride-2 3x3, transposed 3x3, 7x7, 1x1); all = every convolution
"""
import argparse
import os
import sys

import torch
import torch.nn.functional as F

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, 'pasta-gan-plusplus_amd'), os.path.join(ROOT, 'tests', 'golden')]


def split3(t):
    """float32 -> three float32 tensors holding bf16-representable values (truncation: mask the low 16 bits), t == hi + mid + lo exactly
    (24 significand bits = 3 x 8) unless the low terms underflow."""
    def trunc(v):
        return (v.view(torch.int32) & -65536).view(torch.float32)
    hi = trunc(t)
    r = t - hi
    mid = trunc(r)
    lo = trunc(r - mid)
    return hi, mid, lo


def make(real, families, stats, transposed=False):
    def conv(input, weight, bias=None, stride=1, padding=0, *rest, **kw):
        st = stride if isinstance(stride, int) else stride[0]
        k = tuple(weight.shape[2:])
        groups = kw.get('groups', rest[-1] if (not transposed and len(rest) >= 2) else (rest[1] if (transposed and len(rest) >= 2) else 1))
        direct = transposed or st == 2 or k in ((7, 7), (1, 1), (4, 4))
        if input.dtype != torch.float32 or groups != 1 or not (families == 'all' or direct):
            return real(input, weight, bias, stride, padding, *rest, **kw)
        stats['layers'] += 1
        xh, xm, xl = split3(input)
        wh, wm, wl = split3(weight)
        y = None
        for a, b in ((xl, wh), (xh, wl), (xm, wm), (xm, wh), (xh, wm), (xh, wh)):      # small terms first
            t = real(a, b, None, stride, padding, *rest, **kw)
            y = t if y is None else y + t
        return y + bias.reshape(1, -1, 1, 1) if bias is not None else y
    return conv


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument('--families', choices=['direct', 'all'], default='direct')
    ap.add_argument('--seed', type=int, default=0)
    args = ap.parse_args()
    import bench
    bench.torch = torch
    from oracle import network_ref as NR
    torch.set_num_threads(min(16, len(os.sched_getaffinity(0))))
    net = bench.init_weights(NR.SynthesisNetworkFull_v18(**bench.CFG2)).eval()
    inp = bench.make_inputs(1, 'cpu', seed=args.seed)
    with torch.no_grad():
        ref = bench.run_net(net, inp)
        stats = dict(layers=0)
        real_c, real_t = F.conv2d, F.conv_transpose2d
        F.conv2d, F.conv_transpose2d = make(real_c, args.families, stats), make(real_t, args.families, stats, transposed=True)
        try:
            got = bench.run_net(net, inp)
        finally:
            F.conv2d, F.conv_transpose2d = real_c, real_t
        net64 = net.double()
        ref64 = bench.run_net(net64, {k: (v.double() if torch.is_tensor(v) and v.dtype == torch.float32 else v) for k, v in inp.items()})
    print(f'families={args.families}: {stats["layers"]} convolutions emulated as 6 bf16 products')
    for nm, a, b, c in zip(('img', 'finetune_img', 'pred_parsing'), got, ref, ref64):
        d = lambda u, v: float((u.double() - v.double()).abs().max())
        print(f'  {nm:13s} split vs f32 {d(a, b):.3e}   f32 vs f64 {d(b, c):.3e}   split vs f64 {d(a, c):.3e}   (output range {float(b.abs().max()):.1f})')


if __name__ == '__main__':
    main()
