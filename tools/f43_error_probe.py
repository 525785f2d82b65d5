"""Measured answer to "what does F(4x4,3x3) cost in accuracy on THIS network?" (VERDICT r2 item 2 ii), CPU only.

The float32 CPU oracle network (oracle/network_ref.py, config 2, N=1) is run twice on the same inputs: as is (direct float32
convolutions), and with every stride-1 3x3 convolution whose Cin >= --min-cin replaced by a float32 emulation of Winograd
F(4x4,3x3) (Lavin & Gray's matrices, points 0, +-1, +-2, inf; alternative point set +-1/2 with --points half): input / weight /
output transforms and the 36 channel GEMMs all in float32, exactly the arithmetic a gfx950 kernel would do (weights transformed
in float64 and rounded once, as a pack kernel can).  Prints the max-abs delta of the three network outputs against the direct
run -- the quantity north_star bounds by 1e-3 -- and, for scale, the same for an F(2x2,3x3) emulation.

    python tools/f43_error_probe.py [--min-cin 128] [--points std|half] [--algo f43|f23]
"""
import argparse
import os
import sys

import numpy as np
import torch
import torch.nn.functional as F

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, 'pasta-gan-plusplus_amd'), os.path.join(ROOT, 'tests', 'golden')]


def matrices(algo, points):
    if algo == 'f23':
        BT = np.array([[1, 0, -1, 0], [0, 1, 1, 0], [0, -1, 1, 0], [0, 1, 0, -1]], dtype=np.float64)
        G = np.array([[1, 0, 0], [.5, .5, .5], [.5, -.5, .5], [0, 0, 1]], dtype=np.float64)
        AT = np.array([[1, 1, 1, 0], [0, 1, -1, -1]], dtype=np.float64)
        return BT, G, AT, 2
    if points == 'std':
        BT = np.array([[4, 0, -5, 0, 1, 0], [0, -4, -4, 1, 1, 0], [0, 4, -4, -1, 1, 0], [0, -2, -1, 2, 1, 0], [0, 2, -1, -2, 1, 0], [0, 4, 0, -5, 0, 1]], dtype=np.float64)
        G = np.array([[1 / 4, 0, 0], [-1 / 6, -1 / 6, -1 / 6], [-1 / 6, 1 / 6, -1 / 6], [1 / 24, 1 / 12, 1 / 6], [1 / 24, -1 / 12, 1 / 6], [0, 0, 1]], dtype=np.float64)
        AT = np.array([[1, 1, 1, 1, 1, 0], [0, 1, -1, 2, -2, 0], [0, 1, 1, 4, 4, 0], [0, 1, -1, 8, -8, 1]], dtype=np.float64)
        return BT, G, AT, 4
    # points 0, +-1, +-1/2, inf built by the Toom-Cook construction (Vandermonde): lower dynamic range than +-2
    pts = [0.0, 1.0, -1.0, 0.5, -0.5]
    n, r = 4, 3
    a = n + r - 1
    AT = np.zeros((n, a)); G = np.zeros((a, r)); Bm = np.zeros((a, a))
    for j, p in enumerate(pts):
        for i in range(n):
            AT[i, j] = p ** i
        for i in range(r):
            G[j, i] = p ** i
    AT[n - 1, a - 1] = 1.0
    G[a - 1, r - 1] = 1.0
    # B^T from the Lagrange basis: solve so that  AT [(G g) * (BT d)] = correlation  for all g, d  (least squares on the identity)
    # unknown BT (a x a): for every unit g_k, d_l :  sum_j AT[i,j] G[j,k] BT[j,l] = [l == i + k]
    rows, rhs = [], []
    for i in range(n):
        for k in range(r):
            for l in range(a):
                row = np.zeros((a, a))
                row[:, l] = AT[i, :] * G[:, k]
                rows.append(row.reshape(-1)); rhs.append(1.0 if l == i + k else 0.0)
    BT = np.linalg.lstsq(np.array(rows), np.array(rhs), rcond=None)[0].reshape(a, a)
    return BT, G, AT, 4


def make_wino_conv(algo, points, min_cin, stats):
    BT, G, AT, m = matrices(algo, points)
    t = m + 2
    BTf, ATf = torch.tensor(BT, dtype=torch.float32), torch.tensor(AT, dtype=torch.float32)
    Gd = torch.tensor(G, dtype=torch.float64)
    real_conv2d = F.conv2d

    def conv2d(input, weight, bias=None, stride=1, padding=0, dilation=1, groups=1):
        st = stride if isinstance(stride, int) else stride[0]
        pd = (padding, padding) if isinstance(padding, int) else tuple(padding)
        if not (weight.shape[2:] == (3, 3) and st == 1 and groups == 1 and weight.shape[1] >= min_cin and input.dtype == torch.float32 and weight.shape[0] > 32):
            return real_conv2d(input, weight, bias, stride, padding, dilation, groups)
        stats['layers'] += 1
        n, c, h, w = input.shape
        oh, ow = h + 2 * pd[0] - 2, w + 2 * pd[1] - 2
        th, tw = -(-oh // m), -(-ow // m)
        xp = F.pad(input, (pd[1], tw * m + 2 - w - pd[1], pd[0], th * m + 2 - h - pd[0]))
        d = xp.unfold(2, t, m).unfold(3, t, m)                               # [n, c, th, tw, t, t]
        V = torch.einsum('ij,ncyxjk,lk->ncyxil', BTf, d, BTf)                   # float32 input transform
        U = torch.einsum('ij,ocjk,lk->ocil', Gd, weight.double(), Gd).float()   # weights: transformed in float64, rounded once
        M = torch.einsum('ocil,ncyxil->noyxil', U, V)                           # 36 (16) float32 GEMMs over the channels
        Y = torch.einsum('ij,noyxjk,lk->noyxil', ATf, M, ATf)                   # float32 output transform
        y = Y.permute(0, 1, 2, 4, 3, 5).reshape(n, weight.shape[0], th * m, tw * m)[:, :, :oh, :ow]
        return y + bias.reshape(1, -1, 1, 1) if bias is not None else y
    return conv2d


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument('--min-cin', type=int, default=128)
    ap.add_argument('--points', choices=['std', 'half'], default='std')
    ap.add_argument('--algo', choices=['f43', 'f23'], default='f43')
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
        real = F.conv2d
        F.conv2d = make_wino_conv(args.algo, args.points, args.min_cin, stats)
        try:
            got = bench.run_net(net, inp)
        finally:
            F.conv2d = real
    print(f'{args.algo} points={args.points} min_cin={args.min_cin}: {stats["layers"]} convolutions replaced')
    flips = float((got[2].argmax(1) != ref[2].argmax(1)).float().mean())
    for nm, a, b in zip(('img', 'finetune_img', 'pred_parsing'), got, ref):
        print(f'  {nm:13s} max-abs delta {float((a.double() - b.double()).abs().max()):.3e}   (output range {float(b.abs().max()):.1f})')
    print(f'  argmax labels flipped: {flips:.2e}')


if __name__ == '__main__':
    main()
