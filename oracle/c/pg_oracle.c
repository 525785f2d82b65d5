/*
 * pg_oracle.c -- scalar C restatement of the reference's two native kernels and of a direct
 * convolution.  TEST INFRASTRUCTURE ONLY (see oracle/__init__.py): linked by tests/ and by
 * __graft_entry__.smoke() through ctypes, never by the product.
 *
 * Follows, in plain loops and double accumulation:
 *   oracle_upfirdn2d : torch_utils/ops/upfirdn2d.cu:29-92 (upfirdn2d_kernel_large) -- receptive-field
 *                      maths on the zero-stuffed grid, flipped taps unless `flip`.
 *   oracle_bias_act  : torch_utils/ops/bias_act.cu:38-146, grad 0 only, with the Python fallback's
 *                      definitions of each activation (bias_act.py:23-33).
 *   oracle_conv2d    : the cross-correlation F.conv2d computes (conv2d_gradfix.py:38), NCHW / OIHW.
 * Inputs/outputs are float32; sums run in double so this is also a higher-precision yardstick.
 */
#include <math.h>
#include <stdint.h>

static int floor_div_i(int a, int b) { int q = a / b; return (a % b != 0 && ((a < 0) != (b < 0))) ? q - 1 : q; }

int oracle_upfirdn2d(const float* x, const float* f, float* y, int N, int C, int inH, int inW, int fh, int fw,
                     int outH, int outW, int upx, int upy, int downx, int downy, int padx0, int pady0, int flip, float gain)
{
    for (int nc = 0; nc < N * C; nc++)
    for (int oy = 0; oy < outH; oy++)
    for (int ox = 0; ox < outW; ox++) {
        double acc = 0.0;
        for (int ky = 0; ky < fh; ky++) {
            int uy = oy * downy + ky - pady0;                 /* coordinate on the zero-stuffed grid */
            if (uy < 0 || uy % upy != 0) continue;
            int iy = uy / upy;
            if (iy >= inH) continue;
            for (int kx = 0; kx < fw; kx++) {
                int ux = ox * downx + kx - padx0;
                if (ux < 0 || ux % upx != 0) continue;
                int ix = ux / upx;
                if (ix >= inW) continue;
                int fy = flip ? ky : fh - 1 - ky, fx = flip ? kx : fw - 1 - kx;
                acc += (double)x[((int64_t)nc * inH + iy) * inW + ix] * (double)f[fy * fw + fx];
            }
        }
        y[((int64_t)nc * outH + oy) * outW + ox] = (float)(acc * (double)gain);
    }
    (void)floor_div_i;
    return 0;
}

static double act_fwd(int act, double x, double alpha)
{
    switch (act) {
        case 1: return x;
        case 2: return x > 0 ? x : 0;
        case 3: return x > 0 ? x : x * alpha;
        case 4: return tanh(x);
        case 5: return 1.0 / (1.0 + exp(-x));
        case 6: return x >= 0 ? x : expm1(x);
        case 7: return x >= 0 ? 1.0507009873554804934193349852946 * x
                              : 1.0507009873554804934193349852946 * 1.6732632423543772848170429916717 * expm1(x);
        case 8: return x > 20 ? x : log1p(exp(x));
        case 9: return x / (1.0 + exp(-x));
    }
    return NAN;
}

/* y[i] = clamp(act(x[i] + b[(i / stepB) % sizeB]) * gain); b may be NULL; clamp < 0 disables. */
int oracle_bias_act(const float* x, const float* b, float* y, int64_t sizeX, int sizeB, int64_t stepB,
                    int act, float alpha, float gain, float clamp)
{
    if (act < 1 || act > 9) return -1;
    for (int64_t i = 0; i < sizeX; i++) {
        double v = x[i];
        if (b) v += b[(i / stepB) % sizeB];
        v = act_fwd(act, v, alpha) * gain;
        if (clamp >= 0) v = v > clamp ? clamp : (v < -clamp ? -clamp : v);
        y[i] = (float)v;
    }
    return 0;
}

/* y[n,co,oy,ox] = bias[co] + sum_{ci,ky,kx} w[co,ci,ky,kx] * x[n,ci,oy*stride+ky-pad, ox*stride+kx-pad] */
int oracle_conv2d(const float* x, const float* w, const float* bias, float* y, int N, int Cin, int H, int W,
                  int Cout, int KH, int KW, int stride, int pad_y, int pad_x)
{
    int OH = (H + 2 * pad_y - KH) / stride + 1, OW = (W + 2 * pad_x - KW) / stride + 1;
    for (int n = 0; n < N; n++)
    for (int co = 0; co < Cout; co++)
    for (int oy = 0; oy < OH; oy++)
    for (int ox = 0; ox < OW; ox++) {
        double acc = bias ? bias[co] : 0.0;
        for (int ci = 0; ci < Cin; ci++)
        for (int ky = 0; ky < KH; ky++) {
            int iy = oy * stride + ky - pad_y;
            if (iy < 0 || iy >= H) continue;
            for (int kx = 0; kx < KW; kx++) {
                int ix = ox * stride + kx - pad_x;
                if (ix < 0 || ix >= W) continue;
                acc += (double)w[(((int64_t)co * Cin + ci) * KH + ky) * KW + kx] * (double)x[(((int64_t)n * Cin + ci) * H + iy) * W + ix];
            }
        }
        y[(((int64_t)n * Cout + co) * OH + oy) * OW + ox] = (float)acc;
    }
    return 0;
}
