// C ABI of the conv2d plugin + its small support kernels (weight packing, demodulation
// coefficients, instance-norm statistics, SPADE combine).  The MFMA implicit-GEMM kernel itself
// lives in conv2d_kernel.h and is instantiated per geometry in conv2d_inst_*.hip.
#include "conv2d_kernel.h"

namespace {

using namespace pg;
using pgconv::ConvParams;
typedef float f32x4 __attribute__((ext_vector_type(4)));

// ------------------------------------------------------------------ weight packing
__global__ __launch_bounds__(256) void pack_weight_kernel(const float* __restrict__ w, float* __restrict__ wp, int Cout, int Cin, int KH, int KW,
                                                          int CinP, int CoutP, float scale, int flip, int transpose_oi) {
    const int T = KH * KW;
    const int64_t total = (int64_t)CinP * T * CoutP;
    for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < total; i += (int64_t)gridDim.x * 256) {
        const int co = (int)(i % CoutP);
        const int tap = (int)((i / CoutP) % T);
        const int ci = (int)(i / ((int64_t)CoutP * T));
        float v = 0.f;
        int ky = tap / KW, kx = tap % KW;
        int cs = ci;
        // ROWPAIR form (conv2d_kernel.h; every 7x7 kernel with Cin = 3): "channel 3" of tap (ky, kx), ky even, is channel 2 of tap (ky + 1, kx)
        if (Cin == 3 && KH == 7 && KW == 7 && ci == 3 && (ky & 1) == 0 && ky + 1 < KH) { cs = 2; ky += 1; }
        if (co < Cout && cs < Cin) {
            const int ci = cs;
            if (flip) { ky = KH - 1 - ky; kx = KW - 1 - kx; }
            const int64_t src = transpose_oi ? (((int64_t)ci * Cout + co) * KH + ky) * KW + kx
                                             : (((int64_t)co * Cin + ci) * KH + ky) * KW + kx;
            v = w[src] * scale;
        }
        wp[i] = v;
    }
}

// Winograd F(2x2, 3x3) weight transform: U = G g G^T per (ci, co), stored in the operand-stream order of conv2d_wino.h.
//   G = [[1, 0, 0], [1/2, 1/2, 1/2], [1/2, -1/2, 1/2], [0, 0, 1]]
__global__ __launch_bounds__(256) void wino_pack_kernel(const float* __restrict__ w, float* __restrict__ up, int Cout, int Cin,
                                                        int CinP, int CoutP, float scale, int flip, int transpose_oi) {
    const int64_t total = (int64_t)CinP * CoutP;
    for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < total; i += (int64_t)gridDim.x * 256) {
        const int co = (int)(i % CoutP), ci = (int)(i / CoutP);
        float g[3][3];
#pragma unroll
        for (int ky = 0; ky < 3; ky++)
#pragma unroll
            for (int kx = 0; kx < 3; kx++) {
                float v = 0.f;
                if (co < Cout && ci < Cin) {
                    const int sy = flip ? 2 - ky : ky, sx = flip ? 2 - kx : kx;
                    const int64_t src = transpose_oi ? (((int64_t)ci * Cout + co) * 3 + sy) * 3 + sx
                                                     : (((int64_t)co * Cin + ci) * 3 + sy) * 3 + sx;
                    v = w[src] * scale;
                }
                g[ky][kx] = v;
            }
        float t[4][3];                                   // G g
#pragma unroll
        for (int kx = 0; kx < 3; kx++) {
            t[0][kx] = g[0][kx];
            t[1][kx] = 0.5f * (g[0][kx] + g[1][kx] + g[2][kx]);
            t[2][kx] = 0.5f * (g[0][kx] - g[1][kx] + g[2][kx]);
            t[3][kx] = g[2][kx];
        }
#pragma unroll
        for (int a = 0; a < 4; a++) {                    // (G g) G^T
            const float u[4] = {t[a][0], 0.5f * (t[a][0] + t[a][1] + t[a][2]), 0.5f * (t[a][0] - t[a][1] + t[a][2]), t[a][2]};
#pragma unroll
            for (int b = 0; b < 4; b++) {
                // [a][co / 32][ci / 2][ci & 1][co & 31][b]: the four b values of one MFMA lane are one 16-byte word
                const int64_t dst = ((((int64_t)a * (CoutP / 32) + (co >> 5)) * (CinP / 2) + (ci >> 1)) * 2 + (ci & 1)) * 128 + (co & 31) * 4 + b;
                up[dst] = u[b];
            }
        }
    }
}


// Winograd F(4x4, 3x3) weight transform: U = G g G^T (6x6) per (ci, co) in float64, rounded once, stored in the order the waves of
// conv2d_wino4.h walk it: [m-block 64][mt][a][chunk 16 ch][group = (b half jg, pair quad)][b third jj][lane = (h, co & 31)][pair s], with
//   xi = 6a + b,  b = 3jg + jj,   channel = 16 chunk + 2 (4 quad + s) + h.
//   G = [[1/4, 0, 0], [-1/6, -1/6, -1/6], [-1/6, 1/6, -1/6], [1/24, 1/12, 1/6], [1/24, -1/12, 1/6], [0, 0, 1]]
__global__ __launch_bounds__(256) void wino4_pack_kernel(const float* __restrict__ w, float* __restrict__ up, int Cout, int Cin,
                                                         int CinP, int CoutP, float scale, int flip, int transpose_oi) {
    const double G[6][3] = {{0.25, 0.0, 0.0}, {-1.0 / 6, -1.0 / 6, -1.0 / 6}, {-1.0 / 6, 1.0 / 6, -1.0 / 6},
                            {1.0 / 24, 1.0 / 12, 1.0 / 6}, {1.0 / 24, -1.0 / 12, 1.0 / 6}, {0.0, 0.0, 1.0}};
    const int nchunks = CinP / 16;
    const int64_t total = (int64_t)CinP * CoutP;
    for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < total; i += (int64_t)gridDim.x * 256) {
        const int co = (int)(i % CoutP), ci = (int)(i / CoutP);
        double g[3][3];
#pragma unroll
        for (int ky = 0; ky < 3; ky++)
#pragma unroll
            for (int kx = 0; kx < 3; kx++) {
                double v = 0.0;
                if (co < Cout && ci < Cin) {
                    const int sy = flip ? 2 - ky : ky, sx = flip ? 2 - kx : kx;
                    const int64_t src = transpose_oi ? (((int64_t)ci * Cout + co) * 3 + sy) * 3 + sx
                                                     : (((int64_t)co * Cin + ci) * 3 + sy) * 3 + sx;
                    v = (double)(w[src] * scale);
                }
                g[ky][kx] = v;
            }
        double tg[6][3];                                 // G g
#pragma unroll
        for (int a = 0; a < 6; a++)
#pragma unroll
            for (int kx = 0; kx < 3; kx++) tg[a][kx] = G[a][0] * g[0][kx] + G[a][1] * g[1][kx] + G[a][2] * g[2][kx];
        const int mb = co >> 6, mt = (co >> 5) & 1, m = co & 31;
        const int k = ci >> 4, cc = ci & 15, pair = cc >> 1, h = cc & 1, quad = pair >> 2, sp = pair & 3;
#pragma unroll
        for (int a = 0; a < 6; a++)
#pragma unroll
            for (int b = 0; b < 6; b++) {
                const double u = tg[a][0] * G[b][0] + tg[a][1] * G[b][1] + tg[a][2] * G[b][2];      // (G g) G^T
                const int jg = b / 3, jj = b % 3;
                const int64_t unit = ((int64_t)(mb * 2 + mt) * 6 + a) * nchunks + k;
                const int64_t dst = ((((unit * 2 + jg) * 2 + quad) * 3 + jj) * 64 + (h * 32 + m)) * 4 + sp;
                up[dst] = (float)u;
            }
    }
}

// The same transform for the X3 form of conv2d_wino4.h: U rounded to float32 once (the value the fp32 form multiplies by), then split exactly into three bf16
// planes by truncation (u = p0 + p1 + p2, 8 + 8 + 8 significand bits), stored [m-block 64][mt][a][chunk 16 ch][b 6][plane 3][lane = (h, co & 31)][j 8] as 16-bit
// words with channel = 16 chunk + 2 j + h: one 16-byte word per lane = the A operand of one v_mfma_f32_32x32x16_bf16.
__global__ __launch_bounds__(256) void wino4x3_pack_kernel(const float* __restrict__ w, unsigned short* __restrict__ up, int Cout, int Cin,
                                                           int CinP, int CoutP, float scale, int flip, int transpose_oi) {
    const double G[6][3] = {{0.25, 0.0, 0.0}, {-1.0 / 6, -1.0 / 6, -1.0 / 6}, {-1.0 / 6, 1.0 / 6, -1.0 / 6},
                            {1.0 / 24, 1.0 / 12, 1.0 / 6}, {1.0 / 24, -1.0 / 12, 1.0 / 6}, {0.0, 0.0, 1.0}};
    const int nchunks = CinP / 16;
    const int64_t total = (int64_t)CinP * CoutP;
    for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < total; i += (int64_t)gridDim.x * 256) {
        const int co = (int)(i % CoutP), ci = (int)(i / CoutP);
        double g[3][3];
#pragma unroll
        for (int ky = 0; ky < 3; ky++)
#pragma unroll
            for (int kx = 0; kx < 3; kx++) {
                double v = 0.0;
                if (co < Cout && ci < Cin) {
                    const int sy = flip ? 2 - ky : ky, sx = flip ? 2 - kx : kx;
                    const int64_t src = transpose_oi ? (((int64_t)ci * Cout + co) * 3 + sy) * 3 + sx
                                                     : (((int64_t)co * Cin + ci) * 3 + sy) * 3 + sx;
                    v = (double)(w[src] * scale);
                }
                g[ky][kx] = v;
            }
        double tg[6][3];                                 // G g
#pragma unroll
        for (int a = 0; a < 6; a++)
#pragma unroll
            for (int kx = 0; kx < 3; kx++) tg[a][kx] = G[a][0] * g[0][kx] + G[a][1] * g[1][kx] + G[a][2] * g[2][kx];
        const int mb = co >> 6, mt = (co >> 5) & 1, m = co & 31;
        const int k = ci >> 4, cc = ci & 15, j = cc >> 1, h = cc & 1;
#pragma unroll
        for (int a = 0; a < 6; a++)
#pragma unroll
            for (int b = 0; b < 6; b++) {
                const float u = (float)(tg[a][0] * G[b][0] + tg[a][1] * G[b][1] + tg[a][2] * G[b][2]);      // (G g) G^T, rounded once
                const unsigned u0 = __float_as_uint(u) & 0xffff0000u;
                const float r = u - __uint_as_float(u0);
                const unsigned u1 = __float_as_uint(r) & 0xffff0000u;
                const unsigned u2 = __float_as_uint(r - __uint_as_float(u1)) & 0xffff0000u;
                const int64_t unit = ((int64_t)(mb * 2 + mt) * 6 + a) * nchunks + k;
                const int64_t dst = (((unit * 6 + b) * 3) * 64 + (h * 32 + m)) * 8 + j;                     // plane 0; planes are 64 * 8 words apart
                up[dst] = (unsigned short)(u0 >> 16);
                up[dst + 512] = (unsigned short)(u1 >> 16);
                up[dst + 1024] = (unsigned short)(u2 >> 16);
            }
    }
}

// The same transform in the order the waves of conv2d_wino4b.h walk it (two workgroups per CU, v_mfma_f32_16x16x4_f32):
//   [m-block 64][cout block cb 4][row half ah 2][chunk 16 ch][e = 6 a' + b (18)][lane = (kq, m) 64][j 4]
//   with xi = 6 (3 ah + a') + b, cout = 64 mb + 16 cb + m, channel = 16 chunk + 4 j + kq: one 16-byte word per lane = the A operands of the four K steps of one xi.
__global__ __launch_bounds__(256) void wino4b_pack_kernel(const float* __restrict__ w, float* __restrict__ up, int Cout, int Cin,
                                                          int CinP, int CoutP, float scale, int flip, int transpose_oi) {
    const double G[6][3] = {{0.25, 0.0, 0.0}, {-1.0 / 6, -1.0 / 6, -1.0 / 6}, {-1.0 / 6, 1.0 / 6, -1.0 / 6},
                            {1.0 / 24, 1.0 / 12, 1.0 / 6}, {1.0 / 24, -1.0 / 12, 1.0 / 6}, {0.0, 0.0, 1.0}};
    const int nchunks = CinP / 16;
    const int64_t total = (int64_t)CinP * CoutP;
    for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < total; i += (int64_t)gridDim.x * 256) {
        const int co = (int)(i % CoutP), ci = (int)(i / CoutP);
        double g[3][3];
#pragma unroll
        for (int ky = 0; ky < 3; ky++)
#pragma unroll
            for (int kx = 0; kx < 3; kx++) {
                double v = 0.0;
                if (co < Cout && ci < Cin) {
                    const int sy = flip ? 2 - ky : ky, sx = flip ? 2 - kx : kx;
                    const int64_t src = transpose_oi ? (((int64_t)ci * Cout + co) * 3 + sy) * 3 + sx
                                                     : (((int64_t)co * Cin + ci) * 3 + sy) * 3 + sx;
                    v = (double)(w[src] * scale);
                }
                g[ky][kx] = v;
            }
        double tg[6][3];                                 // G g
#pragma unroll
        for (int a = 0; a < 6; a++)
#pragma unroll
            for (int kx = 0; kx < 3; kx++) tg[a][kx] = G[a][0] * g[0][kx] + G[a][1] * g[1][kx] + G[a][2] * g[2][kx];
        const int mb = co >> 6, cbk = (co >> 4) & 3, m = co & 15;
        const int k = ci >> 4, cc = ci & 15, j = cc >> 2, kq = cc & 3;
#pragma unroll
        for (int a = 0; a < 6; a++)
#pragma unroll
            for (int b = 0; b < 6; b++) {
                const double u = tg[a][0] * G[b][0] + tg[a][1] * G[b][1] + tg[a][2] * G[b][2];      // (G g) G^T
                const int64_t unit = ((int64_t)(mb * 4 + cbk) * 2 + a / 3) * nchunks + k;
                const int64_t dst = ((unit * 18 + (a % 3) * 6 + b) * 64 + (kq * 16 + m)) * 4 + j;
                up[dst] = (float)u;
            }
    }
}

// ------------------------------------------------------------------ demodulation coefficients
// one workgroup per (n, o): rsqrt(sum_{i,k} (w[o,i,k] * scale * s[n,i])^2 + 1e-8)
__global__ __launch_bounds__(256) void dcoefs_kernel(const float* __restrict__ w, const float* __restrict__ styles, float* __restrict__ d,
                                                     int Cout, int Cin, int KHW, float scale) {
    const int o = blockIdx.x % Cout, n = blockIdx.x / Cout;
    const float* wo = w + (int64_t)o * Cin * KHW;
    const float* sn = styles + (int64_t)n * Cin;
    float acc = 0.f;
    for (int e = threadIdx.x; e < Cin * KHW; e += 256) {
        const float v = wo[e] * scale * sn[e / KHW];
        acc += v * v;
    }
    __shared__ float red[4];
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) acc += __shfl_down(acc, off, 64);
    if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = acc;
    __syncthreads();
    if (threadIdx.x == 0) d[blockIdx.x] = rsqrtf(red[0] + red[1] + red[2] + red[3] + 1e-8f);
}

// The same coefficients from the per-(o, i) tap energy  W2[o,i] = scale^2 * sum_k w[o,i,k]^2  (weights only: computed once per
// weight version and cached by the caller):  dcoefs[n,o] = rsqrt(sum_i W2[o,i] * s[n,i]^2 + 1e-8) -- 1 / KHW of the weight bytes
// per call.  One thread per (o, i).
__global__ __launch_bounds__(256) void modconv_w2_kernel(const float* __restrict__ w, float* __restrict__ w2, int64_t total, int KHW, float scale2) {
    const int64_t e = (int64_t)blockIdx.x * 256 + threadIdx.x;
    if (e >= total) return;
    const float* we = w + e * KHW;
    float acc = 0.f;
    for (int k = 0; k < KHW; k++) acc = fmaf(we[k], we[k], acc);
    w2[e] = acc * scale2;
}

// Style preparation of one modulated convolution in a single launch (workgroup = 4 waves = 16 couts of sample blockIdx.y):
//   normalize != 0 (the half-precision pre-normalisation of networks.py:57-59):  smax = max(max_i |s[n,i]|, 1e-20), s' = s / smax,
//       s_norm[n,i] = s' (float32) and, when s16 is given, the same rounded to bf16 / fp16;   otherwise s' = s
//   out[n,o] = demodulate ? rsqrt(sum_i W2[o,i] * s'^2 + 1e-8) : smax
__device__ __forceinline__ void modconv_prep_body(const float* __restrict__ w2, const float* __restrict__ styles, float* __restrict__ out,
                                                  float* __restrict__ s_norm, unsigned short* __restrict__ s16, int half_dtype,
                                                  int Cout, int Cin, int normalize, int demodulate) {
    extern __shared__ float sq[];                    // s'^2 of this sample
    __shared__ float red[4];
    const int n = blockIdx.y, t = threadIdx.x, lane = t & 63, wave = t >> 6;
    const float* sn = styles + (int64_t)n * Cin;
    float smax = 1.f;
    if (normalize) {
        float m = 0.f;
        for (int i = t; i < Cin; i += 256) m = fmaxf(m, fabsf(sn[i]));
#pragma unroll
        for (int off = 32; off > 0; off >>= 1) m = fmaxf(m, __shfl_down(m, off, 64));
        if (lane == 0) red[wave] = m;
        __syncthreads();
        smax = fmaxf(fmaxf(fmaxf(red[0], red[1]), fmaxf(red[2], red[3])), 1e-20f);
    }
    for (int i = t; i < Cin; i += 256) {
        const float v = normalize ? sn[i] / smax : sn[i];
        sq[i] = v * v;
        if (normalize && blockIdx.x == 0) {
            if (s_norm) s_norm[(int64_t)n * Cin + i] = v;
            if (s16) {
                unsigned short h;
                if (half_dtype == PG_BF16) { const __bf16 b = (__bf16)v; h = __builtin_bit_cast(unsigned short, b); }
                else { const _Float16 b = (_Float16)v; h = __builtin_bit_cast(unsigned short, b); }
                s16[(int64_t)n * Cin + i] = h;
            }
        }
    }
    __syncthreads();
    const int o0 = blockIdx.x * 16 + wave * 4;
    if (o0 >= Cout) return;                                          // wave-uniform
    float acc[4] = {0.f, 0.f, 0.f, 0.f};
    if (demodulate) {
        const float* wo = w2 + (int64_t)o0 * Cin;
        const int64_t r1 = o0 + 1 < Cout ? Cin : 0, r2 = o0 + 2 < Cout ? 2 * (int64_t)Cin : 0, r3 = o0 + 3 < Cout ? 3 * (int64_t)Cin : 0;   // rows past Cout re-read row o0
        for (int i = lane; i < Cin; i += 64) {                       // four independent load streams per lane
            const float q = sq[i];
            acc[0] = fmaf(wo[i], q, acc[0]); acc[1] = fmaf(wo[r1 + i], q, acc[1]);
            acc[2] = fmaf(wo[r2 + i], q, acc[2]); acc[3] = fmaf(wo[r3 + i], q, acc[3]);
        }
#pragma unroll
        for (int j = 0; j < 4; j++)
#pragma unroll
            for (int off = 32; off > 0; off >>= 1) acc[j] += __shfl_down(acc[j], off, 64);
    }
    if (lane == 0) {
#pragma unroll
        for (int j = 0; j < 4; j++)
            if (o0 + j < Cout) out[(int64_t)n * Cout + o0 + j] = demodulate ? rsqrtf(acc[j] + 1e-8f) : smax;
    }
}

__global__ __launch_bounds__(256) void modconv_prep_kernel(const float* __restrict__ w2, const float* __restrict__ styles, float* __restrict__ out,
                                                           float* __restrict__ s_norm, unsigned short* __restrict__ s16, int half_dtype,
                                                           int Cout, int Cin, int normalize, int demodulate) {
    modconv_prep_body(w2, styles, out, s_norm, s16, half_dtype, Cout, Cin, normalize, demodulate);
}

// Every modulated convolution of a network in ONE launch (grid z = job): the job table travels by value in the kernel arguments, so a captured
// graph holds it and no host -> device copy is needed per call.
__global__ __launch_bounds__(256) void modconv_prep_batched_kernel(pg_modconv_prep_jobs J) {
    const int j = blockIdx.z;
    const int Cout = J.cout[j];
    if ((int)blockIdx.x * 16 >= Cout) return;            // grid x is sized for the widest job
    modconv_prep_body(J.w2[j], J.styles[j], J.out[j], J.s_norm[j], (unsigned short*)J.s16[j], J.half_dtype, Cout, J.cin[j], J.flags[j] & 1, (J.flags[j] >> 1) & 1);
}

// ------------------------------------------------------------------ instance-norm statistics (two passes over one plane)
constexpr int IN_THREADS = 1024;
__device__ __forceinline__ float block_sum_1024(float v, float* red) {
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) v += __shfl_down(v, off, 64);
    __syncthreads();
    if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = v;
    __syncthreads();
    float s = 0.f;
    if (threadIdx.x < 64) {
        s = threadIdx.x < IN_THREADS / 64 ? red[threadIdx.x] : 0.f;
#pragma unroll
        for (int off = 8; off > 0; off >>= 1) s += __shfl_down(s, off, 64);
    }
    __syncthreads();
    if (threadIdx.x == 0) red[0] = s;
    __syncthreads();
    return red[0];
}

// One pass: sums of (x - K) and (x - K)^2 with K = the plane's first element (shifted-data variance: immune to the
// cancellation of E[x^2] - E[x]^2 when |mean| >> std), 1024 threads per (n, c) plane, 16-byte loads.
__global__ __launch_bounds__(IN_THREADS) void instance_norm_stats_kernel(const float* __restrict__ x, float* __restrict__ mean, float* __restrict__ rstd, int64_t HW, float eps) {
    __shared__ float red[IN_THREADS / 64];
    const float* xp = x + (int64_t)blockIdx.x * HW;
    const float K = xp[0];
    float s = 0.f, q = 0.f;
    if ((HW % 4 == 0) && aligned16(xp)) {
        const int64_t n4 = HW / 4;
        int64_t i = threadIdx.x;
        for (; i + 3 * IN_THREADS < n4; i += 4 * IN_THREADS) {        // four 16-byte loads in flight per thread
            f32x4 v[4];
#pragma unroll
            for (int u = 0; u < 4; u++) v[u] = ((const f32x4*)xp)[i + u * IN_THREADS];
#pragma unroll
            for (int u = 0; u < 4; u++) {
                const float a = v[u][0] - K, b = v[u][1] - K, c = v[u][2] - K, d = v[u][3] - K;
                s += (a + b) + (c + d);
                q += (a * a + b * b) + (c * c + d * d);
            }
        }
        for (; i < n4; i += IN_THREADS) {
            const f32x4 v = ((const f32x4*)xp)[i];
            const float a = v[0] - K, b = v[1] - K, c = v[2] - K, d = v[3] - K;
            s += (a + b) + (c + d);
            q += (a * a + b * b) + (c * c + d * d);
        }
    } else {
        for (int64_t i = threadIdx.x; i < HW; i += IN_THREADS) { const float a = xp[i] - K; s += a; q += a * a; }
    }
    const float S = block_sum_1024(s, red), Q = block_sum_1024(q, red);
    if (threadIdx.x == 0) {
        const float ms = S / (float)HW;                          // mean of the shifted data
        const float var = fmaxf(Q / (float)HW - ms * ms, 0.f);
        mean[blockIdx.x] = K + ms;
        rstd[blockIdx.x] = 1.0f / sqrtf(var + eps);
    }
}

// out = (x - mean) * rstd * (1 + gamma) + beta
__global__ __launch_bounds__(256) void spade_norm_kernel(const float* __restrict__ x, const float* __restrict__ mean, const float* __restrict__ rstd,
                                                         const float* __restrict__ gamma, const float* __restrict__ beta, float* __restrict__ y,
                                                         int64_t HW4, int64_t total4) {
    for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < total4; i += (int64_t)gridDim.x * 256) {
        const int64_t plane = i / HW4;
        const float m = mean[plane], r = rstd[plane];
        const f32x4 xv = ((const f32x4*)x)[i], g = ((const f32x4*)gamma)[i], b = ((const f32x4*)beta)[i];
        f32x4 o;
#pragma unroll
        for (int k = 0; k < 4; k++) o[k] = (xv[k] - m) * r * (1.f + g[k]) + b[k];
        ((f32x4*)y)[i] = o;
    }
}
__global__ __launch_bounds__(256) void spade_norm_scalar_kernel(const float* __restrict__ x, const float* __restrict__ mean, const float* __restrict__ rstd,
                                                                const float* __restrict__ gamma, const float* __restrict__ beta, float* __restrict__ y,
                                                                int64_t HW, int64_t total) {
    for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < total; i += (int64_t)gridDim.x * 256) {
        const int64_t plane = i / HW;
        y[i] = (x[i] - mean[plane]) * rstd[plane] * (1.f + gamma[i]) + beta[i];
    }
}


// ------------------------------------------------------------------ SPADE combine on the TRAINING route (networks.py:1715-1723 and its autograd)
// gamma / beta are planes of one [N, 2C, H, W] tensor (the two 3x3 convolutions run as one launch over the stacked weights): plane (n, c)
// of gamma starts at gamma + n*gs + c*HW.  Forward: y = xh (1 + gamma) + beta, xh = (x - mean) rstd.  Backward of that and of the instance
// norm inside it, for upstream dy:   dbeta = dy,  dgamma = dy xh,  g = dy (1 + gamma),
//     dx = rstd (g - mean_hw(g) - xh mean_hw(g xh))
// kernel A: the two plane sums (one 1024-thread workgroup per plane, like the statistics pass); kernel B: everything elementwise.
template <int VEC>
__global__ __launch_bounds__(256) void spade_train_forward_kernel(const float* __restrict__ x, const float* __restrict__ mean, const float* __restrict__ rstd,
                                                                  const float* __restrict__ gamma, const float* __restrict__ beta, float* __restrict__ y,
                                                                  int C, int64_t HWV, int64_t totalV, int64_t gs, int64_t bs) {
    typedef float V __attribute__((ext_vector_type(VEC)));
    for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < totalV; i += (int64_t)gridDim.x * 256) {
        const int64_t plane = i / HWV, in = i - plane * HWV;
        const int64_t n = plane / C, c = plane - n * C;
        const float m = mean[plane], r = rstd[plane];
        const V xv = ((const V*)x)[i];
        const V g = ((const V*)(gamma + n * gs + c * HWV * VEC))[in], b = ((const V*)(beta + n * bs + c * HWV * VEC))[in];
        ((V*)y)[i] = (xv - m) * r * (1.f + g) + b;
    }
}

__global__ __launch_bounds__(IN_THREADS) void spade_train_sums_kernel(const float* __restrict__ dy, const float* __restrict__ x, const float* __restrict__ mean,
                                                                     const float* __restrict__ rstd, const float* __restrict__ gamma, float* __restrict__ sums,
                                                                     int C, int64_t HW, int64_t gs, int vec) {
    __shared__ float red[IN_THREADS / 64];
    const int64_t plane = blockIdx.x, n = plane / C, c = plane - n * C;
    const float* dp = dy + plane * HW;
    const float* xp = x + plane * HW;
    const float* gp = gamma + n * gs + c * HW;
    const float m = mean[plane], r = rstd[plane];
    float s1 = 0.f, s2 = 0.f;
    if (vec) {
        const int64_t n4 = HW / 4;
        for (int64_t i = threadIdx.x; i < n4; i += IN_THREADS) {
            const f32x4 d = ((const f32x4*)dp)[i], xv = ((const f32x4*)xp)[i], g = ((const f32x4*)gp)[i];
#pragma unroll
            for (int k = 0; k < 4; k++) { const float gg = d[k] * (1.f + g[k]); s1 += gg; s2 += gg * ((xv[k] - m) * r); }
        }
    } else {
        for (int64_t i = threadIdx.x; i < HW; i += IN_THREADS) { const float gg = dp[i] * (1.f + gp[i]); s1 += gg; s2 += gg * ((xp[i] - m) * r); }
    }
    const float S1 = block_sum_1024(s1, red), S2 = block_sum_1024(s2, red);
    if (threadIdx.x == 0) { sums[2 * plane] = S1 / (float)HW; sums[2 * plane + 1] = S2 / (float)HW; }
}

template <int VEC>
__global__ __launch_bounds__(256) void spade_train_backward_kernel(const float* __restrict__ dy, const float* __restrict__ x, const float* __restrict__ mean,
                                                                   const float* __restrict__ rstd, const float* __restrict__ gamma, const float* __restrict__ sums,
                                                                   float* __restrict__ dx, float* __restrict__ dgamma, float* __restrict__ dbeta,
                                                                   int C, int64_t HWV, int64_t totalV, int64_t gs, int64_t dgs, int64_t dbs) {
    typedef float V __attribute__((ext_vector_type(VEC)));
    for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < totalV; i += (int64_t)gridDim.x * 256) {
        const int64_t plane = i / HWV, in = i - plane * HWV;
        const int64_t n = plane / C, c = plane - n * C;
        const float m = mean[plane], r = rstd[plane], a1 = sums[2 * plane], a2 = sums[2 * plane + 1];
        const V d = ((const V*)dy)[i], xv = ((const V*)x)[i];
        const V g = ((const V*)(gamma + n * gs + c * HWV * VEC))[in];
        const V xh = (xv - m) * r;
        if (dx) ((V*)dx)[i] = r * (d * (1.f + g) - a1 - xh * a2);
        if (dgamma) {
            ((V*)(dgamma + n * dgs + c * HWV * VEC))[in] = d * xh;
            ((V*)(dbeta + n * dbs + c * HWV * VEC))[in] = d;
        }
    }
}


// ------------------------------------------------------------------ SPADE feature assembly (networks.py:2253-2276, 2311-2316)
// Garment features of the upper / lower branch are inpainted where the predicted parsing mask exceeds the warped-garment
// mask, then merged:  with m = (mask[2y,2x] > 0.9), v = m && (denorm_mask[2y,2x] > 0.9), r = m - v
//   B   = feat * (1 - r) + (sum_p feat*v / count) * r          count = sum_p v, replaced by 256*256 when <= 10
//   out = B_upper * m_upper + B_lower * m_lower
// Kernel 1: per (n, c) plane the masked sum (and per n the count); kernel 2: the blend.  Masks are [N,1,2H,2W], read at the
// even pixels ("nearest" down-sampling by 2); every product with a 0/1 mask is exact, so the only rounding differences to
// the unfused composition are in the order of the plane sum.
// even elements of 8 consecutive mask floats (row 2y, columns 2x .. 2x+7) thresholded at 0.9
__device__ __forceinline__ f32x4 even_mask4(const float* __restrict__ row8) {
    const f32x4 a = ((const f32x4*)row8)[0], b = ((const f32x4*)row8)[1];
    f32x4 m;
    m[0] = a[0] > 0.9f ? 1.f : 0.f; m[1] = a[2] > 0.9f ? 1.f : 0.f; m[2] = b[0] > 0.9f ? 1.f : 0.f; m[3] = b[2] > 0.9f ? 1.f : 0.f;
    return m;
}

template <int G>       // G channel planes of one image per workgroup (round 5): the thresholded masks -- 4x the bytes of a feature plane through even_mask4 -- are read once for G planes;
__global__ __launch_bounds__(IN_THREADS) void spade_masked_sums_kernel(const float* __restrict__ feat, const float* __restrict__ mask, const float* __restrict__ dmask,
                                                                       float* __restrict__ sums, float* __restrict__ counts, int C, int H, int W, int vec) {
    // every plane's sum keeps the order it had with one plane per workgroup (same pixel -> thread map, same block reduction)
    __shared__ float red[IN_THREADS / 64];
    const int plane0 = blockIdx.x * G, n = plane0 / C;
    const float* fp = feat + (int64_t)plane0 * H * W;
    const int64_t HWl = (int64_t)H * W;
    const float* mp = mask + (int64_t)n * 4 * H * W;
    const float* dp = dmask + (int64_t)n * 4 * H * W;
    float s[G], cnt = 0.f;
#pragma unroll
    for (int g = 0; g < G; g++) s[g] = 0.f;
    if (vec) {                                   // W % 4 == 0, 16-byte aligned planes: 4 pixels per step, two steps in flight
        const int W4 = W / 4, n4 = H * W4;
        for (int i0 = threadIdx.x; i0 < n4; i0 += 2 * IN_THREADS) {
            f32x4 f[2][G], m[2], d[2];
            bool ok[2];
#pragma unroll
            for (int u = 0; u < 2; u++) {
                const int i = i0 + u * IN_THREADS;
                ok[u] = i < n4;
                const int ic = ok[u] ? i : 0;
                const int y = ic / W4, x4 = ic - y * W4;
                const int64_t mi = (int64_t)(2 * y) * (2 * W) + 8 * x4;
#pragma unroll
                for (int g = 0; g < G; g++) f[u][g] = ((const f32x4*)(fp + g * HWl))[ic];
                m[u] = even_mask4(mp + mi);
                d[u] = even_mask4(dp + mi);
            }
#pragma unroll
            for (int u = 0; u < 2; u++)
                if (ok[u]) {
#pragma unroll
                    for (int k = 0; k < 4; k++) {
                        const float v = m[u][k] * d[u][k];
#pragma unroll
                        for (int g = 0; g < G; g++) s[g] += f[u][g][k] * v;
                        cnt += v;
                    }
                }
        }
    } else {
        for (int i = threadIdx.x; i < H * W; i += IN_THREADS) {
            const int y = i / W, x = i - y * W;
            const int64_t mi = (int64_t)(2 * y) * (2 * W) + 2 * x;
            const float v = (mp[mi] > 0.9f && dp[mi] > 0.9f) ? 1.f : 0.f;
#pragma unroll
            for (int g = 0; g < G; g++) s[g] += fp[g * HWl + i] * v;
            cnt += v;
        }
    }
    const float Cn = block_sum_1024(cnt, red);
#pragma unroll
    for (int g = 0; g < G; g++) {
        const float S = block_sum_1024(s[g], red);
        if (threadIdx.x == 0) sums[plane0 + g] = S;
    }
    if (threadIdx.x == 0 && plane0 % C == 0) counts[n] = Cn;
}

__global__ __launch_bounds__(256) void spade_feat_assemble_kernel(const float* __restrict__ fu, const float* __restrict__ fl,
                                                                  const float* __restrict__ mu, const float* __restrict__ ml,
                                                                  const float* __restrict__ du, const float* __restrict__ dl,
                                                                  const float* __restrict__ su, const float* __restrict__ sl,
                                                                  const float* __restrict__ cu, const float* __restrict__ cl,
                                                                  float* __restrict__ out, int C, int H, int W, int vec) {
    if (vec) {
        // round 5: blockIdx.x = image, the channel loop INSIDE -- the four thresholded masks of a pixel quad (8 loads, 128 bytes) are read once for all C planes
        // instead of once per plane (they were 2.7x the bytes of the feature traffic: 250 us for 0.4 GB of HBM traffic at N = 8, C = 64, 256^2)
        const int n = blockIdx.x;
        const float nu = cu[n] > 10.f ? cu[n] : 65536.f, nl = cl[n] > 10.f ? cl[n] : 65536.f;      // the reference's literal 256 * 256
        const int64_t mb = (int64_t)n * 4 * H * W, HWl = (int64_t)H * W;
        const int W4 = W / 4, n4 = H * W4;
        for (int i = blockIdx.y * 256 + threadIdx.x; i < n4; i += gridDim.y * 256) {
            const int y = i / W4, x4 = i - y * W4;
            const int64_t mi = mb + (int64_t)(2 * y) * (2 * W) + 8 * x4;
            const f32x4 m_u = even_mask4(mu + mi), m_l = even_mask4(ml + mi), d_u = even_mask4(du + mi), d_l = even_mask4(dl + mi);
            f32x4 r_u, r_l;
#pragma unroll
            for (int k = 0; k < 4; k++) { r_u[k] = m_u[k] - m_u[k] * d_u[k]; r_l[k] = m_l[k] - m_l[k] * d_l[k]; }
            const int cz = (C + gridDim.z - 1) / gridDim.z, c_lo = blockIdx.z * cz, c_hi = c_lo + cz < C ? c_lo + cz : C;      // this workgroup's share of the channels
#pragma unroll 4
            for (int c = c_lo; c < c_hi; c++) {
                const int plane = n * C + c;
                const float au = su[plane] / nu, al = sl[plane] / nl;
                const int64_t pb = (int64_t)plane * HWl;
                const f32x4 a = ((const f32x4*)(fu + pb))[i], b = ((const f32x4*)(fl + pb))[i];
                f32x4 o;
#pragma unroll
                for (int k = 0; k < 4; k++) {
                    const float b_u = a[k] * (1.f - r_u[k]) + au * r_u[k];
                    const float b_l = b[k] * (1.f - r_l[k]) + al * r_l[k];
                    o[k] = b_u * m_u[k] + b_l * m_l[k];
                }
                ((f32x4*)(out + pb))[i] = o;
            }
        }
        return;
    }
    const int plane = blockIdx.x, n = plane / C;
    const float nu = cu[n] > 10.f ? cu[n] : 65536.f, nl = cl[n] > 10.f ? cl[n] : 65536.f;
    const float au = su[plane] / nu, al = sl[plane] / nl;
    const int64_t pb = (int64_t)plane * H * W, mb = (int64_t)n * 4 * H * W;
    for (int i = blockIdx.y * 256 + threadIdx.x; i < H * W; i += gridDim.y * 256) {
        const int y = i / W, x = i - y * W;
        const int64_t mi = mb + (int64_t)(2 * y) * (2 * W) + 2 * x;
        const float m_u = mu[mi] > 0.9f ? 1.f : 0.f, m_l = ml[mi] > 0.9f ? 1.f : 0.f;
        const float v_u = (m_u != 0.f && du[mi] > 0.9f) ? 1.f : 0.f, v_l = (m_l != 0.f && dl[mi] > 0.9f) ? 1.f : 0.f;
        const float r_u = m_u - v_u, r_l = m_l - v_l;
        const float b_u = fu[pb + i] * (1.f - r_u) + au * r_u;
        const float b_l = fl[pb + i] * (1.f - r_l) + al * r_l;
        out[pb + i] = b_u * m_u + b_l * m_l;
    }
}

// ------------------------------------------------------------------ split-K finish
// y[n, co, oy*osy+ooy, ox*osx+oox] = epilogue(sum_z ws[z][n, co, oy, ox]): the fused epilogue of pg_conv2d_forward applied to the
// sum of the partial convolutions (fixed order z = 0, 1, ...: deterministic, no atomics).
__global__ __launch_bounds__(256) void splitk_finish_kernel(const float* __restrict__ ws, float* __restrict__ y, int ksplit, int64_t slice,
                                                            int N, int Cout, int OH, int OW, int64_t ys0, int64_t ys1, int64_t ys2, int64_t ys3,
                                                            int osy, int osx, int ooy, int oox, pg_conv2d_fusion f) {
    const int64_t total = (int64_t)N * Cout * OH * OW;
    const float slope = f.act == PG_ACT_LINEAR ? 1.f : (f.act == PG_ACT_RELU ? 0.f : f.alpha);
    const float cl = f.clamp >= 0.f ? f.clamp : __builtin_inff();
    for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < total; i += (int64_t)gridDim.x * 256) {
        const int ox = (int)(i % OW), oy = (int)((i / OW) % OH), co = (int)((i / ((int64_t)OW * OH)) % Cout), n = (int)(i / ((int64_t)OW * OH * Cout));
        float v = 0.f;
        for (int z = 0; z < ksplit; z++) v += ws[(int64_t)z * slice + i];
        if (f.out_scale) v *= f.out_scale[(int64_t)n * Cout + co];
        if (f.noise) v += f.noise[n * f.noise_batch_stride + (int64_t)oy * OW + ox] * f.noise_gain;
        if (f.bias) v += f.bias[co];
        v = v > 0.f ? v : v * slope;
        v = fminf(fmaxf(v * f.gain, -cl), cl);
        const int64_t off = n * ys0 + co * ys1 + (int64_t)(oy * osy + ooy) * ys2 + (int64_t)(ox * osx + oox) * ys3;
        if (f.residual) v += f.residual[off];
        y[off] = v;
    }
}

inline int round_up(int v, int m) { return (v + m - 1) / m * m; }

}  // namespace

PG_EXPORT int pg_conv2d_abi_version(void) { return PG_ABI_VERSION; }

// See include/pasta_gan_ops.h: CUs left to other work (RCCL's channels); returns the CU count the plugin's grids are sized for from now on.
PG_EXPORT int pg_conv2d_reserve_cus(int n) {
    pg::reserved_cus().store(n > 0 ? n : 0, std::memory_order_relaxed);
    return pg::num_cu();
}

PG_EXPORT int64_t pg_conv2d_packed_size(int Cout, int Cin, int KH, int KW) {
    if (Cout <= 0 || Cin <= 0 || KH <= 0 || KW <= 0) return 0;
    return (int64_t)round_up(Cin, 16) * KH * KW * round_up(Cout, 32);
}

PG_EXPORT int pg_conv2d_pack_weight(const float* w, float* packed, int Cout, int Cin, int KH, int KW,
                                    float scale, int flip_hw, int transpose_oi, void* stream) {
    if (!w || !packed || Cout <= 0 || Cin <= 0 || KH <= 0 || KW <= 0) return PG_ERR_INVALID_ARG;
    const int CinP = round_up(Cin, 16), CoutP = round_up(Cout, 32);
    const int64_t total = (int64_t)CinP * KH * KW * CoutP;
    int64_t blocks = (total + 255) / 256;
    if (blocks > pg::max_stream_blocks()) blocks = pg::max_stream_blocks();
    hipLaunchKernelGGL(pack_weight_kernel, dim3((unsigned)blocks), dim3(256), 0, (hipStream_t)stream, w, packed, Cout, Cin, KH, KW, CinP, CoutP, scale, flip_hw, transpose_oi);
    return pg::launch_status();
}

static int conv_forward(int winograd, const float* x, const float* packed_w, float* y,
                        int N, int Cin, int H, int W, int Cout, int KH, int KW,
                        int stride, int pad_y, int pad_x, int OH, int OW,
                        const int64_t ystride[4], int out_step_y, int out_step_x, int out_off_y, int out_off_x,
                        const pg_conv2d_fusion* fusion, void* stream, float* workspace = nullptr, int ksplit = 1) {
    if (!x || !packed_w || !y || !ystride) return PG_ERR_INVALID_ARG;
    if (N <= 0 || Cin <= 0 || H <= 0 || W <= 0 || Cout <= 0 || OH <= 0 || OW <= 0) return PG_ERR_INVALID_ARG;
    if (out_step_y < 1 || out_step_x < 1) return PG_ERR_INVALID_ARG;
    // one image of x is addressed through a 32-bit buffer descriptor: byte offsets (and the sentinel 2^31) must stay below 2^31
    if ((int64_t)Cin * H * W * 4 > 0x7fffffffLL || (int64_t)16 * H * W * 4 > 0x7fffffffLL) return PG_ERR_TOO_LARGE;
    {   // the epilogue indexes y (and residual / noise) with 32-bit element offsets
        int64_t ext = 1 + (int64_t)(N - 1) * ystride[0] + (int64_t)(Cout - 1) * ystride[1] +
                      ((int64_t)(OH - 1) * out_step_y + out_off_y) * ystride[2] + ((int64_t)(OW - 1) * out_step_x + out_off_x) * ystride[3];
        if (ext > 0x7fffffffLL || (int64_t)N * OH * OW > 0x7fffffffLL) return PG_ERR_TOO_LARGE;
    }
    ConvParams p;
    p.ksplit = 1; p.kpart = 0; p.ws_slice = 0; p.wino_gmap = 0; p.rowpair = 0;
    p.rowpair_pack = 1;                 // every packed 7x7 / Cin = 3 kernel comes from pg_conv2d_pack_weight, i.e. is in the ROWPAIR form
    p.x = x; p.wp = packed_w; p.y = y;
    p.N = N; p.Cin = Cin; p.H = H; p.W = W; p.Cout = Cout; p.CoutP = round_up(Cout, 32); p.OH = OH; p.OW = OW;
    p.pad_y = pad_y; p.pad_x = pad_x;
    for (int i = 0; i < 4; i++) p.ys[i] = ystride[i];
    p.osy = out_step_y; p.osx = out_step_x; p.ooy = out_off_y; p.oox = out_off_x;
    if (fusion) {
        p.f = *fusion;
    } else {
        pg_conv2d_fusion z = {};
        z.in_clamp = -1.f; z.clamp = -1.f;
        p.f = z;
    }
    if (p.f.in_gain == 0.f) p.f.in_gain = 1.f;
    if (p.f.gain == 0.f) p.f.gain = 1.f;
    if (p.f.in_act == 0) p.f.in_act = PG_ACT_LINEAR;
    if (p.f.act == 0) p.f.act = PG_ACT_LINEAR;
    if (p.f.in_act < PG_ACT_LINEAR || p.f.in_act > PG_ACT_SWISH || p.f.act < PG_ACT_LINEAR || p.f.act > PG_ACT_SWISH) return PG_ERR_INVALID_ARG;
    if (p.f.in_act > PG_ACT_LRELU || p.f.act > PG_ACT_LRELU) return PG_ERR_UNSUPPORTED;   // fused stages: linear / relu / lrelu only
    if (p.f.spade_x) {
        if (!p.f.spade_mean || !p.f.spade_rstd) return PG_ERR_INVALID_ARG;
        if (Cout % 64 != 0 || out_step_y != 1 || out_step_x != 1) return PG_ERR_UNSUPPORTED;   // 32 gamma + 32 beta rows per 64-row tile
    }
    if (p.f.x2 && (p.f.cin_split <= 0 || p.f.cin_split >= Cin || p.f.cin_split % 16 != 0)) return PG_ERR_INVALID_ARG;
    if (!p.f.x2) p.f.cin_split = 0;
    if (p.f.in_bias) return PG_ERR_UNSUPPORTED;   // the prologue runs on the zero-padded tile: act(0 + b) != 0 would corrupt the padding
    p.in_xform = (p.f.in_act != PG_ACT_LINEAR || p.f.in_gain != 1.f || p.f.in_clamp >= 0.f) ? 1 : 0;
    if (p.in_xform && (!(p.f.in_gain > 0.f) || p.f.in_alpha < 0.f || p.f.in_alpha > 1.f)) return PG_ERR_UNSUPPORTED;
    hipStream_t s = (hipStream_t)stream;

    if (p.f.stats_partial && winograd != 2 && winograd != 4) return PG_ERR_UNSUPPORTED;      // output statistics: the F(4x4) one-workgroup kernel's plain tail only
    if (winograd) {
        if (p.f.x2) return PG_ERR_UNSUPPORTED;                  // two-source launches stay on the direct kernel
        if (pad_x < 0 || pad_x > 4) return PG_ERR_UNSUPPORTED;  // the LDS halo row starts 4 columns left of the tile
        p.CoutP = round_up(Cout, 64);
        return winograd == 4 ? pgconv::launch_wino4x3(p, s) : winograd == 3 ? pgconv::launch_wino4b(p, s) : winograd == 2 ? pgconv::launch_wino4(p, s) : pgconv::launch_wino(p, s);
    }
    pg_conv2d_fusion tail = p.f;
    const int64_t slice = (int64_t)N * Cout * OH * OW;
    if (ksplit > 1) {
        // split-K: `ksplit` workgroups per output tile each reduce a share of the input channels into their own workspace slice
        // (prologue kept, epilogue off), then one elementwise pass sums the slices and applies the epilogue
        const int kc = pgconv::kc_for(KH, KW, stride);
        const int nchunks = round_up(Cin, kc) / kc;
        if (!workspace || p.f.spade_x || p.f.x2 || nchunks % ksplit != 0 || slice * ksplit > 0x7fffffffLL) return PG_ERR_INVALID_ARG;
        p.ksplit = ksplit; p.kpart = nchunks / ksplit * kc; p.ws_slice = slice;
        p.y = workspace;
        p.ys[0] = (int64_t)Cout * OH * OW; p.ys[1] = (int64_t)OH * OW; p.ys[2] = OW; p.ys[3] = 1;
        p.osy = p.osx = 1; p.ooy = p.oox = 0;
        p.f.out_scale = nullptr; p.f.noise = nullptr; p.f.bias = nullptr; p.f.residual = nullptr;
        p.f.act = PG_ACT_LINEAR; p.f.gain = 1.f; p.f.clamp = -1.f;
    }
    int st = PG_ERR_UNSUPPORTED;
    if (stride == 1) {
        if (KH == 3 && KW == 3) st = pgconv::launch_k3s1(p, s);
        else if (KH == 1 && KW == 1) { st = pgconv::launch_s1x1(p, s); if (st == PG_ERR_UNSUPPORTED) st = pgconv::launch_k1s1(p, s); }
        else if (KH == 2 && KW == 2) st = pgconv::launch_k2x2(p, s);     // polyphase pieces of a stride-2 transposed 3x3
        else if (KH == 2 && KW == 1) st = pgconv::launch_k2x1(p, s);
        else if (KH == 1 && KW == 2) st = pgconv::launch_k1x2(p, s);
        else if (KH == 7 && KW == 7) st = pgconv::launch_k7s1(p, s);
    } else if (stride == 2) {
        if (KH == 3 && KW == 3) st = pgconv::launch_k3s2(p, s);
        else if (KH == 1 && KW == 1) st = pgconv::launch_k1s2(p, s);
    }
    if (st != PG_OK || ksplit <= 1) return st;
    int64_t blocks = (slice + 255) / 256;
    if (blocks > pg::max_stream_blocks()) blocks = pg::max_stream_blocks();
    hipLaunchKernelGGL(splitk_finish_kernel, dim3((unsigned)blocks), dim3(256), 0, s, workspace, y, ksplit, slice, N, Cout, OH, OW,
                       ystride[0], ystride[1], ystride[2], ystride[3], out_step_y, out_step_x, out_off_y, out_off_x, tail);
    return pg::launch_status();
}

PG_EXPORT int pg_conv2d_forward(const float* x, const float* packed_w, float* y,
                                int N, int Cin, int H, int W, int Cout, int KH, int KW,
                                int stride, int pad_y, int pad_x, int OH, int OW,
                                const int64_t ystride[4], int out_step_y, int out_step_x, int out_off_y, int out_off_x,
                                const pg_conv2d_fusion* fusion, void* stream) {
    return conv_forward(0, x, packed_w, y, N, Cin, H, W, Cout, KH, KW, stride, pad_y, pad_x, OH, OW,
                        ystride, out_step_y, out_step_x, out_off_y, out_off_x, fusion, stream);
}

PG_EXPORT int pg_conv2d_splitk_plan(int N, int Cin, int OH, int OW, int Cout, int KH, int KW, int stride) {
    if (N <= 0 || Cin <= 0 || OH <= 0 || OW <= 0 || Cout <= 0) return 1;
    if (!((stride == 1 && ((KH == 3 && KW == 3) || (KH == 1 && KW == 1) || (KH == 2 && KW == 2) || (KH == 2 && KW == 1) || (KH == 1 && KW == 2) || (KH == 7 && KW == 7))) ||
          (stride == 2 && ((KH == 3 && KW == 3) || (KH == 1 && KW == 1))))) return 1;
    const int kc = pgconv::kc_for(KH, KW, stride);
    const int nchunks = round_up(Cin, kc) / kc;
    const int64_t tiles64 = (int64_t)N * ((OW + pgconv::TW - 1) / pgconv::TW) * ((OH + pgconv::TH - 1) / pgconv::TH) * ((round_up(Cout, 32) + 63) / 64);
    if (tiles64 >= pg::num_cu() || nchunks < 8) return 1;            // enough tiles to occupy the chip, or too short a K loop to share
    int best = 1;                                                  // largest divisor of nchunks that keeps >= 4 chunks per share and does not
    for (int k = 2; k <= 16; k++)                                  // overshoot ~4 workgroups per CU
        if (nchunks % k == 0 && nchunks / k >= 4 && tiles64 * k <= 4 * pg::num_cu()) best = k;
    return best;
}

PG_EXPORT int pg_conv2d_forward_splitk(const float* x, const float* packed_w, float* y,
                                       int N, int Cin, int H, int W, int Cout, int KH, int KW,
                                       int stride, int pad_y, int pad_x, int OH, int OW,
                                       const int64_t ystride[4], int out_step_y, int out_step_x, int out_off_y, int out_off_x,
                                       const pg_conv2d_fusion* fusion, float* workspace, int ksplit, void* stream) {
    if (ksplit < 1 || (ksplit > 1 && !workspace)) return PG_ERR_INVALID_ARG;
    return conv_forward(0, x, packed_w, y, N, Cin, H, W, Cout, KH, KW, stride, pad_y, pad_x, OH, OW,
                        ystride, out_step_y, out_step_x, out_off_y, out_off_x, fusion, stream, workspace, ksplit);
}

PG_EXPORT int64_t pg_conv2d_winograd_packed_size(int Cout, int Cin) {
    if (Cout <= 0 || Cin <= 0) return 0;
    return (int64_t)16 * round_up(Cin, 16) * round_up(Cout, 64);
}

PG_EXPORT int pg_conv2d_winograd_pack_weight(const float* w, float* packed, int Cout, int Cin,
                                             float scale, int flip_hw, int transpose_oi, void* stream) {
    if (!w || !packed || Cout <= 0 || Cin <= 0) return PG_ERR_INVALID_ARG;
    const int CinP = round_up(Cin, 16), CoutP = round_up(Cout, 64);
    const int64_t total = (int64_t)CinP * CoutP;
    int64_t blocks = (total + 255) / 256;
    if (blocks > pg::max_stream_blocks()) blocks = pg::max_stream_blocks();
    hipLaunchKernelGGL(wino_pack_kernel, dim3((unsigned)blocks), dim3(256), 0, (hipStream_t)stream, w, packed, Cout, Cin, CinP, CoutP, scale, flip_hw, transpose_oi);
    return pg::launch_status();
}

PG_EXPORT int pg_conv2d_winograd_forward(const float* x, const float* packed_u, float* y,
                                         int N, int Cin, int H, int W, int Cout, int pad_y, int pad_x, int OH, int OW,
                                         const int64_t ystride[4], const pg_conv2d_fusion* fusion, void* stream) {
    return conv_forward(1, x, packed_u, y, N, Cin, H, W, Cout, 3, 3, 1, pad_y, pad_x, OH, OW, ystride, 1, 1, 0, 0, fusion, stream);
}

PG_EXPORT int64_t pg_conv2d_winograd4_packed_size(int Cout, int Cin) {
    if (Cout <= 0 || Cin <= 0) return 0;
    return (int64_t)36 * round_up(Cin, 16) * round_up(Cout, 64);
}

PG_EXPORT int pg_conv2d_winograd4_pack_weight(const float* w, float* packed, int Cout, int Cin,
                                              float scale, int flip_hw, int transpose_oi, void* stream) {
    if (!w || !packed || Cout <= 0 || Cin <= 0) return PG_ERR_INVALID_ARG;
    const int CinP = round_up(Cin, 16), CoutP = round_up(Cout, 64);
    const int64_t total = (int64_t)CinP * CoutP;
    int64_t blocks = (total + 255) / 256;
    if (blocks > pg::max_stream_blocks()) blocks = pg::max_stream_blocks();
    hipLaunchKernelGGL(wino4_pack_kernel, dim3((unsigned)blocks), dim3(256), 0, (hipStream_t)stream, w, packed, Cout, Cin, CinP, CoutP, scale, flip_hw, transpose_oi);
    return pg::launch_status();
}

PG_EXPORT int pg_conv2d_winograd4_forward(const float* x, const float* packed_u, float* y,
                                          int N, int Cin, int H, int W, int Cout, int pad_y, int pad_x, int OH, int OW,
                                          const int64_t ystride[4], const pg_conv2d_fusion* fusion, void* stream) {
    return conv_forward(2, x, packed_u, y, N, Cin, H, W, Cout, 3, 3, 1, pad_y, pad_x, OH, OW, ystride, 1, 1, 0, 0, fusion, stream);
}

PG_EXPORT int64_t pg_conv2d_winograd4x3_packed_size(int Cout, int Cin) {      // in float32 units (6 bytes per transformed weight)
    if (Cout <= 0 || Cin <= 0) return 0;
    return (int64_t)54 * round_up(Cin, 16) * round_up(Cout, 64);
}

PG_EXPORT int pg_conv2d_winograd4x3_pack_weight(const float* w, float* packed, int Cout, int Cin,
                                                float scale, int flip_hw, int transpose_oi, void* stream) {
    if (!w || !packed || Cout <= 0 || Cin <= 0) return PG_ERR_INVALID_ARG;
    const int CinP = round_up(Cin, 16), CoutP = round_up(Cout, 64);
    const int64_t total = (int64_t)CinP * CoutP;
    int64_t blocks = (total + 255) / 256;
    if (blocks > pg::max_stream_blocks()) blocks = pg::max_stream_blocks();
    hipLaunchKernelGGL(wino4x3_pack_kernel, dim3((unsigned)blocks), dim3(256), 0, (hipStream_t)stream, w, (unsigned short*)packed, Cout, Cin, CinP, CoutP, scale, flip_hw, transpose_oi);
    return pg::launch_status();
}

PG_EXPORT int pg_conv2d_winograd4x3_forward(const float* x, const float* packed_u, float* y,
                                            int N, int Cin, int H, int W, int Cout, int pad_y, int pad_x, int OH, int OW,
                                            const int64_t ystride[4], const pg_conv2d_fusion* fusion, void* stream) {
    return conv_forward(4, x, packed_u, y, N, Cin, H, W, Cout, 3, 3, 1, pad_y, pad_x, OH, OW, ystride, 1, 1, 0, 0, fusion, stream);
}

PG_EXPORT int pg_conv2d_winograd4b_pack_weight(const float* w, float* packed, int Cout, int Cin,
                                               float scale, int flip_hw, int transpose_oi, void* stream) {
    if (!w || !packed || Cout <= 0 || Cin <= 0) return PG_ERR_INVALID_ARG;
    const int CinP = round_up(Cin, 16), CoutP = round_up(Cout, 64);
    const int64_t total = (int64_t)CinP * CoutP;
    int64_t blocks = (total + 255) / 256;
    if (blocks > pg::max_stream_blocks()) blocks = pg::max_stream_blocks();
    hipLaunchKernelGGL(wino4b_pack_kernel, dim3((unsigned)blocks), dim3(256), 0, (hipStream_t)stream, w, packed, Cout, Cin, CinP, CoutP, scale, flip_hw, transpose_oi);
    return pg::launch_status();
}

PG_EXPORT int pg_conv2d_winograd4b_forward(const float* x, const float* packed_u, float* y,
                                           int N, int Cin, int H, int W, int Cout, int pad_y, int pad_x, int OH, int OW,
                                           const int64_t ystride[4], const pg_conv2d_fusion* fusion, void* stream) {
    return conv_forward(3, x, packed_u, y, N, Cin, H, W, Cout, 3, 3, 1, pad_y, pad_x, OH, OW, ystride, 1, 1, 0, 0, fusion, stream);
}

// Streaming 1x1 head for fp32 NCHW tensors (the ToRGB / parsing heads: Cout <= 8, networks.py:287-316): a thread owns 4 adjacent
// pixels and walks the input channels with 16-byte loads (one contiguous 1 KB per wave-instruction, 8 channels in flight), the
// Cout x Cin weights of the block's image -- already multiplied by its styles -- sit in LDS as [Cin][8] (two broadcast reads per
// channel).  HBM-bound: 4*Cin bytes read + 4*Cout written (+ 4*Cout skip image) per pixel; the tiled MFMA kernel pads such a
// layer to 32 output channels and runs it at 3 TB/s.
namespace {
typedef float f32x4s __attribute__((ext_vector_type(4)));
template <int COUT>
__global__ __launch_bounds__(256) void conv1x1_small_f32_kernel(const float* __restrict__ x, const float* __restrict__ w, const float* __restrict__ styles,
                                                                const float* __restrict__ bias, const float* __restrict__ skip, float* __restrict__ y,
                                                                int Cin, int64_t HW4, float scale, float clamp) {
    extern __shared__ __attribute__((aligned(16))) float wl[];          // [Cin][8]
    const int n = blockIdx.y;
    for (int e = threadIdx.x; e < 8 * Cin; e += 256) {
        const int c = e >> 3, o = e & 7;
        wl[e] = o < COUT ? w[o * Cin + c] * scale * (styles ? styles[(int64_t)n * Cin + c] : 1.f) : 0.f;
    }
    __syncthreads();
    const float cl = clamp >= 0.f ? clamp : __builtin_inff();
    const f32x4s* xn = (const f32x4s*)x + (int64_t)n * Cin * HW4;
    for (int64_t p = (int64_t)blockIdx.x * 256 + threadIdx.x; p < HW4; p += (int64_t)gridDim.x * 256) {
        f32x4s acc[COUT];
#pragma unroll
        for (int o = 0; o < COUT; o++) acc[o] = (f32x4s){0.f, 0.f, 0.f, 0.f};
        auto mac = [&](const f32x4s& xv, int c) __attribute__((always_inline)) {
            const f32x4s w0 = *(const f32x4s*)(wl + c * 8), w1 = *(const f32x4s*)(wl + c * 8 + 4);
#pragma unroll
            for (int o = 0; o < COUT; o++) acc[o] += xv * (o < 4 ? w0[o & 3] : w1[o & 3]);
        };
        int c0 = 0;
        for (; c0 + 8 <= Cin; c0 += 8) {                                  // eight channel planes in flight, no branch between them
            f32x4s v[8];
#pragma unroll
            for (int u = 0; u < 8; u++) v[u] = xn[(int64_t)(c0 + u) * HW4 + p];
            __builtin_amdgcn_sched_barrier(0);
#pragma unroll
            for (int u = 0; u < 8; u++) mac(v[u], c0 + u);
        }
        for (; c0 < Cin; c0++) mac(xn[(int64_t)c0 * HW4 + p], c0);
#pragma unroll
        for (int o = 0; o < COUT; o++) {
            f32x4s r = acc[o] + (bias ? bias[o] : 0.f);
#pragma unroll
            for (int e = 0; e < 4; e++) r[e] = fminf(fmaxf(r[e], -cl), cl);
            const int64_t off = ((int64_t)n * COUT + o) * HW4 + p;
            if (skip) r += ((const f32x4s*)skip)[off];
            ((f32x4s*)y)[off] = r;
        }
    }
}


// The same head on SMALL images (round 5).  The form above gives a thread four pixels and walks ALL input channels: on the 4^2 ... 64^2 images of the low
// blocks (512 channels) that is one to four workgroups per image, each behind a serial LDS weight prologue (16 dependent rounds) and 64 dependent groups of
// eight loads -- 32-47 us per launch for 0.1-67 MB, five launches per config-2 step.  Here a workgroup takes QB pixel quads of one image and its 256 / QB
// thread columns split the channels (thread = quad q, slice cs: channels cs, cs + CS, ...; its modulated weights come straight from global memory, nothing
// is staged), the partial sums meet in LDS and QB x COUT threads add them up, apply bias / clamp / skip and store.  Every load of the launch is in flight
// within two rounds.
template <int COUT, int QB>      // QB * COUT <= 256
__global__ __launch_bounds__(256) void conv1x1_small_f32_split_kernel(const float* __restrict__ x, const float* __restrict__ w, const float* __restrict__ styles,
                                                                      const float* __restrict__ bias, const float* __restrict__ skip, float* __restrict__ y,
                                                                      int Cin, int64_t HW4, float scale, float clamp) {
    constexpr int CS = 256 / QB;
    static_assert(QB * COUT <= 256, "one output thread per (quad, channel)");
    __shared__ f32x4s part[CS][COUT][QB];
    const int n = blockIdx.y;
    const int q = threadIdx.x % QB, cs = threadIdx.x / QB;
    const int64_t p = (int64_t)blockIdx.x * QB + q;
    const bool live = p < HW4;
    const f32x4s* xn = (const f32x4s*)x + (int64_t)n * Cin * HW4 + (live ? p : 0);
    const float* sn = styles ? styles + (int64_t)n * Cin : nullptr;
    f32x4s acc[COUT];
#pragma unroll
    for (int o = 0; o < COUT; o++) acc[o] = (f32x4s){0.f, 0.f, 0.f, 0.f};
    for (int c0 = cs; c0 < Cin; c0 += 8 * CS) {                          // eight channel planes of this slice in flight
        f32x4s v[8];
        float wv[8][COUT];
#pragma unroll
        for (int u = 0; u < 8; u++) {
            const int c = c0 + u * CS;
            const bool ok = c < Cin;
            v[u] = ok ? xn[(int64_t)c * HW4] : (f32x4s){0.f, 0.f, 0.f, 0.f};
            const float sc = ok ? scale * (sn ? sn[c] : 1.f) : 0.f;
#pragma unroll
            for (int o = 0; o < COUT; o++) wv[u][o] = ok ? w[o * Cin + c] * sc : 0.f;
        }
#pragma unroll
        for (int u = 0; u < 8; u++)
#pragma unroll
            for (int o = 0; o < COUT; o++) acc[o] += v[u] * wv[u][o];
    }
#pragma unroll
    for (int o = 0; o < COUT; o++) part[cs][o][q] = acc[o];
    __syncthreads();
    if (threadIdx.x < QB * COUT) {
        const int oq = threadIdx.x % QB, o = threadIdx.x / QB;
        const int64_t po = (int64_t)blockIdx.x * QB + oq;
        f32x4s r = part[0][o][oq];
#pragma unroll 8
        for (int z = 1; z < CS; z++) r += part[z][o][oq];
        r += bias ? bias[o] : 0.f;
        const float cl = clamp >= 0.f ? clamp : __builtin_inff();
#pragma unroll
        for (int e = 0; e < 4; e++) r[e] = fminf(fmaxf(r[e], -cl), cl);
        if (po < HW4) {
            const int64_t off = ((int64_t)n * COUT + o) * HW4 + po;
            if (skip) r += ((const f32x4s*)skip)[off];
            ((f32x4s*)y)[off] = r;
        }
    }
}


// 3x3 convolution of a ONE-channel image (the SPADE blocks' first layer on the parsing / mask map, networks.py:1708-1712:
// 1 -> 64 channels + ReLU): a 9-tap stencil per output channel.  A thread keeps the 3 x 6 samples its 4 adjacent pixels touch
// in registers and walks the output channels (weights are uniform: scalar loads), one 16-byte store each -- the kernel is a
// pure output stream (4*Cout bytes per pixel); the MFMA kernel would pad the single input channel to a 16-channel K chunk.
template <bool RELU>
__global__ __launch_bounds__(256) void conv3x3_cin1_kernel(const float* __restrict__ x, const float* __restrict__ w, float* __restrict__ y,
                                                           int H, int W, int Cout, float scale) {
    const int n = blockIdx.y;
    const int W4 = W >> 2;
    const int64_t HW = (int64_t)H * W;
    const float* xn = x + (int64_t)n * HW;
    for (int64_t q = (int64_t)blockIdx.x * 256 + threadIdx.x; q < (int64_t)H * W4; q += (int64_t)gridDim.x * 256) {
        const int oy = (int)(q / W4), ox = 4 * (int)(q % W4);
        float v[3][6];
#pragma unroll
        for (int r = 0; r < 3; r++) {
            const int gy = oy + r - 1;
            const bool row_ok = gy >= 0 && gy < H;
            const float* row = xn + (int64_t)(row_ok ? gy : 0) * W;
            const f32x4s m = *(const f32x4s*)(row + ox);
            const float l = row[ox > 0 ? ox - 1 : 0], rr = row[ox + 4 < W ? ox + 4 : W - 1];
            v[r][0] = (row_ok && ox > 0) ? l : 0.f;
            v[r][5] = (row_ok && ox + 4 < W) ? rr : 0.f;
#pragma unroll
            for (int e = 0; e < 4; e++) v[r][1 + e] = row_ok ? m[e] : 0.f;
        }
        f32x4s* yo = (f32x4s*)(y + (int64_t)n * Cout * HW + (int64_t)oy * W + ox);
        for (int co = 0; co < Cout; co++) {
            const float* wc = w + co * 9;
            f32x4s acc = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
            for (int r = 0; r < 3; r++)
#pragma unroll
                for (int k = 0; k < 3; k++) {
                    const float wk = wc[r * 3 + k] * scale;
#pragma unroll
                    for (int e = 0; e < 4; e++) acc[e] = fmaf(v[r][e + k], wk, acc[e]);
                }
            if (RELU) {
#pragma unroll
                for (int e = 0; e < 4; e++) acc[e] = fmaxf(acc[e], 0.f);
            }
            yo[(int64_t)co * (HW >> 2)] = acc;
        }
    }
}
}  // namespace

/* y[n,o,p] = clamp(sum_c x[n,c,p] * w[o,c] * scale * styles[n,c] + bias[o]) + skip[n,o,p]   (ToRGB: networks.py:306-316, demodulate=False) */
PG_EXPORT int pg_conv1x1_small(const float* x, const float* w, const float* styles, const float* bias, const float* skip, float* y,
                               int N, int Cin, int64_t HW, int Cout, float scale, float clamp, void* stream) {
    if (!x || !w || !y || N <= 0 || Cin <= 0 || HW <= 0 || Cout <= 0) return PG_ERR_INVALID_ARG;
    if (Cout > 8 || HW % 4 != 0 || !pg::aligned16(x) || !pg::aligned16(y) || (skip && !pg::aligned16(skip)) || (size_t)Cin * 32 > 64 * 1024) return PG_ERR_UNSUPPORTED;
    if (N > 65535) return PG_ERR_TOO_LARGE;
    const int64_t HW4 = HW / 4;
    hipStream_t s = (hipStream_t)stream;
    // small images: the channel-split form wherever one workgroup per 256 quads would leave CUs idle (PG_HEAD32_SPLIT=0: never, A/B)
    static const bool split_on = [] { const char* e = getenv("PG_HEAD32_SPLIT"); return !e || atoi(e) != 0; }();
    // (measured at N = 8, 512 / 256 channels: 4^2 ... 16^2 32 -> 6.6 us, 32^2 34 -> 11, 64^2 47 -> 25; at 128^2, where the first form has half a workgroup per CU, it
    // wins 27 : 41 -- hence "fewer than half the CUs")
    if (split_on && (int64_t)N * ((HW4 + 255) / 256) < (int64_t)pg::num_cu() / 2 && Cin >= 64) {
        int qb = 4;                                                        // the wider quad block where it still gives every CU two workgroups
        if ((int64_t)N * ((HW4 + 15) / 16) >= 2 * (int64_t)pg::num_cu()) qb = 16;
        const dim3 g2((unsigned)((HW4 + qb - 1) / qb), (unsigned)N);
#define PG_SPLIT(C) case C: \
        if (qb == 16) hipLaunchKernelGGL((conv1x1_small_f32_split_kernel<C, 16>), g2, dim3(256), 0, s, x, w, styles, bias, skip, y, Cin, HW4, scale, clamp); \
        else hipLaunchKernelGGL((conv1x1_small_f32_split_kernel<C, 4>), g2, dim3(256), 0, s, x, w, styles, bias, skip, y, Cin, HW4, scale, clamp); \
        break;
        switch (Cout) { PG_SPLIT(1) PG_SPLIT(2) PG_SPLIT(3) PG_SPLIT(4) PG_SPLIT(5) PG_SPLIT(6) PG_SPLIT(7) PG_SPLIT(8) }
#undef PG_SPLIT
        return pg::launch_status();
    }
    int64_t bx = (HW4 + 255) / 256;
    const int64_t cap = (int64_t)pg::num_cu() * 8 / N + 1;
    if (bx > cap) bx = cap;
    const dim3 grid((unsigned)bx, (unsigned)N);
    const size_t lds = (size_t)Cin * 32;
#define PG_SMALL(C) case C: hipLaunchKernelGGL((conv1x1_small_f32_kernel<C>), grid, dim3(256), lds, s, x, w, styles, bias, skip, y, Cin, HW4, scale, clamp); break;
    switch (Cout) { PG_SMALL(1) PG_SMALL(2) PG_SMALL(3) PG_SMALL(4) PG_SMALL(5) PG_SMALL(6) PG_SMALL(7) PG_SMALL(8) }
#undef PG_SMALL
    return pg::launch_status();
}

/* y = act(conv2d(x, w * scale, padding=1)) for a single input channel: x [N,1,H,W], w [Cout,1,3,3] (cross-correlation, as
 * F.conv2d), y [N,Cout,H,W]; act = PG_ACT_LINEAR or PG_ACT_RELU with gain 1.  W % 4 == 0 and 16-byte aligned tensors. */
PG_EXPORT int pg_conv3x3_cin1(const float* x, const float* w, float* y, int N, int H, int W, int Cout, float scale, int act, void* stream) {
    if (!x || !w || !y || N <= 0 || H <= 0 || W <= 0 || Cout <= 0) return PG_ERR_INVALID_ARG;
    if (act != PG_ACT_LINEAR && act != PG_ACT_RELU) return PG_ERR_UNSUPPORTED;
    if (W % 4 != 0 || !pg::aligned16(x) || !pg::aligned16(y)) return PG_ERR_UNSUPPORTED;
    if (N > 65535 || (int64_t)Cout * H * W > 0x7fffffffLL * 4) return PG_ERR_TOO_LARGE;
    int64_t bx = ((int64_t)H * (W / 4) + 255) / 256;
    const int64_t cap = (int64_t)pg::num_cu() * 8 / N + 1;
    if (bx > cap) bx = cap;
    const dim3 grid((unsigned)bx, (unsigned)N);
    if (act == PG_ACT_RELU) hipLaunchKernelGGL(conv3x3_cin1_kernel<true>, grid, dim3(256), 0, (hipStream_t)stream, x, w, y, H, W, Cout, scale);
    else hipLaunchKernelGGL(conv3x3_cin1_kernel<false>, grid, dim3(256), 0, (hipStream_t)stream, x, w, y, H, W, Cout, scale);
    return pg::launch_status();
}

PG_EXPORT int pg_modconv_dcoefs(const float* w, const float* styles, float* dcoefs,
                                int N, int Cout, int Cin, int KHW, float scale, void* stream) {
    if (!w || !styles || !dcoefs || N <= 0 || Cout <= 0 || Cin <= 0 || KHW <= 0) return PG_ERR_INVALID_ARG;
    hipLaunchKernelGGL(dcoefs_kernel, dim3((unsigned)(N * Cout)), dim3(256), 0, (hipStream_t)stream, w, styles, dcoefs, Cout, Cin, KHW, scale);
    return pg::launch_status();
}

PG_EXPORT int pg_modconv_w2(const float* w, float* w2, int Cout, int Cin, int KHW, float scale, void* stream) {
    if (!w || !w2 || Cout <= 0 || Cin <= 0 || KHW <= 0) return PG_ERR_INVALID_ARG;
    const int64_t total = (int64_t)Cout * Cin;
    hipLaunchKernelGGL(modconv_w2_kernel, dim3((unsigned)((total + 255) / 256)), dim3(256), 0, (hipStream_t)stream, w, w2, total, KHW, scale * scale);
    return pg::launch_status();
}

PG_EXPORT int pg_modconv_prep(const float* w2, const float* styles, float* out, float* s_norm, void* s16, int half_dtype,
                              int N, int Cout, int Cin, int normalize, int demodulate, void* stream) {
    if (!styles || !out || N <= 0 || Cout <= 0 || Cin <= 0 || (demodulate && !w2)) return PG_ERR_INVALID_ARG;
    if (s16 && half_dtype != PG_BF16 && half_dtype != PG_F16) return PG_ERR_INVALID_ARG;
    if (N > 65535 || (size_t)Cin * 4 > 64 * 1024) return PG_ERR_TOO_LARGE;
    hipLaunchKernelGGL(modconv_prep_kernel, dim3((unsigned)((Cout + 15) / 16), (unsigned)N), dim3(256), (size_t)Cin * 4, (hipStream_t)stream,
                       w2, styles, out, s_norm, (unsigned short*)s16, half_dtype, Cout, Cin, normalize, demodulate);
    return pg::launch_status();
}

PG_EXPORT int pg_modconv_prep_batched(const pg_modconv_prep_jobs* jobs, int N, void* stream) {
    if (!jobs || N <= 0 || N > 65535 || jobs->njobs <= 0 || jobs->njobs > PG_MODCONV_PREP_MAX_JOBS) return PG_ERR_INVALID_ARG;
    if (jobs->half_dtype != 0 && jobs->half_dtype != PG_BF16 && jobs->half_dtype != PG_F16) return PG_ERR_INVALID_ARG;
    int max_cout = 0, max_cin = 0;
    for (int j = 0; j < jobs->njobs; j++) {
        const int dem = (jobs->flags[j] >> 1) & 1;
        if (!jobs->styles[j] || !jobs->out[j] || jobs->cout[j] <= 0 || jobs->cin[j] <= 0 || (dem && !jobs->w2[j])) return PG_ERR_INVALID_ARG;
        if (jobs->s16[j] && !jobs->half_dtype) return PG_ERR_INVALID_ARG;
        if (jobs->cout[j] > max_cout) max_cout = jobs->cout[j];
        if (jobs->cin[j] > max_cin) max_cin = jobs->cin[j];
    }
    if ((size_t)max_cin * 4 > 64 * 1024) return PG_ERR_TOO_LARGE;
    hipLaunchKernelGGL(modconv_prep_batched_kernel, dim3((unsigned)((max_cout + 15) / 16), (unsigned)N, (unsigned)jobs->njobs), dim3(256), (size_t)max_cin * 4,
                       (hipStream_t)stream, *jobs);
    return pg::launch_status();
}

PG_EXPORT int pg_instance_norm_stats(const float* x, float* mean, float* rstd, int NC, int64_t HW, float eps, void* stream) {
    if (!x || !mean || !rstd || NC <= 0 || HW <= 0) return PG_ERR_INVALID_ARG;
    hipLaunchKernelGGL(instance_norm_stats_kernel, dim3((unsigned)NC), dim3(IN_THREADS), 0, (hipStream_t)stream, x, mean, rstd, HW, eps);
    return pg::launch_status();
}

// mean / rstd from the per-tile (sum, M2) pairs the F(4x4) kernel's plain tail wrote (pg_conv2d_fusion::stats_partial; M2 = squared deviations from the
// tile's own mean): one wave per (n, c) plane, every lane merges its tiles in tile order with Chan's pairwise formula in float64, then a fixed-shape wave
// reduction with the same merge -- deterministic, and as insensitive to |mean| >> std as the two-pass kernel above.  A tile's count follows from the geometry.
__global__ __launch_bounds__(64) void instance_norm_finish_kernel(const float* __restrict__ part, float* __restrict__ mean, float* __restrict__ rstd, int T, int OH, int OW, float eps) {
    const float* pp = part + (int64_t)blockIdx.x * T * 2;
    const int tiles_x = (OW + 63) / 64;
    double n = 0.0, s = 0.0, q = 0.0;
    auto merge = [](double& na, double& sa, double& qa, double nb, double sb, double qb) {
        if (nb <= 0.0) return;
        if (na <= 0.0) { na = nb; sa = sb; qa = qb; return; }
        const double d = sb / nb - sa / na;
        qa += qb + d * d * na * nb / (na + nb);
        sa += sb; na += nb;
    };
    for (int t = threadIdx.x; t < T; t += 64) {
        const int ty = t / tiles_x, tx = t - ty * tiles_x;
        const int rows = OH - ty * 8 < 8 ? OH - ty * 8 : 8, cols = OW - tx * 64 < 64 ? OW - tx * 64 : 64;
        merge(n, s, q, (double)rows * cols, (double)pp[2 * t], (double)pp[2 * t + 1]);
    }
#pragma unroll
    for (int m = 32; m >= 1; m >>= 1) {
        const double nb = __shfl_xor(n, m, 64), sb = __shfl_xor(s, m, 64), qb = __shfl_xor(q, m, 64);
        // (both partners compute the same merged set: symmetric up to the order of the two operands, which the lane id fixes)
        if (threadIdx.x & m) { double na = nb, sa = sb, qa = qb; merge(na, sa, qa, n, s, q); n = na; s = sa; q = qa; }
        else merge(n, s, q, nb, sb, qb);
    }
    if (threadIdx.x == 0) {
        const double mu = n > 0.0 ? s / n : 0.0, var = n > 0.0 ? q / n : 0.0;
        mean[blockIdx.x] = (float)mu;
        rstd[blockIdx.x] = (float)(1.0 / sqrt(var + (double)eps));
    }
}

PG_EXPORT int pg_conv2d_winograd4_stats_tiles(int OH, int OW) {
    if (OH <= 0 || OW <= 0) return 0;
    return ((OH + 7) / 8) * ((OW + 63) / 64);
}

PG_EXPORT int pg_instance_norm_finish(const float* stats_partial, float* mean, float* rstd, int NC, int T, int OH, int OW, float eps, void* stream) {
    if (!stats_partial || !mean || !rstd || NC <= 0 || T <= 0 || OH <= 0 || OW <= 0 || T != pg_conv2d_winograd4_stats_tiles(OH, OW)) return PG_ERR_INVALID_ARG;
    hipLaunchKernelGGL(instance_norm_finish_kernel, dim3((unsigned)NC), dim3(64), 0, (hipStream_t)stream, stats_partial, mean, rstd, T, OH, OW, eps);
    return pg::launch_status();
}

PG_EXPORT int pg_spade_norm(const float* x, const float* mean, const float* rstd, const float* gamma, const float* beta,
                            float* y, int NC, int64_t HW, void* stream) {
    if (!x || !mean || !rstd || !gamma || !beta || !y || NC <= 0 || HW <= 0) return PG_ERR_INVALID_ARG;
    const int64_t total = (int64_t)NC * HW;
    const bool vec = HW % 4 == 0 && pg::aligned16(x) && pg::aligned16(gamma) && pg::aligned16(beta) && pg::aligned16(y);
    int64_t blocks = ((vec ? total / 4 : total) + 255) / 256;
    if (blocks > pg::max_stream_blocks()) blocks = pg::max_stream_blocks();
    if (vec)
        hipLaunchKernelGGL(spade_norm_kernel, dim3((unsigned)blocks), dim3(256), 0, (hipStream_t)stream, x, mean, rstd, gamma, beta, y, HW / 4, total / 4);
    else
        hipLaunchKernelGGL(spade_norm_scalar_kernel, dim3((unsigned)blocks), dim3(256), 0, (hipStream_t)stream, x, mean, rstd, gamma, beta, y, HW, total);
    return pg::launch_status();
}

PG_EXPORT int pg_spade_train_forward(const float* x, const float* mean, const float* rstd, const float* gamma, const float* beta, float* y,
                                     int N, int C, int64_t HW, int64_t gamma_sample_stride, int64_t beta_sample_stride, void* stream) {
    if (!x || !mean || !rstd || !gamma || !beta || !y || N <= 0 || C <= 0 || HW <= 0) return PG_ERR_INVALID_ARG;
    const int64_t total = (int64_t)N * C * HW;
    const bool vec = HW % 4 == 0 && gamma_sample_stride % 4 == 0 && beta_sample_stride % 4 == 0 && pg::aligned16(x) && pg::aligned16(gamma) && pg::aligned16(beta) && pg::aligned16(y);
    int64_t blocks = ((vec ? total / 4 : total) + 255) / 256;
    if (blocks > pg::max_stream_blocks()) blocks = pg::max_stream_blocks();
    if (vec)
        hipLaunchKernelGGL(spade_train_forward_kernel<4>, dim3((unsigned)blocks), dim3(256), 0, (hipStream_t)stream, x, mean, rstd, gamma, beta, y, C, HW / 4, total / 4,
                           gamma_sample_stride, beta_sample_stride);
    else
        hipLaunchKernelGGL(spade_train_forward_kernel<1>, dim3((unsigned)blocks), dim3(256), 0, (hipStream_t)stream, x, mean, rstd, gamma, beta, y, C, HW, total,
                           gamma_sample_stride, beta_sample_stride);
    return pg::launch_status();
}

PG_EXPORT int pg_spade_train_backward(const float* dy, const float* x, const float* mean, const float* rstd, const float* gamma, float* sums,
                                      float* dx, float* dgamma, float* dbeta, int N, int C, int64_t HW,
                                      int64_t gamma_sample_stride, int64_t dgamma_sample_stride, int64_t dbeta_sample_stride, void* stream) {
    if (!dy || !x || !mean || !rstd || !gamma || !sums || N <= 0 || C <= 0 || HW <= 0 || (!dgamma != !dbeta) || (!dx && !dgamma)) return PG_ERR_INVALID_ARG;
    const int64_t total = (int64_t)N * C * HW;
    const bool vec = HW % 4 == 0 && gamma_sample_stride % 4 == 0 && dgamma_sample_stride % 4 == 0 && dbeta_sample_stride % 4 == 0 && pg::aligned16(dy) && pg::aligned16(x) &&
                     pg::aligned16(gamma) && (!dx || pg::aligned16(dx)) && (!dgamma || (pg::aligned16(dgamma) && pg::aligned16(dbeta)));
    if (dx)
        hipLaunchKernelGGL(spade_train_sums_kernel, dim3((unsigned)(N * C)), dim3(IN_THREADS), 0, (hipStream_t)stream, dy, x, mean, rstd, gamma, sums, C, HW,
                           gamma_sample_stride, (int)vec);
    int64_t blocks = ((vec ? total / 4 : total) + 255) / 256;
    if (blocks > pg::max_stream_blocks()) blocks = pg::max_stream_blocks();
    if (vec)
        hipLaunchKernelGGL(spade_train_backward_kernel<4>, dim3((unsigned)blocks), dim3(256), 0, (hipStream_t)stream, dy, x, mean, rstd, gamma, (const float*)sums, dx, dgamma, dbeta,
                           C, HW / 4, total / 4, gamma_sample_stride, dgamma_sample_stride, dbeta_sample_stride);
    else
        hipLaunchKernelGGL(spade_train_backward_kernel<1>, dim3((unsigned)blocks), dim3(256), 0, (hipStream_t)stream, dy, x, mean, rstd, gamma, (const float*)sums, dx, dgamma, dbeta,
                           C, HW, total, gamma_sample_stride, dgamma_sample_stride, dbeta_sample_stride);
    return pg::launch_status();
}

PG_EXPORT int pg_spade_masked_sums(const float* feat, const float* mask, const float* denorm_mask, float* sums, float* counts,
                                   int N, int C, int H, int W, void* stream) {
    if (!feat || !mask || !denorm_mask || !sums || !counts || N <= 0 || C <= 0 || H <= 0 || W <= 0) return PG_ERR_INVALID_ARG;
    if ((int64_t)H * W > 0x3fffffffLL) return PG_ERR_TOO_LARGE;
    const int vec = (W % 4 == 0 && pg::aligned16(feat) && pg::aligned16(mask) && pg::aligned16(denorm_mask)) ? 1 : 0;
    static const int grp = [] { const char* e = getenv("PG_SPADE_SUMS_G"); return e ? atoi(e) : 0; }();      // dev A/B: planes per workgroup (default: 4 where that still gives every CU a workgroup, else 2)
    const int g = grp ? grp : ((int64_t)N * C / 4 >= pg::num_cu() ? 4 : 2);
    if (g == 4 && C % 4 == 0) hipLaunchKernelGGL(spade_masked_sums_kernel<4>, dim3((unsigned)(N * C / 4)), dim3(IN_THREADS), 0, (hipStream_t)stream, feat, mask, denorm_mask, sums, counts, C, H, W, vec);
    else if (g >= 2 && C % 2 == 0) hipLaunchKernelGGL(spade_masked_sums_kernel<2>, dim3((unsigned)(N * C / 2)), dim3(IN_THREADS), 0, (hipStream_t)stream, feat, mask, denorm_mask, sums, counts, C, H, W, vec);
    else hipLaunchKernelGGL(spade_masked_sums_kernel<1>, dim3((unsigned)(N * C)), dim3(IN_THREADS), 0, (hipStream_t)stream, feat, mask, denorm_mask, sums, counts, C, H, W, vec);
    return pg::launch_status();
}

PG_EXPORT int pg_spade_feat_assemble(const float* feat_upper, const float* feat_lower, const float* mask_upper, const float* mask_lower,
                                     const float* denorm_mask_upper, const float* denorm_mask_lower,
                                     const float* sums_upper, const float* sums_lower, const float* counts_upper, const float* counts_lower,
                                     float* out, int N, int C, int H, int W, void* stream) {
    if (!feat_upper || !feat_lower || !mask_upper || !mask_lower || !denorm_mask_upper || !denorm_mask_lower || !sums_upper || !sums_lower ||
        !counts_upper || !counts_lower || !out || N <= 0 || C <= 0 || H <= 0 || W <= 0) return PG_ERR_INVALID_ARG;
    if ((int64_t)H * W > 0x3fffffffLL) return PG_ERR_TOO_LARGE;
    const int vec = (W % 4 == 0 && pg::aligned16(feat_upper) && pg::aligned16(feat_lower) && pg::aligned16(out) && pg::aligned16(mask_upper) && pg::aligned16(mask_lower) &&
                     pg::aligned16(denorm_mask_upper) && pg::aligned16(denorm_mask_lower)) ? 1 : 0;
    int chunks = (int)(((int64_t)H * W + 256 * 16 - 1) / (256 * 16));
    if (chunks < 1) chunks = 1;
    if (chunks > 64) chunks = 64;
    if (vec) {              // one image per blockIdx.x, the channels inside: enough pixel chunks to fill the chip
        chunks = (int)(((int64_t)H * (W / 4) + 255) / 256);
        const int want = (8 * pg::num_cu() + N - 1) / N;
        if (chunks > want) chunks = want;
        if (chunks < 1) chunks = 1;
    }
    static const int zsplit = [] { const char* e = getenv("PG_SPADE_ASM_Z"); return e && atoi(e) > 0 ? atoi(e) : 2; }();      // channel groups per pixel chunk (the masks are re-read per group: cheap)
    const int gz = vec ? (C >= 4 * zsplit ? zsplit : 1) : 1;
    hipLaunchKernelGGL(spade_feat_assemble_kernel, dim3((unsigned)(vec ? N : N * C), (unsigned)chunks, (unsigned)gz), dim3(256), 0, (hipStream_t)stream,
                       feat_upper, feat_lower, mask_upper, mask_lower, denorm_mask_upper, denorm_mask_lower,
                       sums_upper, sums_lower, counts_upper, counts_lower, out, C, H, W, vec);
    return pg::launch_status();
}
