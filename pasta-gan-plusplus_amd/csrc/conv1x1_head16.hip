// Streaming 1x1 head of the 16-bit stack (ToRGB, networks.py:1957-1967), round 5: the lanes of a wave share a pixel's channels.
//
// The first form (conv1x1_small16_kernel, conv2d16.hip) gives a pixel to ONE lane: its 16-byte loads sit Cin * 2 bytes apart across the wave (every load
// instruction touches 64 cache lines for 1 KB of data), the modulated weights go through LDS, and the half-resolution skip image costs 12 scattered loads per
// pixel -- 3.5 TB/s on the 1024^2 image, and 12-15 us on the 8^2 ... 32^2 images, which are pure latency (a serial weight prologue, a dependent channel loop).
// Here:
//   * LP = Cin / 8 lanes (<= 64) hold one pixel's 16-byte channel groups, so a load instruction of the wave reads 64 / LP whole pixels: contiguous runs of
//     Cin * 2 bytes.  A lane keeps ITS 8 channels' modulated weights (COUT x 8 floats per group) in registers for the whole launch: no LDS, no prologue loop;
//   * a wave takes U pixels per lane group at a time (all U * KI loads in flight), multiplies, and folds the LP partial sums with a reduce-scatter over
//     xor-shuffles (LP - 1 per output instead of U * log2 LP): lane `sub` of a group ends up with pixel `sub`'s sums.  With U == LP that is pixel == lane: 64
//     consecutive pixels per wave, 256-byte stores per colour plane;
//   * the half-resolution skip image (upfirdn2d.upsample2d with [1, 3, 3, 1]: per axis 1/4, 3/4 of two neighbours) is read as ONE column per lane and row;
//     the other column is the neighbour lane's value (an even output column needs the column to its left = what lane - 1 loaded, an odd one the column to
//     its right = lane + 1's): v_mov_b32_dpp wave_shr / wave_shl, with the two edge lanes of the wave loading their own;
//   * small images get U < LP (down to one pixel per wave pass) so that every pixel's loads are in flight at once on as many CUs as there are pixels.
// HBM-bound: 2 * Cin bytes in + 4 * COUT out (+ COUT for the quarter-size skip) per pixel.
#include "conv2d_kernel16.h"
#include "pg_common.h"

namespace pgconv16 {
namespace {

template <int CTRL>
__device__ __forceinline__ float dpp_mov(float v) {
    return __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), CTRL, 0xf, 0xf, true));
}

template <typename T, int COUT, int LP, int U, int KI>
__global__ __launch_bounds__(256) void conv1x1_head16_kernel(const unsigned short* __restrict__ x, const float* __restrict__ w, const float* __restrict__ styles,
                                                             const float* __restrict__ bias, const float* __restrict__ skip, float* __restrict__ y,
                                                             int Cin, int HW, float clamp, int up_w) {
    constexpr int G = 64 / LP, PB = G * U;                           // lane groups per wave, pixels per wave pass
    static_assert(U <= LP && (U & (U - 1)) == 0 && (LP & (LP - 1)) == 0, "U, LP: powers of two, U <= LP");
    const int n = blockIdx.y;
    const int lane = threadIdx.x & 63, sub = lane % LP, g = lane / LP;
    const int groups = Cin >> 3;
    // this lane's modulated weights: channels 8 * (sub + LP * k) ... + 7 of every output
    float wr[KI][COUT][8];
    bool valid[KI];
#pragma unroll
    for (int k = 0; k < KI; k++) {
        const int grp = sub + LP * k;
        valid[k] = grp < groups;
        const int c0 = (valid[k] ? grp : 0) * 8;
        f32x4 s0 = {1.f, 1.f, 1.f, 1.f}, s1 = s0;
        if (styles) { s0 = *(const f32x4*)(styles + (int64_t)n * Cin + c0); s1 = *(const f32x4*)(styles + (int64_t)n * Cin + c0 + 4); }
#pragma unroll
        for (int o = 0; o < COUT; o++) {
            const f32x4 w0 = *(const f32x4*)(w + o * Cin + c0), w1 = *(const f32x4*)(w + o * Cin + c0 + 4);
#pragma unroll
            for (int d = 0; d < 4; d++) { wr[k][o][d] = valid[k] ? w0[d] * s0[d] : 0.f; wr[k][o][4 + d] = valid[k] ? w1[d] * s1[d] : 0.f; }
        }
    }
    float bo[COUT];
#pragma unroll
    for (int o = 0; o < COUT; o++) bo[o] = bias ? bias[o] : 0.f;
    const float cl = clamp >= 0.f ? clamp : __builtin_inff();
    const unsigned short* xn = x + (int64_t)n * HW * Cin;
    const int npass = (HW + PB - 1) / PB;
    const int wave = blockIdx.x * 4 + (threadIdx.x >> 6), nwaves = gridDim.x * 4;
    const int W = up_w, hw = W >> 1, hh = up_w ? (HW / W) >> 1 : 0;

    for (int b = wave; b < npass; b += nwaves) {
        const int base = b * PB + g * U;
        u32x4 v[U][KI];
#pragma unroll
        for (int u = 0; u < U; u++) {
            const int p = base + u;
            const unsigned short* px = xn + (int64_t)(p < HW ? p : 0) * Cin;
#pragma unroll
            for (int k = 0; k < KI; k++) {
                const u32x4 z = {0u, 0u, 0u, 0u};
                v[u][k] = valid[k] ? *(const u32x4*)(px + (sub + LP * k) * 8) : z;
            }
        }
        float acc[U][COUT];
#pragma unroll
        for (int u = 0; u < U; u++) {
#pragma unroll
            for (int o = 0; o < COUT; o++) acc[u][o] = 0.f;
#pragma unroll
            for (int k = 0; k < KI; k++) {
                float xv[8];
#pragma unroll
                for (int d = 0; d < 4; d++) {
                    xv[2 * d] = Half16<T>::widen((unsigned short)(v[u][k][d] & 0xffff));
                    xv[2 * d + 1] = Half16<T>::widen((unsigned short)(v[u][k][d] >> 16));
                }
#pragma unroll
                for (int o = 0; o < COUT; o++)
#pragma unroll
                    for (int d = 0; d < 8; d++) acc[u][o] = fmaf(xv[d], wr[k][o][d], acc[u][o]);
            }
        }
        // fold the LP partial sums of every pixel: butterflies while a stage is wider than the pixel set, then halving (lane keeps the pixels whose index
        // bit equals its own): acc[0][o] of lane `sub` = pixel base + (sub % U)
#pragma unroll
        for (int m = LP / 2; m >= 1; m >>= 1) {
            if (m >= U) {
#pragma unroll
                for (int u = 0; u < U; u++)
#pragma unroll
                    for (int o = 0; o < COUT; o++) acc[u][o] += __shfl_xor(acc[u][o], m, 64);
            } else {
                const bool hi = (sub & m) != 0;
#pragma unroll
                for (int i = 0; i < m; i++)
#pragma unroll
                    for (int o = 0; o < COUT; o++) {
                        const float keep = hi ? acc[i + m][o] : acc[i][o], send = hi ? acc[i][o] : acc[i + m][o];
                        acc[i][o] = keep + __shfl_xor(send, m, 64);
                    }
            }
        }
        const int p = base + (sub & (U - 1));
        const bool live = sub < U && p < HW;
        float r[COUT];
#pragma unroll
        for (int o = 0; o < COUT; o++) r[o] = fminf(fmaxf(acc[0][o] + bo[o], -cl), cl);
        if (skip && !up_w) {
            if (live) {
#pragma unroll
                for (int o = 0; o < COUT; o++) r[o] += skip[((int64_t)n * COUT + o) * HW + p];
            }
        } else if (skip) {
            // upfirdn2d.upsample2d(skip, [1, 3, 3, 1]) (zero insertion, padding (2, 1), gain 4): per axis an even output 2m takes x[m-1] / 4 + 3 x[m] / 4, an odd
            // one 3 x[m] / 4 + x[m+1] / 4, zeros outside the image
            const int pc = p < HW ? p : HW - 1;
            const int oy = pc / W, ox = pc - oy * W;
            const int my = oy >> 1, mx = ox >> 1;
            const int y0 = (oy & 1) ? my : my - 1;                   // rows y0 (weight wy0), y0 + 1
            const float wy0 = (oy & 1) ? 0.75f : 0.25f;
            const int xo = (ox & 1) ? mx + 1 : mx - 1;               // the other column (weight 1/4); the centre column mx has 3/4
            const bool xo_in = xo >= 0 && xo < hw;
            const float* sp = skip + (int64_t)n * COUT * hh * hw;
            float c[COUT][2], nb[COUT][2];
#pragma unroll
            for (int o = 0; o < COUT; o++)
#pragma unroll
                for (int rr = 0; rr < 2; rr++) {
                    const int yy = y0 + rr;
                    c[o][rr] = (yy >= 0 && yy < hh) ? sp[((int64_t)o * hh + yy) * hw + mx] : 0.f;
                }
            if constexpr (U == LP) {                                 // pixel == lane: the other column is what the neighbour lane holds as its centre
                const bool own = (lane == 0 && !(ox & 1)) || (lane == 63 && (ox & 1));
#pragma unroll
                for (int o = 0; o < COUT; o++)
#pragma unroll
                    for (int rr = 0; rr < 2; rr++) {
                        const float left = dpp_mov<0x138>(c[o][rr]), right = dpp_mov<0x130>(c[o][rr]);      // wave_shr:1 = from lane - 1, wave_shl:1 = from lane + 1
                        nb[o][rr] = (ox & 1) ? right : left;
                    }
                if (own) {
#pragma unroll
                    for (int o = 0; o < COUT; o++)
#pragma unroll
                        for (int rr = 0; rr < 2; rr++) {
                            const int yy = y0 + rr;
                            nb[o][rr] = (yy >= 0 && yy < hh && xo_in) ? sp[((int64_t)o * hh + yy) * hw + xo] : 0.f;
                        }
                }
            } else {
#pragma unroll
                for (int o = 0; o < COUT; o++)
#pragma unroll
                    for (int rr = 0; rr < 2; rr++) {
                        const int yy = y0 + rr;
                        nb[o][rr] = (yy >= 0 && yy < hh && xo_in) ? sp[((int64_t)o * hh + yy) * hw + xo] : 0.f;
                    }
            }
#pragma unroll
            for (int o = 0; o < COUT; o++) {
                const float n0 = xo_in ? nb[o][0] : 0.f, n1 = xo_in ? nb[o][1] : 0.f;
                r[o] += wy0 * (0.75f * c[o][0] + 0.25f * n0) + (1.f - wy0) * (0.75f * c[o][1] + 0.25f * n1);
            }
        }
        if (live) {
#pragma unroll
            for (int o = 0; o < COUT; o++) y[((int64_t)n * COUT + o) * HW + p] = r[o];
        }
    }
}

template <typename T, int COUT, int LP, int U, int KI>
int launch_one(const unsigned short* x, const float* w, const float* styles, const float* bias, const float* skip, float* y, int N, int Cin, int HW, float clamp,
               int up_w, hipStream_t s) {
    constexpr int PB = (64 / LP) * U;
    const int64_t npass = ((int64_t)HW + PB - 1) / PB;
    int64_t bx = (npass + 3) / 4;
    const int64_t cap = (int64_t)pg::max_stream_blocks() / N > 0 ? (int64_t)pg::max_stream_blocks() / N : 1;
    if (bx > cap) bx = cap;
    hipLaunchKernelGGL((conv1x1_head16_kernel<T, COUT, LP, U, KI>), dim3((unsigned)bx, (unsigned)N), dim3(256), 0, s, x, w, styles, bias, skip, y, Cin, HW, clamp, up_w);
    return pg::launch_status();
}

template <typename T, int LP, int KI>
int launch_u(int U, const unsigned short* x, const float* w, const float* styles, const float* bias, const float* skip, float* y, int N, int Cin, int HW, float clamp,
             int up_w, hipStream_t s) {
    constexpr int UF = LP < 16 / KI ? LP : 16 / KI;                  // at most 16 loads of 16 bytes in flight per lane
    if (U >= UF) return launch_one<T, 3, LP, UF, KI>(x, w, styles, bias, skip, y, N, Cin, HW, clamp, up_w, s);
    if constexpr (UF > 4) if (U >= 4) return launch_one<T, 3, LP, 4, KI>(x, w, styles, bias, skip, y, N, Cin, HW, clamp, up_w, s);
    return launch_one<T, 3, LP, 1, KI>(x, w, styles, bias, skip, y, N, Cin, HW, clamp, up_w, s);
}

template <typename T>
int launch_t(const unsigned short* x, const float* w, const float* styles, const float* bias, const float* skip, float* y, int N, int Cin, int HW, int Cout,
             float clamp, int up_w, hipStream_t s) {
    const int groups = Cin / 8;
    if (Cout != 3 || Cin % 8 != 0 || groups < 4 || groups > 128 || (int64_t)HW * Cin > 0x7fffffffLL) return PG_ERR_UNSUPPORTED;
    int LP = 4;
    while (LP * 2 <= groups && LP < 64) LP *= 2;
    const int KI = (groups + LP - 1) / LP;
    if (KI > 2 || (KI == 2 && LP != 64)) return PG_ERR_UNSUPPORTED;  // (power-of-two widths up to 512 channels, anything 520 ... 1024; the rest stays on the first form)
    // pixels per lane group and pass: as many as keep >= ~4 wave passes per CU in the launch
    int U = LP < 16 ? LP : 16;
    while (U > 1 && (int64_t)N * HW / ((64 / LP) * U) < 4 * (int64_t)pg::num_cu()) U >>= 1;
#define PG_HEAD_LP(L, K) if (LP == L && KI == K) return launch_u<T, L, K>(U, x, w, styles, bias, skip, y, N, Cin, HW, clamp, up_w, s);
    PG_HEAD_LP(4, 1) PG_HEAD_LP(8, 1) PG_HEAD_LP(16, 1) PG_HEAD_LP(32, 1) PG_HEAD_LP(64, 1) PG_HEAD_LP(64, 2)
#undef PG_HEAD_LP
    return PG_ERR_UNSUPPORTED;
}

}  // namespace

// PG_ERR_UNSUPPORTED = not this form's shape: the caller (pg_conv1x1_small16) runs the first form.
int launch_head16(const void* x, const float* w, const float* styles, const float* bias, const float* skip, float* y, int dtype, int N, int Cin, int64_t HW, int Cout,
                  float clamp, int up_w, hipStream_t s) {
    if (HW > 0x7fffffffLL || (((uintptr_t)w) & 15) != 0 || (((uintptr_t)styles) & 15) != 0) return PG_ERR_UNSUPPORTED;
    if (dtype == PG_BF16) return launch_t<bf16_t>((const unsigned short*)x, w, styles, bias, skip, y, N, Cin, (int)HW, Cout, clamp, up_w, s);
    if (dtype == PG_F16) return launch_t<f16_t>((const unsigned short*)x, w, styles, bias, skip, y, N, Cin, (int)HW, Cout, clamp, up_w, s);
    return PG_ERR_INVALID_ARG;
}

}  // namespace pgconv16
