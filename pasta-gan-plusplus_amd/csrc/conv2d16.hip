// C ABI of the 16-bit (bf16 / fp16) convolution + its support kernels: weight packing with the style modulation and
// demodulation folded in, the split-K finish pass and the streaming 1x1 head.  The MFMA kernel itself lives in
// conv2d_kernel16.h and is instantiated per geometry in conv2d16_inst_*.hip.
#include <stdlib.h>
#include "conv2d_kernel16.h"
#include "conv2d_up2f16.h"

namespace {

using namespace pg;
using pgconv16::Conv16Params;
using pgconv16::Half16;
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef unsigned u32x4 __attribute__((ext_vector_type(4)));
typedef unsigned u32x2 __attribute__((ext_vector_type(2)));

inline int round_up(int v, int m) { return (v + m - 1) / m * m; }

struct TapSel { int ny, nx; int ys[8], xs[8]; };

// packed[n][ci/16][tap][(ci/8)&1][co][ci&7] = T(w[co][ci][ky][kx] * scale * styles[n][ci] * dcoefs[n][co])
// One workgroup = one (sample, 16-channel block, 64-cout block): the weights of that block are read as 64 contiguous runs
// of 16 * KH * KW floats (OIHW; one cache-line-coalesced run per cout), transposed through LDS, and written as T * 2 rows of
// 64 x 16 bytes -- contiguous 1 KB segments of the packed layout.  Both sides of the transpose are coalesced.
template <typename T, int KK>                                          // KK = KH * KW of the source weight (compile time: no runtime division)
__device__ __forceinline__ void pack16_body(const float* __restrict__ w, unsigned short* __restrict__ out, int Cout, int Cin, int KW, const TapSel& sel,
                                            int CinP, int CoutP, float scale, int flip, int transpose_oi,
                                            const float* __restrict__ styles, const float* __restrict__ dcoefs, int dco_mod, int64_t per_sample,
                                            int nsamples, int64_t w_group_stride, int cb, int k16, int z) {
    constexpr int RUN = 16 * KK, PITCH = RUN + 1;
    __shared__ float tile[64 * PITCH];                                 // [64 couts][16 channels][KK] (+1 pad per cout row)
    const int n = z % nsamples;                                        // z = group * nsamples + sample; dcoefs row = [dco_mod] entries, indexed by cout % dco_mod
    w += (int64_t)(z / nsamples) * w_group_stride;
    const int KH = KK / KW, T_ = sel.ny * sel.nx;
    const int co0 = cb * 64, ci0 = k16 * 16;
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    // ---- load: OIHW keeps (ci, ky, kx) contiguous per cout; IOHW (transposed weights) keeps (co, ky, kx) contiguous per ci
    // (all global loads of a wave are issued before the first value is used: the loops below have compile-time trip counts and the
    // loads land in registers -- one exposed memory round trip per workgroup instead of one per cout / channel row)
    if (!transpose_oi) {
        constexpr int PER_ROW = (RUN + 63) / 64;
        float v[16][PER_ROW], dsc[16];
#pragma unroll
        for (int it = 0; it < 16; it++) {                              // one wave per cout: its 16 * KK floats are one contiguous run
            const int co = co0 + wave + 4 * it;
            const float* src = w + ((int64_t)(co < Cout ? co : 0) * Cin + ci0) * KK;
            dsc[it] = (co < Cout ? scale : 0.f) * ((dcoefs && co < Cout) ? dcoefs[(int64_t)n * dco_mod + co % dco_mod] : 1.f);
#pragma unroll
            for (int k = 0; k < PER_ROW; k++) {
                const int r = lane + 64 * k;
                v[it][k] = (r < RUN && ci0 + r / KK < Cin) ? src[r] : 0.f;
            }
        }
        float sty[PER_ROW];
#pragma unroll
        for (int k = 0; k < PER_ROW; k++) {
            const int r = lane + 64 * k, ci = ci0 + r / KK;
            sty[k] = (styles && r < RUN && ci < Cin) ? styles[(int64_t)n * Cin + ci] : 1.f;
        }
#pragma unroll
        for (int it = 0; it < 16; it++)
#pragma unroll
            for (int k = 0; k < PER_ROW; k++) {
                const int r = lane + 64 * k;
                if (r < RUN) tile[(wave + 4 * it) * PITCH + r] = v[it][k] * dsc[it] * sty[k];
            }
    } else {
        constexpr int PER_CH = KK;                                     // 64 * KK floats per channel = KK per lane
        float v[4][PER_CH], ssc[4];
#pragma unroll
        for (int it = 0; it < 4; it++) {                               // one wave per input channel: 64 couts x KK floats contiguous
            const int ci = ci0 + wave + 4 * it;
            ssc[it] = (ci < Cin ? scale : 0.f) * ((styles && ci < Cin) ? styles[(int64_t)n * Cin + ci] : 1.f);
            const float* src = w + ((int64_t)(ci < Cin ? ci : 0) * Cout + co0) * KK;
#pragma unroll
            for (int k = 0; k < PER_CH; k++) {
                const int e = lane + 64 * k;
                v[it][k] = (co0 + e / KK < Cout) ? src[e] : 0.f;
            }
        }
        float dco[PER_CH];
#pragma unroll
        for (int k = 0; k < PER_CH; k++) {
            const int col = (lane + 64 * k) / KK;
            dco[k] = (dcoefs && co0 + col < Cout) ? dcoefs[(int64_t)n * dco_mod + (co0 + col) % dco_mod] : 1.f;
        }
#pragma unroll
        for (int it = 0; it < 4; it++)
#pragma unroll
            for (int k = 0; k < PER_CH; k++) {
                const int e = lane + 64 * k, col = e / KK, kk = e % KK;
                tile[col * PITCH + (wave + 4 * it) * KK + kk] = v[it][k] * ssc[it] * dco[k];
            }
    }
    __syncthreads();
    // ---- store: thread -> (row = tap * 2 + h, cout): 16 bytes = 8 consecutive channels of one tap
    unsigned short* dst = out + (int64_t)z * per_sample + (int64_t)k16 * T_ * 2 * CoutP * 8;
    for (int e = threadIdx.x; e < T_ * 2 * 64; e += 256) {
        const int col = e & 63, row = e >> 6;
        const int h = row & 1, tap = row >> 1;
        const int ty = sel.nx == 1 ? tap : (sel.nx == 2 ? tap >> 1 : tap / 3), tx = tap - ty * sel.nx;
        int ky = sel.ys[ty], kx = sel.xs[tx];
        if (flip) { ky = KH - 1 - ky; kx = KW - 1 - kx; }
        const float* src = tile + col * PITCH + (h * 8) * KK + ky * KW + kx;
        u32x4 o;
#pragma unroll
        for (int d = 0; d < 4; d++) o[d] = Half16<T>::pack(src[(2 * d) * KK], src[(2 * d + 1) * KK]);
        *(u32x4*)(dst + ((int64_t)row * CoutP + co0 + col) * 8) = o;
    }
}

template <typename T, int KK>
__global__ __launch_bounds__(256) void pack16_kernel(const float* __restrict__ w, unsigned short* __restrict__ out, int Cout, int Cin, int KW, TapSel sel,
                                                     int CinP, int CoutP, float scale, int flip, int transpose_oi,
                                                     const float* __restrict__ styles, const float* __restrict__ dcoefs, int64_t per_sample,
                                                     int nsamples, int64_t w_group_stride) {
    pack16_body<T, KK>(w, out, Cout, Cin, KW, sel, CinP, CoutP, scale, flip, transpose_oi, styles, dcoefs, Cout, per_sample, nsamples, w_group_stride,
                       blockIdx.x, blockIdx.y, blockIdx.z);
}

// The per-sample 3x3 packs of every modulated convolution of a network in ONE launch (grid z = job * nsamples + sample; the job table travels by value).
template <typename T>
__global__ __launch_bounds__(256) void pack16_batched_kernel(pg_conv2d16_pack_jobs J) {
    const int j = blockIdx.z / J.nsamples, n = blockIdx.z % J.nsamples;
    const int Cout = J.cout[j], Cin = J.cin[j];
    const int CinP = (Cin + 31) / 32 * 32, CoutP = (Cout + 63) / 64 * 64;
    if ((int)blockIdx.x >= CoutP / 64 || (int)blockIdx.y >= CinP / 16) return;      // the grid is sized for the largest job
    TapSel sel;
    const bool k32 = (J.flags[j] >> 2) & 1;                            // a 3 x 2 kernel (the stack of pg_conv2d16_up2_fused) instead of 3 x 3
    sel.ny = 3; sel.nx = k32 ? 2 : 3;
#pragma unroll
    for (int i = 0; i < 8; i++) { sel.ys[i] = i < 3 ? i : 0; sel.xs[i] = i < sel.nx ? i : 0; }
    if (k32)
        pack16_body<T, 6>(J.w[j], (unsigned short*)J.packed[j], Cout, Cin, 2, sel, CinP, CoutP, J.scale[j], J.flags[j] & 1, (J.flags[j] >> 1) & 1,
                          J.styles[j], J.dcoefs[j], J.dcoefs_mod[j], (int64_t)CinP * 6 * CoutP, J.nsamples, 0, blockIdx.x, blockIdx.y, n);
    else
        pack16_body<T, 9>(J.w[j], (unsigned short*)J.packed[j], Cout, Cin, 3, sel, CinP, CoutP, J.scale[j], J.flags[j] & 1, (J.flags[j] >> 1) & 1,
                          J.styles[j], J.dcoefs[j], J.dcoefs_mod[j], (int64_t)CinP * 9 * CoutP, J.nsamples, 0, blockIdx.x, blockIdx.y, n);
}

template <typename T>
int launch_pack16(int KK, dim3 grid, hipStream_t s, const float* w, unsigned short* packed, int Cout, int Cin, int KW, const TapSel& sel, int CinP, int CoutP,
                  float scale, int flip, int transpose_oi, const float* styles, const float* dcoefs, int64_t per_sample, int nsamples, int64_t w_group_stride) {
#define PG_PACK(K) case K: hipLaunchKernelGGL((pack16_kernel<T, K>), grid, dim3(256), 0, s, w, packed, Cout, Cin, KW, sel, CinP, CoutP, scale, flip, transpose_oi, styles, dcoefs, per_sample, nsamples, w_group_stride); break;
    switch (KK) { PG_PACK(1) PG_PACK(2) PG_PACK(3) PG_PACK(4) PG_PACK(6) PG_PACK(9) default: return PG_ERR_UNSUPPORTED; }
#undef PG_PACK
    return pg::launch_status();
}

// y[n, co, oy*osy+ooy, ox*osx+oox] = T(epilogue(sum_z ws[z][n, oy, ox, co])), fixed order z = 0, 1, ... (deterministic).  The
// workspace is channels-last like the 16-bit activations, so threads that walk it in memory order also write y in order.
template <typename T>
__global__ __launch_bounds__(256) void splitk_finish16_kernel(const float* __restrict__ ws, void* __restrict__ y, int out_f32, int ksplit, int64_t slice,
                                                              int N, int Cout, int OH, int OW, int64_t ys0, int64_t ys1, int64_t ys2, int64_t ys3,
                                                              int osy, int osx, int ooy, int oox, pg_conv2d16_fusion f) {
    const int64_t total = (int64_t)N * Cout * OH * OW;
    const float slope = f.act == PG_ACT_LINEAR ? 1.f : (f.act == PG_ACT_RELU ? 0.f : f.alpha);
    const float cl = f.clamp >= 0.f ? f.clamp : __builtin_inff();
    for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < total; i += (int64_t)gridDim.x * 256) {
        const int cot = (int)(i % Cout), ox = (int)((i / Cout) % OW), oy = (int)((i / ((int64_t)Cout * OW)) % OH), n = (int)(i / ((int64_t)Cout * OW * OH));
        const int pc = f.phase_cout, ph = pc ? cot / pc : 0, co = cot - ph * pc, ce = pc ? pc : Cout;      // four-phase mode: pg_conv2d16_fusion
        float v = 0.f;
        for (int z = 0; z < ksplit; z++) v += ws[(int64_t)z * slice + i];
        if (f.out_scale) v *= f.out_scale[(int64_t)n * ce + co];
        if (f.noise) v += f.noise[n * f.noise_batch_stride + ph * f.noise_phase_stride + (int64_t)oy * OW + ox] * f.noise_gain;
        if (f.bias) v += f.bias[co];
        v = v > 0.f ? v : v * slope;
        v = fminf(fmaxf(v * f.gain, -cl), cl);
        const int64_t off = n * ys0 + co * ys1 + (int64_t)(oy * osy + (pc ? (ph >> 1) : ooy)) * ys2 + (int64_t)(ox * osx + (pc ? (ph & 1) : oox)) * ys3;
        if (out_f32) {
            if (f.residual) v += ((const float*)f.residual)[off];
            ((float*)y)[off] = v;
        } else {
            if (f.residual) v += Half16<T>::widen(((const unsigned short*)f.residual)[off]);
            ((unsigned short*)y)[off] = (unsigned short)(Half16<T>::pack(v, 0.f) & 0xffff);
        }
    }
}

// The same for 16-bit channels-last outputs (cout stride 1, Cout % 4 == 0): four consecutive couts per thread -- 16-byte workspace
// loads, 32-bit index arithmetic, one 8-byte store.
template <typename T>
__global__ __launch_bounds__(256) void splitk_finish16_vec4_kernel(const float* __restrict__ ws, unsigned short* __restrict__ y, int ksplit, int64_t slice,
                                                                   int total4, int Cout, int OH, int OW, int64_t ys0, int64_t ys2, int64_t ys3,
                                                                   int osy, int osx, int ooy, int oox, pg_conv2d16_fusion f) {
    const float slope = f.act == PG_ACT_LINEAR ? 1.f : (f.act == PG_ACT_RELU ? 0.f : f.alpha);
    const float cl = f.clamp >= 0.f ? f.clamp : __builtin_inff();
    const int Cout4 = Cout >> 2;
    for (int q = blockIdx.x * 256 + threadIdx.x; q < total4; q += gridDim.x * 256) {
        const int c4 = q % Cout4;
        int r = q / Cout4;
        const int ox = r % OW; r /= OW;
        const int oy = r % OH, n = r / OH;
        const int pc = f.phase_cout, ph = pc ? (4 * c4) / pc : 0, co = 4 * c4 - ph * pc, ce = pc ? pc : Cout;       // four-phase mode (pc % 4 == 0)
        f32x4 v = {0.f, 0.f, 0.f, 0.f};
        const float* src = ws + 4 * (int64_t)q;
        for (int z = 0; z < ksplit; z++) v += *(const f32x4*)(src + (int64_t)z * slice);
        if (f.out_scale) v *= *(const f32x4*)(f.out_scale + (int64_t)n * ce + co);
        if (f.noise) v += f.noise[n * f.noise_batch_stride + ph * f.noise_phase_stride + (int64_t)oy * OW + ox] * f.noise_gain;
        if (f.bias) { v[0] += f.bias[co]; v[1] += f.bias[co + 1]; v[2] += f.bias[co + 2]; v[3] += f.bias[co + 3]; }
#pragma unroll
        for (int j = 0; j < 4; j++) {
            const float a = v[j] > 0.f ? v[j] : v[j] * slope;
            v[j] = fminf(fmaxf(a * f.gain, -cl), cl);
        }
        const int64_t off = n * ys0 + co + (int64_t)(oy * osy + (pc ? (ph >> 1) : ooy)) * ys2 + (int64_t)(ox * osx + (pc ? (ph & 1) : oox)) * ys3;
        if (f.residual) {
            const u32x2 rr = *(const u32x2*)((const unsigned short*)f.residual + off);
            v[0] += Half16<T>::widen((unsigned short)(rr[0] & 0xffff)); v[1] += Half16<T>::widen((unsigned short)(rr[0] >> 16));
            v[2] += Half16<T>::widen((unsigned short)(rr[1] & 0xffff)); v[3] += Half16<T>::widen((unsigned short)(rr[1] >> 16));
        }
        const u32x2 o = {Half16<T>::pack(v[0], v[1]), Half16<T>::pack(v[2], v[3])};
        *(u32x2*)(y + off) = o;
    }
}

// Streaming 1x1 head: LP lanes per pixel (LP = 1 on large images: all of a pixel's channels in flight as 16-byte loads from one thread;
// LP = 4 / 16 on small ones, where one thread per pixel would leave most of the chip idle behind a Cin / 8-step dependent loop: the lanes of a
// pixel take the 16-byte channel groups round-robin and fold their sums with xor-shuffles), the (<= 8) x Cin modulated weights of the block's
// image in LDS (broadcast reads).  HBM-bound: 2*Cin bytes read + 4*Cout (+ 4*Cout skip) per pixel.
template <typename T, int COUT, int LP>
__global__ __launch_bounds__(256) void conv1x1_small16_kernel(const unsigned short* __restrict__ x, const float* __restrict__ w, const float* __restrict__ styles,
                                                              const float* __restrict__ bias, const float* __restrict__ skip, float* __restrict__ y,
                                                              int Cin, int64_t HW, float clamp, int up_w) {
    extern __shared__ __attribute__((aligned(16))) float wl[];       // [COUT][Cin]
    const int n = blockIdx.y;
    for (int e = threadIdx.x; e < COUT * Cin; e += 256) {
        const int c = e % Cin;
        wl[e] = w[e] * (styles ? styles[(int64_t)n * Cin + c] : 1.f);
    }
    __syncthreads();
    const float cl = clamp >= 0.f ? clamp : __builtin_inff();
    const unsigned short* xn = x + (int64_t)n * HW * Cin;
    const int sub = threadIdx.x % LP;
    constexpr int PPB = 256 / LP;                                     // pixels per block and pass
    const int64_t npass = (HW + (int64_t)gridDim.x * PPB - 1) / ((int64_t)gridDim.x * PPB);      // every lane runs every pass: the shuffles below need whole groups
    for (int64_t it = 0; it < npass; it++) {
        const int64_t p = (it * gridDim.x + blockIdx.x) * PPB + threadIdx.x / LP;
        const bool live = p < HW;
        const u32x4* px = (const u32x4*)(xn + (live ? p : 0) * Cin);
        float acc[COUT];
#pragma unroll
        for (int o = 0; o < COUT; o++) acc[o] = 0.f;
        for (int c8 = 4 * sub; c8 < Cin / 8; c8 += 4 * LP) {          // up to four 16-byte loads in flight per pass
            u32x4 v[4];
#pragma unroll
            for (int u = 0; u < 4; u++) if (c8 + u < Cin / 8) v[u] = px[c8 + u];
#pragma unroll
            for (int u = 0; u < 4; u++) {
                if (c8 + u >= Cin / 8) break;
                float xv[8];
#pragma unroll
                for (int d = 0; d < 4; d++) {
                    xv[2 * d] = Half16<T>::widen((unsigned short)(v[u][d] & 0xffff));
                    xv[2 * d + 1] = Half16<T>::widen((unsigned short)(v[u][d] >> 16));
                }
#pragma unroll
                for (int o = 0; o < COUT; o++) {
                    const f32x4 w0 = *(const f32x4*)(wl + o * Cin + (c8 + u) * 8), w1 = *(const f32x4*)(wl + o * Cin + (c8 + u) * 8 + 4);
#pragma unroll
                    for (int d = 0; d < 4; d++) acc[o] = fmaf(xv[d], w0[d], acc[o]);
#pragma unroll
                    for (int d = 0; d < 4; d++) acc[o] = fmaf(xv[4 + d], w1[d], acc[o]);
                }
            }
        }
        if (LP > 1) {
#pragma unroll
            for (int o = 0; o < COUT; o++)
#pragma unroll
                for (int m = LP / 2; m >= 1; m >>= 1) acc[o] += __shfl_xor(acc[o], m, 64);
        }
        if (live && sub == 0) {
#pragma unroll
            for (int o = 0; o < COUT; o++) {
                float v = acc[o] + (bias ? bias[o] : 0.f);
                v = fminf(fmaxf(v, -cl), cl);
                const int64_t off = ((int64_t)n * COUT + o) * HW + p;
                if (skip) {
                    if (!up_w) {
                        v += skip[off];
                    } else {
                        // skip = the HALF-resolution image, up-sampled here as upfirdn2d.upsample2d does with the [1, 3, 3, 1] filter (zero insertion, padding (2, 1),
                        // gain 4): per axis an even output 2m takes x[m-1] / 4 + 3 x[m] / 4, an odd one 3 x[m] / 4 + x[m+1] / 4, zeros outside the image
                        const int W = up_w, H = (int)(HW / W), hw = W >> 1, hh = H >> 1;
                        const int oy = (int)(p / W), ox = (int)(p - (int64_t)oy * W);
                        const int my = oy >> 1, mx = ox >> 1;
                        const int y0 = (oy & 1) ? my : my - 1, x0 = (ox & 1) ? mx : mx - 1;          // the first of the two source rows / columns
                        const float wy0 = (oy & 1) ? 0.75f : 0.25f, wx0 = (ox & 1) ? 0.75f : 0.25f;
                        const float* sp = skip + ((int64_t)n * COUT + o) * hh * hw;
                        auto at = [&](int yy, int xx) { return (yy >= 0 && yy < hh && xx >= 0 && xx < hw) ? sp[(int64_t)yy * hw + xx] : 0.f; };
                        v += wy0 * (wx0 * at(y0, x0) + (1.f - wx0) * at(y0, x0 + 1)) + (1.f - wy0) * (wx0 * at(y0 + 1, x0) + (1.f - wx0) * at(y0 + 1, x0 + 1));
                    }
                }
                y[off] = v;
            }
        }
    }
}

template <typename T, int LP>
int launch_small(int cout, dim3 grid, size_t lds, hipStream_t s, const unsigned short* x, const float* w, const float* styles, const float* bias,
                 const float* skip, float* y, int Cin, int64_t HW, float clamp, int up_w) {
#define PG_SMALL(C) case C: hipLaunchKernelGGL((conv1x1_small16_kernel<T, C, LP>), grid, dim3(256), lds, s, x, w, styles, bias, skip, y, Cin, HW, clamp, up_w); break;
    switch (cout) { PG_SMALL(1) PG_SMALL(2) PG_SMALL(3) PG_SMALL(4) PG_SMALL(5) PG_SMALL(6) PG_SMALL(7) PG_SMALL(8) default: return PG_ERR_UNSUPPORTED; }
#undef PG_SMALL
    return pg::launch_status();
}

unsigned long long* g_debug_stamps = nullptr;     // dev hook, see pg_conv2d16_debug_stamps

int kc_for16(int kh, int kw) { return kh * kw <= 2 ? 32 : 16; }

bool geometry_ok(int KH, int KW, int stride) {
    return (stride == 1 && ((KH == 3 && KW == 3) || (KH == 1 && KW == 1) || (KH == 2 && KW == 2) || (KH == 2 && KW == 1) || (KH == 1 && KW == 2))) ||
           (stride == 2 && KH == 3 && KW == 3);
}

int conv16_forward(const void* x, const void* packed, void* y, int dtype, int out_dtype,
                   int N, int Cin, int H, int W, int Cout, int KH, int KW,
                   int stride, int pad_y, int pad_x, int OH, int OW, int64_t w_sample_stride,
                   const int64_t ystride[4], int osy, int osx, int ooy, int oox,
                   const pg_conv2d16_fusion* fusion, void* stream, float* workspace, int ksplit) {
    if (!x || !packed || !y || !ystride) return PG_ERR_INVALID_ARG;
    if (dtype != PG_BF16 && dtype != PG_F16) return PG_ERR_INVALID_ARG;
    if (out_dtype != dtype && out_dtype != PG_F32) return PG_ERR_INVALID_ARG;
    if (N <= 0 || Cin <= 0 || H <= 0 || W <= 0 || Cout <= 0 || OH <= 0 || OW <= 0 || osy < 1 || osx < 1 || w_sample_stride < 0) return PG_ERR_INVALID_ARG;
    if (!geometry_ok(KH, KW, stride)) return PG_ERR_UNSUPPORTED;
    if (Cin % 16 != 0 || (((uintptr_t)x) & 15) != 0 || (((uintptr_t)packed) & 15) != 0) return PG_ERR_UNSUPPORTED;   // 16-byte channel vectors
    // one image of x and the whole weight buffer are addressed through 32-bit buffer descriptors
    if ((int64_t)H * W * Cin * 2 > 0x7fffffffLL) return PG_ERR_TOO_LARGE;
    const int T_ = KH * KW;
    const int CoutP = round_up(Cout, 64), CinP = round_up(Cin, 32);
    const int64_t per_sample = (int64_t)CinP * T_ * CoutP;
    if (w_sample_stride != 0 && w_sample_stride != per_sample) return PG_ERR_INVALID_ARG;
    const int64_t w_bytes = (w_sample_stride ? (int64_t)N * per_sample : per_sample) * 2;
    if (w_bytes > 0x7fffffffLL) return PG_ERR_TOO_LARGE;
    const int pcout = fusion ? fusion->phase_cout : 0;                  // four-phase mode: see pg_conv2d16_fusion
    if (pcout) {
        if (pcout < 0 || Cout != 4 * pcout || osy < 2 || osx < 2 || fusion->residual || fusion->noise_phase_stride < 0) return PG_ERR_INVALID_ARG;
        if (pcout != 32 && pcout % 64 != 0) return PG_ERR_UNSUPPORTED;
    }
    const int c_ext = pcout ? pcout : Cout, oy_ext = pcout ? 1 : ooy, ox_ext = pcout ? 1 : oox;
    const int64_t ext = 1 + (int64_t)(N - 1) * ystride[0] + (int64_t)(c_ext - 1) * ystride[1] +
                        ((int64_t)(OH - 1) * osy + oy_ext) * ystride[2] + ((int64_t)(OW - 1) * osx + ox_ext) * ystride[3];
    if (ext > 0x3fffffffLL) return PG_ERR_TOO_LARGE;

    Conv16Params p;
    p.x = x; p.wp = packed; p.y = y;
    p.w_nstride = w_sample_stride; p.w_bytes = w_bytes;
    p.y_bytes = ext * (out_dtype == PG_F32 ? 4 : 2);
    p.N = N; p.Cin = Cin; p.xC = Cin; p.H = H; p.W = W; p.Cout = Cout; p.CoutP = CoutP; p.OH = OH; p.OW = OW;
    p.pad_y = pad_y; p.pad_x = pad_x;
    p.ksplit = 1; p.kpart = 0; p.ws_slice = 0;
    { const char* e = getenv("PG_CONV16_DBG"); p.dbg = e ? atoi(e) : 0; p.stamps = g_debug_stamps; if (!p.stamps) p.dbg &= ~32; }
    for (int i = 0; i < 4; i++) p.ys[i] = ystride[i];
    p.osy = osy; p.osx = osx; p.ooy = ooy; p.oox = oox;
    if (fusion) {
        p.f = *fusion;
    } else {
        pg_conv2d16_fusion z = {};
        z.clamp = -1.f;
        p.f = z;
    }
    if (p.f.gain == 0.f) p.f.gain = 1.f;
    if (!(p.f.gain > 0.f)) return PG_ERR_UNSUPPORTED;                   // the epilogue folds the gain through the activation
    if (p.f.act == 0) p.f.act = PG_ACT_LINEAR;
    if (p.f.act < PG_ACT_LINEAR || p.f.act > PG_ACT_SWISH) return PG_ERR_INVALID_ARG;
    if (p.f.act > PG_ACT_LRELU) return PG_ERR_UNSUPPORTED;
    if (p.f.act == PG_ACT_LRELU && (p.f.alpha < 0.f || p.f.alpha > 1.f)) return PG_ERR_UNSUPPORTED;
    if (out_dtype == PG_F32) p.out_mode = pgconv16::OUT_SCALAR32;
    else p.out_mode = (ystride[1] == 1 && c_ext % 8 == 0 && ystride[3] % 8 == 0 && ystride[2] % 8 == 0 && ystride[0] % 8 == 0 &&
                       (((uintptr_t)y) & 15) == 0 && (!p.f.residual || (((uintptr_t)p.f.residual) & 7) == 0)) ? pgconv16::OUT_VEC16 : pgconv16::OUT_SCALAR16;
    hipStream_t s = (hipStream_t)stream;

    const pg_conv2d16_fusion tail = p.f;
    const int64_t slice = (int64_t)N * Cout * OH * OW;
    if (ksplit > 1) {
        const int kc = kc_for16(KH, KW);
        if (!workspace || Cin % (ksplit * kc) != 0 || slice * ksplit > 0x1fffffffLL) return PG_ERR_INVALID_ARG;      // float32 workspace addressed with 32-bit byte offsets
        p.ksplit = ksplit; p.kpart = Cin / ksplit; p.ws_slice = slice;
        p.y = workspace; p.y_bytes = slice * ksplit * 4;
        p.ys[0] = (int64_t)Cout * OH * OW; p.ys[1] = 1; p.ys[2] = (int64_t)OW * Cout; p.ys[3] = Cout;        // channels-last slices
        p.osy = p.osx = 1; p.ooy = p.oox = 0;
        p.out_mode = (Cout % 4 == 0 && (((uintptr_t)workspace) & 15) == 0) ? pgconv16::OUT_VEC32 : pgconv16::OUT_SCALAR32;
        pg_conv2d16_fusion z = {};
        z.clamp = -1.f; z.gain = 1.f; z.act = PG_ACT_LINEAR;
        p.f = z;
    }
    int st = PG_ERR_UNSUPPORTED;
    if (stride == 1) {
        if (KH == 3 && KW == 3) st = pgconv16::launch16_k3s1(p, dtype, s);
        else if (KH == 1 && KW == 1) st = pgconv16::launch16_k1s1(p, dtype, s);
        else if (KH == 2 && KW == 2) st = pgconv16::launch16_k2x2(p, dtype, s);
        else if (KH == 2 && KW == 1) st = pgconv16::launch16_k2x1(p, dtype, s);
        else if (KH == 1 && KW == 2) st = pgconv16::launch16_k1x2(p, dtype, s);
    } else if (KH == 3 && KW == 3) {
        st = pgconv16::launch16_k3s2(p, dtype, s);
    }
    if (st != PG_OK || ksplit <= 1) return st;
    const bool vec4 = out_dtype != PG_F32 && Cout % 4 == 0 && c_ext % 4 == 0 && ystride[1] == 1 && ((ystride[0] | ystride[2] | ystride[3]) & 3) == 0 && (((uintptr_t)y) & 7) == 0 &&
                      (((uintptr_t)workspace) & 15) == 0 && (!tail.out_scale || (((uintptr_t)tail.out_scale) & 15) == 0) &&
                      (!tail.residual || (((uintptr_t)tail.residual) & 7) == 0);
    if (vec4) {
        int64_t blocks4 = (slice / 4 + 255) / 256;
        if (blocks4 > pg::max_stream_blocks()) blocks4 = pg::max_stream_blocks();
        if (dtype == PG_BF16)
            hipLaunchKernelGGL((splitk_finish16_vec4_kernel<bf16_t>), dim3((unsigned)blocks4), dim3(256), 0, s, workspace, (unsigned short*)y, ksplit, slice, (int)(slice / 4), Cout, OH, OW,
                               ystride[0], ystride[2], ystride[3], osy, osx, ooy, oox, tail);
        else
            hipLaunchKernelGGL((splitk_finish16_vec4_kernel<f16_t>), dim3((unsigned)blocks4), dim3(256), 0, s, workspace, (unsigned short*)y, ksplit, slice, (int)(slice / 4), Cout, OH, OW,
                               ystride[0], ystride[2], ystride[3], osy, osx, ooy, oox, tail);
        return pg::launch_status();
    }
    int64_t blocks = (slice + 255) / 256;
    if (blocks > pg::max_stream_blocks()) blocks = pg::max_stream_blocks();
    if (dtype == PG_BF16)
        hipLaunchKernelGGL((splitk_finish16_kernel<bf16_t>), dim3((unsigned)blocks), dim3(256), 0, s, workspace, y, out_dtype == PG_F32, ksplit, slice, N, Cout, OH, OW,
                           ystride[0], ystride[1], ystride[2], ystride[3], osy, osx, ooy, oox, tail);
    else
        hipLaunchKernelGGL((splitk_finish16_kernel<f16_t>), dim3((unsigned)blocks), dim3(256), 0, s, workspace, y, out_dtype == PG_F32, ksplit, slice, N, Cout, OH, OW,
                           ystride[0], ystride[1], ystride[2], ystride[3], osy, osx, ooy, oox, tail);
    return pg::launch_status();
}

}  // namespace

PG_EXPORT int64_t pg_conv2d16_packed_size(int Cout, int Cin, int KH, int KW) {
    if (Cout <= 0 || Cin <= 0 || KH <= 0 || KW <= 0) return 0;
    return (int64_t)round_up(Cin, 32) * KH * KW * round_up(Cout, 64);
}

static int pack_weight16(const float* w, void* packed, int dtype, int ngroups, int64_t w_group_stride, int Cout, int Cin, int KH, int KW,
                         const int* taps_y, int ntaps_y, const int* taps_x, int ntaps_x,
                         float scale, int flip_hw, int transpose_oi,
                         const float* styles, const float* dcoefs, int nsamples, void* stream) {
    if (!w || !packed || Cout <= 0 || Cin <= 0 || KH <= 0 || KW <= 0 || nsamples <= 0 || ngroups <= 0 || w_group_stride < 0) return PG_ERR_INVALID_ARG;
    if (dtype != PG_BF16 && dtype != PG_F16) return PG_ERR_INVALID_ARG;
    TapSel sel;
    sel.ny = taps_y ? ntaps_y : KH;
    sel.nx = taps_x ? ntaps_x : KW;
    if (sel.ny <= 0 || sel.nx <= 0 || sel.ny > 8 || sel.nx > 8) return PG_ERR_UNSUPPORTED;
    for (int i = 0; i < 8; i++) {
        sel.ys[i] = i < sel.ny ? (taps_y ? taps_y[i] : i) : 0;
        sel.xs[i] = i < sel.nx ? (taps_x ? taps_x[i] : i) : 0;
        if (sel.ys[i] < 0 || sel.ys[i] >= KH || sel.xs[i] < 0 || sel.xs[i] >= KW) return PG_ERR_INVALID_ARG;
    }
    const int CinP = round_up(Cin, 32), CoutP = round_up(Cout, 64);
    const int64_t per_sample = (int64_t)CinP * sel.ny * sel.nx * CoutP;
    if (CinP / 16 > 65535 || (int64_t)nsamples * ngroups > 65535) return PG_ERR_TOO_LARGE;
    const dim3 grid((unsigned)(CoutP / 64), (unsigned)(CinP / 16), (unsigned)(nsamples * ngroups));
    if (sel.nx > 3) return PG_ERR_UNSUPPORTED;
    if (dtype == PG_BF16)
        return launch_pack16<bf16_t>(KH * KW, grid, (hipStream_t)stream, w, (unsigned short*)packed, Cout, Cin, KW, sel, CinP, CoutP, scale, flip_hw, transpose_oi,
                                     styles, dcoefs, per_sample, nsamples, w_group_stride);
    return launch_pack16<f16_t>(KH * KW, grid, (hipStream_t)stream, w, (unsigned short*)packed, Cout, Cin, KW, sel, CinP, CoutP, scale, flip_hw, transpose_oi,
                                styles, dcoefs, per_sample, nsamples, w_group_stride);
}

PG_EXPORT int pg_conv2d16_pack_weight(const float* w, void* packed, int dtype, int Cout, int Cin, int KH, int KW,
                                      const int* taps_y, int ntaps_y, const int* taps_x, int ntaps_x,
                                      float scale, int flip_hw, int transpose_oi,
                                      const float* styles, const float* dcoefs, int nsamples, void* stream) {
    return pack_weight16(w, packed, dtype, 1, 0, Cout, Cin, KH, KW, taps_y, ntaps_y, taps_x, ntaps_x, scale, flip_hw, transpose_oi, styles, dcoefs, nsamples, stream);
}

PG_EXPORT int pg_conv2d16_pack_weight_grouped(const float* w, void* packed, int dtype, int ngroups, int64_t w_group_stride,
                                              int Cout, int Cin, int KH, int KW, float scale, int flip_hw, int transpose_oi,
                                              const float* styles, const float* dcoefs, int nsamples, void* stream) {
    return pack_weight16(w, packed, dtype, ngroups, w_group_stride, Cout, Cin, KH, KW, nullptr, 0, nullptr, 0, scale, flip_hw, transpose_oi, styles, dcoefs, nsamples, stream);
}

PG_EXPORT int pg_conv2d16_pack_weight_batched(const pg_conv2d16_pack_jobs* jobs, void* stream) {
    if (!jobs || jobs->njobs <= 0 || jobs->njobs > PG_CONV2D16_PACK_MAX_JOBS || jobs->nsamples <= 0) return PG_ERR_INVALID_ARG;
    if (jobs->dtype != PG_BF16 && jobs->dtype != PG_F16) return PG_ERR_INVALID_ARG;
    int gx = 0, gy = 0;
    for (int j = 0; j < jobs->njobs; j++) {
        if (!jobs->w[j] || !jobs->packed[j] || jobs->cout[j] <= 0 || jobs->cin[j] <= 0) return PG_ERR_INVALID_ARG;
        if (jobs->dcoefs[j] && (jobs->dcoefs_mod[j] <= 0 || jobs->cout[j] % jobs->dcoefs_mod[j] != 0)) return PG_ERR_INVALID_ARG;
        const int cx = round_up(jobs->cout[j], 64) / 64, cy = round_up(jobs->cin[j], 32) / 16;
        if (cx > gx) gx = cx;
        if (cy > gy) gy = cy;
    }
    if (gy > 65535 || (int64_t)jobs->njobs * jobs->nsamples > 65535) return PG_ERR_TOO_LARGE;
    const dim3 grid((unsigned)gx, (unsigned)gy, (unsigned)(jobs->njobs * jobs->nsamples));
    if (jobs->dtype == PG_BF16) hipLaunchKernelGGL((pack16_batched_kernel<bf16_t>), grid, dim3(256), 0, (hipStream_t)stream, *jobs);
    else hipLaunchKernelGGL((pack16_batched_kernel<f16_t>), grid, dim3(256), 0, (hipStream_t)stream, *jobs);
    return pg::launch_status();
}

// Dev hook (not part of include/pasta_gan_ops.h): device buffer of >= 4000 uint64 that PG_CONV16_DBG=32 fills with
// (s_memtime << 8 | tag) stamps of workgroup 0 / wave (dbg >> 8); tools/conv16_stamps.py reads it.
PG_EXPORT void pg_conv2d16_debug_stamps(void* buf) { g_debug_stamps = (unsigned long long*)buf; }

PG_EXPORT int pg_conv2d16_forward(const void* x, const void* packed, void* y, int dtype, int out_dtype,
                                  int N, int Cin, int H, int W, int Cout, int KH, int KW,
                                  int stride, int pad_y, int pad_x, int OH, int OW, int64_t w_sample_stride,
                                  const int64_t ystride[4], int out_step_y, int out_step_x, int out_off_y, int out_off_x,
                                  const pg_conv2d16_fusion* fusion, void* stream) {
    return conv16_forward(x, packed, y, dtype, out_dtype, N, Cin, H, W, Cout, KH, KW, stride, pad_y, pad_x, OH, OW, w_sample_stride,
                          ystride, out_step_y, out_step_x, out_off_y, out_off_x, fusion, stream, nullptr, 1);
}

PG_EXPORT int pg_conv2d16_splitk_plan(int N, int Cin, int OH, int OW, int Cout, int KH, int KW, int stride) {
    if (N <= 0 || Cin <= 0 || OH <= 0 || OW <= 0 || Cout <= 0 || !geometry_ok(KH, KW, stride)) return 1;
    const int kc = kc_for16(KH, KW);
    if (Cin % kc != 0) return 1;
    const int nchunks = Cin / kc;
    int th = stride == 2 ? 8 : 16, tw = 32, bm = Cout <= 32 ? 32 : 64; // the workgroup tile (conv2d16_inst_*.hip, launch16_mt)
    const int sm = pgconv16::small_tile16(KH, KW, stride, OH, OW, Cout);
    if (sm == 1) { th = 8; tw = 8; bm = 128; }
    if (sm == 2) { th = 16; tw = 16; bm = 64; }
    const int64_t tiles = (int64_t)N * ((OW + tw - 1) / tw) * ((OH + th - 1) / th) * ((Cout + bm - 1) / bm);
    if (tiles >= pg::num_cu() || nchunks < 8) return 1;
    int best = 1;                                                      // largest divisor of nchunks that keeps >= 4 chunks per share and
    static const int per_cu = [] { const char* e = getenv("PG_CONV16_SPLITK_PER_CU"); return e && atoi(e) > 0 ? atoi(e) : 1; }();
    for (int k = 2; k <= 16; k++)                                      // does not overshoot one workgroup per CU (round 5, tools/small16_probe.py at N = 4: 1024 -> 1024 at 8^2
        if (nchunks % k == 0 && nchunks / k >= 4 && tiles * k <= per_cu * pg::num_cu()) best = k;      // 28 -> 24 us, at 16^2 39 -> 32, 1024 -> 4096 at 8^2 56 -> 54; until then: two per CU)
    return best;
}

PG_EXPORT int pg_conv2d16_forward_splitk(const void* x, const void* packed, void* y, int dtype, int out_dtype,
                                         int N, int Cin, int H, int W, int Cout, int KH, int KW,
                                         int stride, int pad_y, int pad_x, int OH, int OW, int64_t w_sample_stride,
                                         const int64_t ystride[4], int out_step_y, int out_step_x, int out_off_y, int out_off_x,
                                         const pg_conv2d16_fusion* fusion, float* workspace, int ksplit, void* stream) {
    if (ksplit < 1 || (ksplit > 1 && !workspace)) return PG_ERR_INVALID_ARG;
    return conv16_forward(x, packed, y, dtype, out_dtype, N, Cin, H, W, Cout, KH, KW, stride, pad_y, pad_x, OH, OW, w_sample_stride,
                          ystride, out_step_y, out_step_x, out_off_y, out_off_x, fusion, stream, workspace, ksplit);
}

namespace pgconv16 {
int launch_head16(const void* x, const float* w, const float* styles, const float* bias, const float* skip, float* y, int dtype, int N, int Cin, int64_t HW, int Cout,
                  float clamp, int up_w, hipStream_t s);      // conv1x1_head16.hip
}

PG_EXPORT int pg_conv1x1_small16(const void* x, const float* w, const float* styles, const float* bias, const float* skip, float* y,
                                 int dtype, int N, int Cin, int64_t HW, int Cout, float clamp, int skip_up2_width, void* stream) {
    if (!x || !w || !y || N <= 0 || Cin <= 0 || HW <= 0 || Cout <= 0) return PG_ERR_INVALID_ARG;
    if (skip_up2_width < 0 || (skip_up2_width && (!skip || skip_up2_width % 2 || HW % skip_up2_width || (HW / skip_up2_width) % 2))) return PG_ERR_INVALID_ARG;
    const int up_w = skip_up2_width;
    if (dtype != PG_BF16 && dtype != PG_F16) return PG_ERR_INVALID_ARG;
    if (Cout > 8 || Cin % 8 != 0 || (((uintptr_t)x) & 15) != 0 || (size_t)Cout * Cin * 4 > 64 * 1024) return PG_ERR_UNSUPPORTED;
    // round 5: the form that spreads a pixel's channels over the lanes of a wave (conv1x1_head16.hip) takes the three-channel heads it covers; PG_HEAD16_FORM=1 keeps
    // the first form (A/B)
    static const bool first_form = [] { const char* e = getenv("PG_HEAD16_FORM"); return e && atoi(e) == 1; }();
    if (!first_form) {
        const int st2 = pgconv16::launch_head16(x, w, styles, bias, skip, y, dtype, N, Cin, HW, Cout, clamp, up_w, (hipStream_t)stream);
        if (st2 != PG_ERR_UNSUPPORTED) return st2;
    }
    // lanes per pixel: enough threads for the whole chip on small images (256 CUs x 256 threads = 64 K lanes), never more lanes than 16-byte channel groups / 4
    static const int lp_force = [] { const char* e = getenv("PG_HEAD16_LP"); return e ? atoi(e) : 0; }();
    int lp = 1;
    if ((int64_t)N * HW * 4 <= 65536 && Cin >= 128) lp = 4;
    if ((int64_t)N * HW * 16 <= 65536 && Cin >= 512) lp = 16;
    if (lp_force == 1 || lp_force == 4 || lp_force == 16) lp = lp_force;
    const int ppb = 256 / lp;
    int64_t bx = (HW + ppb - 1) / ppb;
    const int64_t cap = (int64_t)pg::max_stream_blocks() / N > 0 ? (int64_t)pg::max_stream_blocks() / N : 1;
    if (bx > cap) bx = cap;
    const dim3 grid((unsigned)bx, (unsigned)N);
    const size_t lds = (size_t)Cout * Cin * 4;
    const hipStream_t st = (hipStream_t)stream;
    const unsigned short* xs = (const unsigned short*)x;
#define PG_HEAD(TT) (lp == 16 ? launch_small<TT, 16>(Cout, grid, lds, st, xs, w, styles, bias, skip, y, Cin, HW, clamp, up_w) \
                   : lp == 4 ? launch_small<TT, 4>(Cout, grid, lds, st, xs, w, styles, bias, skip, y, Cin, HW, clamp, up_w) \
                             : launch_small<TT, 1>(Cout, grid, lds, st, xs, w, styles, bias, skip, y, Cin, HW, clamp, up_w))
    if (dtype == PG_BF16) return PG_HEAD(bf16_t);
    return PG_HEAD(f16_t);
#undef PG_HEAD
}


// The up = 2 modulated 3x3 layer with the y half of the resampling filter folded into the weights and its x half applied in the epilogue
// (conv2d_up2f16.h).  See include/pasta_gan_ops.h.
PG_EXPORT int pg_conv2d16_up2_fused(const void* x, const void* packed, void* y, int dtype, int N, int Cin, int H, int W, int Cout,
                                    int64_t w_sample_stride, const int64_t ystride[4], const float fir_x[4],
                                    const pg_conv2d16_fusion* fusion, void* stream) {
    if (!x || !packed || !y || !ystride || !fir_x) return PG_ERR_INVALID_ARG;
    if (dtype != PG_BF16 && dtype != PG_F16) return PG_ERR_INVALID_ARG;
    if (N <= 0 || Cin <= 0 || H <= 0 || W <= 0 || Cout <= 0 || w_sample_stride < 0) return PG_ERR_INVALID_ARG;
    if (Cin % 16 != 0 || Cin < 32 || Cout % 32 != 0) return PG_ERR_UNSUPPORTED;                 // >= 2 K chunks per tile (two-role form), whole 32-cout blocks
    if ((((uintptr_t)x) & 15) != 0 || (((uintptr_t)packed) & 15) != 0 || (((uintptr_t)y) & 15) != 0) return PG_ERR_UNSUPPORTED;
    if (ystride[1] != 1 || ystride[0] % 8 != 0 || ystride[2] % 8 != 0 || ystride[3] % 8 != 0) return PG_ERR_UNSUPPORTED;      // 16-byte channels-last stores
    if ((int64_t)N * H * W * Cin * 2 > 0x7fffffffLL) return PG_ERR_TOO_LARGE;
    const int CoutP = round_up(4 * Cout, 64), CinP = round_up(Cin, 32);
    const int64_t per_sample = (int64_t)CinP * 6 * CoutP;
    if (w_sample_stride != 0 && w_sample_stride != per_sample) return PG_ERR_INVALID_ARG;
    const int64_t w_bytes = (w_sample_stride ? (int64_t)N * per_sample : per_sample) * 2;
    if (w_bytes > 0x7fffffffLL) return PG_ERR_TOO_LARGE;
    const int64_t ext = 1 + (int64_t)(N - 1) * ystride[0] + (int64_t)(Cout - 1) + (int64_t)(2 * H - 1) * ystride[2] + (int64_t)(2 * W - 1) * ystride[3];
    if (ext > 0x3fffffffLL) return PG_ERR_TOO_LARGE;
    pgconv16::Up2fParams pp;
    Conv16Params& p = pp.c;
    p = Conv16Params();
    p.x = x; p.wp = packed; p.y = y;
    p.w_nstride = w_sample_stride; p.w_bytes = w_bytes; p.y_bytes = ext * 2;
    p.N = N; p.Cin = Cin; p.xC = Cin; p.H = H; p.W = W; p.Cout = Cout; p.CoutP = CoutP; p.OH = 2 * H; p.OW = 2 * W;
    p.pad_y = 1; p.pad_x = 1; p.ksplit = 1;
    for (int i = 0; i < 4; i++) p.ys[i] = ystride[i];
    p.osy = p.osx = 2; p.out_mode = pgconv16::OUT_VEC16;
    if (fusion) {
        p.f = *fusion;
    } else {
        pg_conv2d16_fusion z = {};
        z.clamp = -1.f;
        p.f = z;
    }
    if (p.f.residual || p.f.noise_phase_stride < 0) return PG_ERR_INVALID_ARG;
    p.f.phase_cout = Cout;
    if (p.f.gain == 0.f) p.f.gain = 1.f;
    if (!(p.f.gain > 0.f)) return PG_ERR_UNSUPPORTED;
    if (p.f.act == 0) p.f.act = PG_ACT_LINEAR;
    if (p.f.act < PG_ACT_LINEAR || p.f.act > PG_ACT_SWISH) return PG_ERR_INVALID_ARG;
    if (p.f.act > PG_ACT_LRELU) return PG_ERR_UNSUPPORTED;
    if (p.f.act == PG_ACT_LRELU && (p.f.alpha < 0.f || p.f.alpha > 1.f)) return PG_ERR_UNSUPPORTED;
    for (int i = 0; i < 4; i++) pp.fir[i] = fir_x[i];
    { const char* e = getenv("PG_CONV16_DBG"); p.dbg = e ? atoi(e) : 0; }
    return pgconv16::launch16_up2f(pp, dtype, (hipStream_t)stream);
}
