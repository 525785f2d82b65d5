// Winograd F(2x2, 3x3) fp32 convolution for gfx950 on v_mfma_f32_32x32x2_f32: the stride-1 3x3 layers of the
// synthesis network (74 % of the step) with 2.25x fewer matrix FLOPs than the direct implicit GEMM of conv2d_kernel.h.
//
//   Y = A^T [ (G g G^T) .* (B^T d B) ] A      per 2x2 output tile and 4x4 input patch, summed over input channels
//   => 16 independent GEMMs  M_xi[co][tile] = sum_ci U_xi[co][ci] * V_xi[ci][tile],   xi = 4a + b
// (Lavin & Gray's matrices for cross-correlation, the operation F.conv2d / conv2d_gradfix.py:38 computes.)
//
// Mapping to the hardware
//   * workgroup = 512 threads = 8 waves; tile = 64 couts x (2 output rows x 64 output cols = 32 Winograd tiles);
//     wave w owns positions xi in {2w, 2w+1} (row transform a = w >> 1, column transforms b = 2(w&1), 2(w&1)+1) for both
//     32-cout M-tiles: 2 x 2 accumulators of 16 VGPRs = 64 VGPRs; MFMA lanes = the 32 tiles, K = input-channel pairs.
//   * B operand (V) is never materialised: the raw 4-row halo tile sits in LDS, de-interleaved by column parity so that
//     the stride-2 patch columns of neighbouring tiles are consecutive dwords (conflict-free), and each wave builds its two
//     V values from 8 LDS reads + a handful of adds (B^T rows have two +-1 entries each).  The halo tile arrives by LDS-DMA
//     with per-lane gather addresses (zero padding via the buffer range check), double buffered, one barrier per 16 channels.
//   * A operand (U = G g G^T, pre-transformed and cached by the host as [16][CinP][CoutP]) is used by exactly one wave,
//     so it is NOT staged in LDS: each wave streams its own slice from L2 into a 4-pair register ring.
//   * inverse transform: the 16 positions of one (cout, tile) live in 8 different waves, so the accumulators go through an
//     LDS exchange (8 couts per round, in the free staging buffer); 512 threads then apply A^T . A, the fused epilogue
//     (demodulation, noise, bias, activation, gain, clamp, residual) and store 2 adjacent pixels each (256-byte rows).
//   * persistent tile stream with cross-tile prefetch, XCD-aware tile order: as conv2d_kernel.h.
// Numerics: exact fp32 products/sums (same MFMA), different summation order; measured max error ~2e-6 of the output
// scale, the same level as the direct kernel (tests/test_hip_parity.py::test_conv2d_winograd_*).
#pragma once
#include "conv2d_kernel.h"

#ifndef WINO_EXP
#define WINO_EXP 0
#endif

namespace pgconv {

constexpr int W_KC = 16;                     // input channels per LDS chunk
constexpr int W_ROWF = 66;                   // halo row: 2 parities x 33 columns
constexpr int W_CHF = 4 * W_ROWF;            // floats per channel (4 halo rows)
constexpr int W_NX = W_KC * W_CHF;           // 4224 staged floats per chunk
constexpr int W_XPT = (W_NX + 511) / 512;    // 9 DMA dwords per thread per chunk
constexpr int W_BUF = W_XPT * 512;           // floats per staging buffer (padded to whole wave-instructions) = 4608
constexpr int W_RING = 2;                    // A-operand groups (of 2 channel pairs) in flight per wave
constexpr int W_EXCH = 16 * 8 * 32;          // exchange floats per round: 16 positions x 8 couts x 32 tiles = 4096 <= W_BUF

template <bool XF>
__global__ __launch_bounds__(512, 4) void conv2d_wino(ConvParams p) {
    extern __shared__ __attribute__((aligned(16))) float smem[];
    const int cin_loop = ((p.Cin + W_KC - 1) / W_KC) * W_KC;
    const int nchunks = cin_loop / W_KC;
    float* ex0 = smem + 2 * W_BUF;               // inverse-transform exchange, double buffered [2][W_EXCH]
    float* cs0 = ex0 + 2 * W_EXCH;               // prologue scale of two consecutive tiles [2][cin_loop]
    float* ep0 = cs0 + 2 * cin_loop;             // epilogue scale / bias of two consecutive tiles [2][64 + 64]

    const int t = threadIdx.x, lane = t & 63;
    const int wave = __builtin_amdgcn_readfirstlane(t >> 6);
    const unsigned smem_b = __builtin_amdgcn_readfirstlane(lds_offset(smem));
    const int half = lane >> 5, l31 = lane & 31;
    const int HW = p.H * p.W;
    const int total = p.total_tiles;
    const int q8 = total >> 3, r8 = total & 7;

    // this wave's transforms.  Row transform a = wave >> 1:  B^T row a = s0 * d[i0] + s1 * d[i1].
    // Column transforms: the wave's two positions need patch columns tb, tb+1, tb+2 (tb = wave & 1) =: A, B, C and are
    //   e = A - C            (b = 0 for tb = 0:  r0 - r2;   b = 3 for tb = 1:  r1 - r3)
    //   f = B + fc*C + fa*A  (b = 1 for tb = 0:  r1 + r2;   b = 2 for tb = 1:  r2 - r1)
    // so slot s of the wave is position xi(s) = 4a + (tb ? 3 - s : s) and the inner loop has no wave-dependent branch.
    const int ta = wave >> 1, tb = wave & 1;
    const int i0 = ta == 0 ? 0 : 1, i1 = ta == 3 ? 3 : 2;
    const float s0 = ta == 2 ? -1.f : 1.f, s1 = (ta == 0 || ta == 3) ? -1.f : 1.f;
    const float fc = tb ? 0.f : 1.f, fa = tb ? -1.f : 0.f;
    const int xi0 = 4 * ta + (tb ? 3 : 0), xi1 = 4 * ta + (tb ? 2 : 1);
    // LDS offsets (floats) of the six samples: rows i0 / i1, columns tb + m -> parity (tb+m)&1, index (tb+m)>>1
    int so0[3], so1[3];
#pragma unroll
    for (int m = 0; m < 3; m++) {
        const int col = ((tb + m) & 1) * 33 + ((tb + m) >> 1);
        so0[m] = i0 * W_ROWF + col;
        so1[m] = i1 * W_ROWF + col;
    }

    int n = 0, oy0 = 0, ox0 = 0, m0 = 0;
    unsigned xoff[W_XPT];
    i32x4 xrsrc;

    auto prep_tile = [&](int tile, float* cs) {
        const int xcd = tile & 7;
        int L = (xcd < r8 ? xcd * (q8 + 1) : r8 * (q8 + 1) + (xcd - r8) * q8) + (tile >> 3);
        const int mb = L % p.mblocks; L /= p.mblocks;
        const int tx = L % p.tilesX; L /= p.tilesX;
        const int ty = L % p.tilesY;
        n = L / p.tilesY;
        oy0 = ty * 2; ox0 = tx * 64; m0 = mb * 64;
        const float* in_scale = p.f.in_scale ? p.f.in_scale + (int64_t)n * p.Cin : nullptr;
        for (int c = t; c < cin_loop; c += 512) cs[c] = (in_scale && c < p.Cin) ? in_scale[c] : 1.f;
        int tt = t;
        asm volatile("" : "+v"(tt));                 // keep the index maths inside the tile loop (see conv2d_kernel.h)
#pragma unroll
        for (int i = 0; i < W_XPT; i++) {
            const int e = tt + 512 * i;
            const int c = e / W_CHF, rem = e % W_CHF;
            const int rr = rem / W_ROWF, qq = rem % W_ROWF;
            const int par = qq / 33, k = qq % 33;
            const int gy = oy0 - p.pad_y + rr, gx = ox0 - p.pad_x + 2 * k + par;
            const bool ok = e < W_NX && gy >= 0 && gy < p.H && gx >= 0 && gx < p.W;
            xoff[i] = ok ? (unsigned)(c * HW + gy * p.W + gx) * 4u : 0x80000000u;
        }
        const uint64_t base = (uint64_t)(uintptr_t)(p.x + (int64_t)n * p.Cin * HW);
        xrsrc[0] = __builtin_amdgcn_readfirstlane((int)(unsigned)base);
        xrsrc[1] = __builtin_amdgcn_readfirstlane((int)(unsigned)(base >> 32) & 0xffff);
        xrsrc[2] = p.Cin * HW * 4;
        xrsrc[3] = 0x00020000;
    };

    auto issue_chunk = [&](int c0, int buf) {
        const unsigned xs_b = smem_b + (unsigned)(buf * W_BUF + 64 * wave) * 4u;
        const int soff = c0 * HW * 4;
#if !(WINO_EXP & 8)
#pragma unroll
        for (int i = 0; i < W_XPT; i++) dma_dword(xrsrc, xs_b + 2048u * i, xoff[i], soff);
#endif
    };

    const float in_slope = act_slope(p.f.in_act, p.f.in_alpha);
    const float in_cl = p.f.in_clamp >= 0.f ? p.f.in_clamp : __builtin_inff();
    const float in_gain = XF ? p.f.in_gain : 1.f;

    f32x16 acc[2][2];                                // [position][M-tile]

    // A-operand stream.  U is packed [xi][co / 32][ci / 4][ci & 1][co & 31][(ci >> 1) & 1]: the two values a lane needs for
    // two consecutive channel pairs are one aligned 8-byte word and a wave-instruction reads 512 contiguous bytes.  Four
    // streams per wave (2 positions x 2 M-tiles) advance together through a ring of W_RING groups (of 2 pairs).  The ring
    // runs across tiles: the last groups of a tile already fetch the first groups of the next one, so no tile starts
    // with an exposed L2 round trip.
    typedef float f32x2 __attribute__((ext_vector_type(2)));
    const int NG = cin_loop / 4;                                             // groups per tile
    const unsigned d_mt = (unsigned)NG * 512u;                               // bytes to the next 32-cout block
    const unsigned d_s = (unsigned)((xi1 - xi0) * (p.CoutP / 32) * NG) * 512u;   // position xi0 -> xi1 (wraps: modular arithmetic)
    unsigned pa;                                                             // byte offset into the packed U (< 2^31: checked by the host)
    auto a_reset = [&]() { pa = (unsigned)((xi0 * (p.CoutP / 32) + (m0 >> 5)) * NG) * 512u + (unsigned)(half * 256 + l31 * 8); };
    // The U loads are issued from inline asm and waited for by hand.  The compiler cannot count the halo DMA (asm as well),
    // so its own `s_waitcnt vmcnt(n)` in front of every group would be too small by the 9 DMA requests issued in between --
    // each chunk would start by waiting for the halo tile it has just requested.  Loads return in order, so the exact
    // counts are (queue, oldest first, at the point of use within a chunk):
    //   group 0: U(k,0) U(k,1) DMA(k+1)          -> vmcnt(4 + 9)
    //   group 1: U(k,1) DMA(k+1) U(k,2)          -> vmcnt(9 + 4)
    //   group 2: DMA(k+1) U(k,2) U(k,3)          -> vmcnt(4)     (this is also what guarantees the DMA has landed)
    //   group 3: U(k,3) U(k+1,0)                 -> vmcnt(4)
    // Anything else in the queue (the epilogue's stores, residual loads) only makes these waits stricter.
    auto load_a = [&](f32x2 (&dst)[4]) {
        asm volatile("global_load_dwordx2 %0, %1, %2" : "=v"(dst[0]) : "v"(pa), "s"(p.wp));
        asm volatile("global_load_dwordx2 %0, %1, %2" : "=v"(dst[1]) : "v"(pa + d_mt), "s"(p.wp));
        asm volatile("global_load_dwordx2 %0, %1, %2" : "=v"(dst[2]) : "v"(pa + d_s), "s"(p.wp));
        asm volatile("global_load_dwordx2 %0, %1, %2" : "=v"(dst[3]) : "v"(pa + d_s + d_mt), "s"(p.wp));
        pa += 512u;
    };
    auto wait_a = [&](f32x2 (&g)[4], bool dma_younger) {     // ties the wait to the registers: every use comes after it
        if (dma_younger) asm volatile("s_waitcnt vmcnt(%4)" : "+v"(g[0]), "+v"(g[1]), "+v"(g[2]), "+v"(g[3]) : "n"(4 + W_XPT));
        else asm volatile("s_waitcnt vmcnt(4)" : "+v"(g[0]), "+v"(g[1]), "+v"(g[2]), "+v"(g[3]));
    };

    const float gain = p.f.gain;
    const float cl = p.f.clamp >= 0.f ? p.f.clamp : __builtin_inff();
    const float slope = act_slope(p.f.act, p.f.alpha);
    // both pixels of a thread's output pair in one 8-byte store: dense rows, even strides, aligned base
    const bool vec_store = p.ys[3] == 1 && ((p.ys[0] | p.ys[1] | p.ys[2]) & 1) == 0 && (((uintptr_t)p.y) & 7) == 0 && (p.OW & 1) == 0;

    int tile = blockIdx.x;
    int par = 0, g = 0;
    prep_tile(tile, cs0);
    f32x2 a_ring[W_RING][4];
    a_reset();
#pragma unroll
    for (int d = 0; d < W_RING; d++) load_a(a_ring[d]);
    issue_chunk(0, 0);
    dma_wait_all();
    __syncthreads();
    while (true) {
#pragma unroll
        for (int s = 0; s < 2; s++)
#pragma unroll
            for (int mt = 0; mt < 2; mt++)
#pragma unroll
                for (int k = 0; k < 16; k++) acc[s][mt][k] = 0.f;

        int e_n = n, e_oy0 = oy0, e_ox0 = ox0, e_m0 = m0;
        bool has_next = false;
        int next = tile;
        const float* cs_cur = cs0 + par * cin_loop;
        float* ep_scale = ep0 + par * 128;           // per tile parity: the next tile's constants are written while slow
        float* ep_bias = ep_scale + 64;              // waves may still be in this tile's epilogue
        for (int k = 0; k < nchunks; k++, g++) {
            const int buf = g & 1;
            if (k + 1 < nchunks) {
                issue_chunk((k + 1) * W_KC, buf ^ 1);
            } else {
                e_n = n; e_oy0 = oy0; e_ox0 = ox0; e_m0 = m0;
                if (p.f.spade_x) {                          // SPADE mode: per-(n, channel) mean / rstd of the normalised tensor
                    if (t < 32) {
                        const int ch = (e_m0 >> 1) + t;     // this tile's 32 output channels
                        ep_scale[t] = p.f.spade_mean[e_n * (p.Cout >> 1) + ch];
                        ep_bias[t] = p.f.spade_rstd[e_n * (p.Cout >> 1) + ch];
                    }
                } else if (t < 64) {
                    const int co = e_m0 + t;
                    const bool ok = co < p.Cout;
                    const int cc = ok ? co : 0;
                    const float sc = p.f.out_scale ? p.f.out_scale[(int64_t)e_n * p.Cout + cc] : 1.f;
                    const float bi = p.f.bias ? p.f.bias[cc] : 0.f;
                    ep_scale[t] = ok ? sc : 0.f;
                    ep_bias[t] = ok ? bi : 0.f;
                }
                next = tile + gridDim.x;
                has_next = next < total;
                if (has_next) {
                    prep_tile(next, cs0 + (par ^ 1) * cin_loop);
                    issue_chunk(0, buf ^ 1);
                } else {
                    dma_wait_all();                          // no DMA in this chunk: the counted waits below assume one
                }
            }
            // ---- multiply this chunk: 8 channel pairs x (2 positions x 2 M-tiles) MFMAs per wave.  The six raw samples
            // (and the prologue scale) of pair pp + 1 are requested from LDS before pair pp's MFMAs are issued.
            const float* xb = smem + buf * W_BUF + half * W_CHF + l31;       // this lane's tile, channel (2 pair + half)
            const float* csb = cs_cur + k * W_KC + half;
            float bq[2][7];
            auto read_b = [&](int pp, float (&dst)[7]) {
                const float* xc = xb + (2 * pp) * W_CHF;
#pragma unroll
                for (int m = 0; m < 3; m++) {
#if WINO_EXP & 2
                    dst[2 * m] = (float)m; dst[2 * m + 1] = (float)-m;
#else
                    dst[2 * m] = xc[so0[m]]; dst[2 * m + 1] = xc[so1[m]];
#endif
                }
                dst[6] = csb[2 * pp];
            };
            read_b(0, bq[0]);
#pragma unroll
            for (int pp = 0; pp < W_KC / 2; pp++) {
                const int gq = pp >> 1, j = pp & 1;
                if (pp + 1 < W_KC / 2) read_b(pp + 1, bq[(pp + 1) & 1]);
                __builtin_amdgcn_sched_barrier(0);                           // the requests go out BEFORE this pair's MFMAs
                const float sc = bq[pp & 1][6] * in_gain;
                float q[3];                                                  // row-transformed patch columns A, B, C
#pragma unroll
                for (int m = 0; m < 3; m++) {
                    float d0 = bq[pp & 1][2 * m], d1 = bq[pp & 1][2 * m + 1];
                    if (XF) {                                                // SPADE pre-activation acts on the raw samples
                        d0 *= sc; d1 *= sc;
                        d0 = __builtin_amdgcn_fmed3f(fmaxf(d0, d0 * in_slope), -in_cl, in_cl);
                        d1 = __builtin_amdgcn_fmed3f(fmaxf(d1, d1 * in_slope), -in_cl, in_cl);
                        q[m] = fmaf(d1, s1, d0 * s0);
                    } else {
                        q[m] = fmaf(d1, sc * s1, d0 * (sc * s0));
                    }
                }
                const float v0 = q[0] - q[2];
                const float v1 = fmaf(fa, q[0], fmaf(fc, q[2], q[1]));
                if (j == 0) wait_a(a_ring[gq % W_RING], gq < 2);
                acc[0][0] = __builtin_amdgcn_mfma_f32_32x32x2f32(a_ring[gq % W_RING][0][j], v0, acc[0][0], 0, 0, 0);
                acc[0][1] = __builtin_amdgcn_mfma_f32_32x32x2f32(a_ring[gq % W_RING][1][j], v0, acc[0][1], 0, 0, 0);
                acc[1][0] = __builtin_amdgcn_mfma_f32_32x32x2f32(a_ring[gq % W_RING][2][j], v1, acc[1][0], 0, 0, 0);
                acc[1][1] = __builtin_amdgcn_mfma_f32_32x32x2f32(a_ring[gq % W_RING][3][j], v1, acc[1][1], 0, 0, 0);
                if (j == 1) {                                                // group consumed: refill its slot W_RING groups ahead
                    if (gq == W_KC / 4 - W_RING && k + 1 == nchunks) a_reset();   // from here on: the next tile's first groups
#if !(WINO_EXP & 1)
                    load_a(a_ring[gq % W_RING]);
#endif
                }
                __builtin_amdgcn_sched_barrier(0);
            }
#if WINO_EXP & 16
            dma_wait_all();
#endif
            // No vmcnt wait here: group 2 already waited for U values requested AFTER this chunk's halo DMA, so the DMA has
            // landed.  (A vmcnt(0) would also wait for the U refills issued a moment ago: one exposed L2 round trip per chunk.)
            __syncthreads();
        }

        // ---- inverse transform + fused epilogue: the 16 positions of one (cout, tile) sit in 8 waves, so 8 couts per round
        // go through a double-buffered LDS exchange (one barrier per round), then each thread finishes two adjacent pixels.
#if WINO_EXP & 4
        { float sm = 0.f;
          for (int s = 0; s < 2; s++) for (int mt = 0; mt < 2; mt++) for (int k = 0; k < 16; k++) sm += acc[s][mt][k];
          if (sm == 12345.678f) p.y[t] = sm; }
        if (!has_next) break;
        tile = next; par ^= 1;
        continue;
#endif
        const int c_l = t >> 6, prow = (t >> 5) & 1, tcol = t & 31;          // this thread's output: cout, row of the 2x2, tile
        const int oy = e_oy0 + prow, ox = e_ox0 + 2 * tcol;
        const bool row_ok = oy < p.OH;
        const int oyc = row_ok ? oy : p.OH - 1;
        const int ox0c = ox < p.OW ? ox : p.OW - 1, ox1c = ox + 1 < p.OW ? ox + 1 : p.OW - 1;
        const int cstride = (int)p.ys[1];
        const bool spade = p.f.spade_x != nullptr;
        const int pix0 = (int)((int64_t)e_n * p.ys[0] + (int64_t)oyc * p.ys[2] + (int64_t)ox0c * p.ys[3]);
        const int pix1 = (int)((int64_t)e_n * p.ys[0] + (int64_t)oyc * p.ys[2] + (int64_t)ox1c * p.ys[3]);
        float nz0 = 0.f, nz1 = 0.f;
        if (p.f.noise) {
            const float* nzp = p.f.noise + (int)(e_n * p.f.noise_batch_stride) + oyc * p.OW;
            nz0 = nzp[ox0c] * p.f.noise_gain; nz1 = nzp[ox1c] * p.f.noise_gain;
        }
#pragma unroll
        for (int rnd = 0; rnd < 8; rnd++) {
            // couts [8 rnd, 8 rnd + 8) of the 64: M-tile mt = rnd >> 2, rows 8q + 4 half + j with q = rnd & 3 -> regs 4q + j
            const int mt = rnd >> 2, q = rnd & 3;
            float* ex = ex0 + (rnd & 1) * W_EXCH;                            // [16 positions][8 couts][32 tiles]
#pragma unroll
            for (int s = 0; s < 2; s++)
#pragma unroll
                for (int j = 0; j < 4; j++) ex[((s == 0 ? xi0 : xi1) * 8 + 4 * half + j) * 32 + l31] = acc[s][mt][4 * q + j];
            if (spade) {
                // SPADE combine (networks.py:1715-1722).  The packed rows interleave 4 gamma rows with the 4 beta rows of the
                // same channels, so one round holds gamma (exchange rows 0-3) and beta (rows 4-7) of 4 output channels:
                //   y = (x - mean) * rstd * (1 + gamma) + beta;      waves 0-3 finish one channel each
                const int chl = 4 * rnd + (c_l & 3);                         // channel within this tile's 32
                const int ch = (e_m0 >> 1) + chl;
                float x0 = 0.f, x1 = 0.f;
                if (c_l < 4) { x0 = p.f.spade_x[pix0 + ch * cstride]; x1 = p.f.spade_x[pix1 + ch * cstride]; }
                __syncthreads();
                if (c_l < 4) {
                    const float* exr = ex + c_l * 32 + tcol + (prow ? 4 * 256 : 0);
                    float Tg[4], Tb[4];
#pragma unroll
                    for (int b = 0; b < 4; b++) {
                        const float g0 = exr[(4 * 0 + b) * 256], g1 = exr[(4 * 1 + b) * 256], g2 = exr[(4 * 2 + b) * 256];
                        const float b0 = exr[(4 * 0 + b) * 256 + 128], b1 = exr[(4 * 1 + b) * 256 + 128], b2 = exr[(4 * 2 + b) * 256 + 128];
                        Tg[b] = prow == 0 ? g0 + g1 + g2 : g0 - g1 - g2;
                        Tb[b] = prow == 0 ? b0 + b1 + b2 : b0 - b1 - b2;
                    }
                    const float mu = ep_scale[chl], rs = ep_bias[chl];
                    const float v0 = (x0 - mu) * rs * (1.f + Tg[0] + Tg[1] + Tg[2]) + (Tb[0] + Tb[1] + Tb[2]);
                    const float v1 = (x1 - mu) * rs * (1.f + Tg[1] - Tg[2] - Tg[3]) + (Tb[1] - Tb[2] - Tb[3]);
                    const bool ok0 = row_ok && ox < p.OW, ok1 = row_ok && ox + 1 < p.OW;
                    if (vec_store) {
                        f32x2 vv; vv[0] = v0; vv[1] = v1;
                        if (ok0) *(f32x2*)(p.y + pix0 + ch * cstride) = vv;
                    } else {
                        if (ok0) p.y[pix0 + ch * cstride] = v0;
                        if (ok1) p.y[pix1 + ch * cstride] = v1;
                    }
                }
                continue;
            }
            const int co = e_m0 + 8 * rnd + c_l;
            const int coc = co < p.Cout ? co : p.Cout - 1;
            float r0 = 0.f, r1 = 0.f;
            if (p.f.residual) { r0 = p.f.residual[pix0 + coc * cstride]; r1 = p.f.residual[pix1 + coc * cstride]; }
            __syncthreads();
            // Y[prow][0..1] = sum_a At[prow][a] * (sum_b M[a][b] * At[q][b])
            float T[4];
            const float* exr = ex + c_l * 32 + tcol + (prow ? 4 * 256 : 0);  // prow 0 reads a = 0, 1, 2; prow 1 reads a = 1, 2, 3
#pragma unroll
            for (int b = 0; b < 4; b++) {
                const float m0v = exr[(4 * 0 + b) * 256], m1v = exr[(4 * 1 + b) * 256], m2v = exr[(4 * 2 + b) * 256];
                T[b] = prow == 0 ? m0v + m1v + m2v : m0v - m1v - m2v;
            }
            const float esc = ep_scale[8 * rnd + c_l], ebi = ep_bias[8 * rnd + c_l];
            float v0 = (T[0] + T[1] + T[2]) * esc + nz0 + ebi;
            float v1 = (T[1] - T[2] - T[3]) * esc + nz1 + ebi;
            v0 = v0 > 0.f ? v0 : v0 * slope;
            v1 = v1 > 0.f ? v1 : v1 * slope;
            v0 = fminf(fmaxf(v0 * gain, -cl), cl) + r0;
            v1 = fminf(fmaxf(v1 * gain, -cl), cl) + r1;
            const bool ok0 = row_ok && ox < p.OW && co < p.Cout, ok1 = row_ok && ox + 1 < p.OW && co < p.Cout;
            if (vec_store) {
                f32x2 vv; vv[0] = v0; vv[1] = v1;
                if (ok0) *(f32x2*)(p.y + pix0 + coc * cstride) = vv;
            } else {
                if (ok0) p.y[pix0 + coc * cstride] = v0;
                if (ok1) p.y[pix1 + coc * cstride] = v1;
            }
        }
        if (!has_next) break;
        tile = next;
        par ^= 1;
    }
}

template <bool XF>
int launch_wino_xf(const ConvParams& p0, hipStream_t s) {
    ConvParams p = p0;
    p.tilesX = (p.OW + 63) / 64;
    p.tilesY = (p.OH + 1) / 2;
    p.mblocks = p.CoutP / 64;
    const int64_t tiles = (int64_t)p.N * p.tilesX * p.tilesY * p.mblocks;
    if (tiles > 0x7fffffffLL) return PG_ERR_TOO_LARGE;
    p.total_tiles = (int)tiles;
    const int cin_loop = ((p.Cin + W_KC - 1) / W_KC) * W_KC;
    const size_t lds = ((size_t)2 * W_BUF + 2 * W_EXCH + 2 * cin_loop + 256) * sizeof(float);
    if ((int64_t)16 * cin_loop * p.CoutP * 4 > 0x7fffffffLL) return PG_ERR_TOO_LARGE;      // the U stream uses 32-bit byte offsets
    if (lds > 160 * 1024) return PG_ERR_UNSUPPORTED;
    int per_cu = (int)((160 * 1024) / lds);
    if (per_cu > 2) per_cu = 2;                             // 128 VGPRs x 8 waves per workgroup
    if (per_cu < 1) per_cu = 1;
    const int64_t blocks = tiles < (int64_t)kNumCU * per_cu ? tiles : (int64_t)kNumCU * per_cu;
    static bool attr_set = false;
    if (!attr_set) {
        hipError_t e = hipFuncSetAttribute((const void*)conv2d_wino<XF>, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
        if (e != hipSuccess) return (int)e;
        attr_set = true;
    }
    hipLaunchKernelGGL((conv2d_wino<XF>), dim3((unsigned)blocks), dim3(512), lds, s, p);
    return launch_status();
}


}  // namespace pgconv
