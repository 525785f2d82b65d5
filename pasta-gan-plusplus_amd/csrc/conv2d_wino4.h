// Winograd F(4x4, 3x3) fp32 convolution for gfx950 on v_mfma_f32_32x32x2_f32 (round 3): 36 multiplies per 16 outputs -- 2.25 per
// output against 4 for F(2x2,3x3) (conv2d_wino.h) and 9 for the direct implicit GEMM (conv2d_kernel.h).
//
//   Y = A^T [ (G g G^T) .* (B^T d B) ] A    per 4x4 output tile / 6x6 input patch (Lavin & Gray's matrices, points 0, +-1, +-2, inf)
//   => 36 independent GEMMs  M_xi[co][tile] = sum_ci U_xi[co][ci] * V_xi[ci][tile],   xi = 6a + b
//
// Why the structure differs from conv2d_wino.h: on gfx950 the f32 MFMA runs at the VECTOR rate, so transform work is priced like
// matrix work.  F(2x2)'s input transform is 32 operations per (channel, tile) and each wave can rebuild its operands; F(4x4)'s is
// 144, which is only affordable once per workgroup tile.  Hence V IS materialised here: per 16-channel chunk a transform phase
// (512 threads = 16 channels x 32 tiles, one 6x6 patch each) writes V[36][16][32] to LDS, and a GEMM phase multiplies from it
// (B operand = V by ds_read, A operand = pre-transformed weights streamed from L2 by each wave, as in conv2d_wino.h).
//
// Mapping
//   * workgroup = 512 threads = 8 waves, ONE per CU (LDS 118-150 KB); tile = 64 couts x (8 output rows x 64 output cols) = 32 tiles of
//     4x4 outputs, MFMA column n = 2 * tile_x + tile_y (this interleave makes the 16-byte patch reads of the transform phase
//     conflict-free: tile_y adds 8 sixteen-byte slots mod 16).
//   * GEMM phase: wave w = (q = w >> 1, mt = w & 1) owns xi in [9q, 9q + 9) for cout M-tile mt: 9 accumulators = 144 VGPRs, two waves
//     per SIMD.  Per chunk 72 MFMAs per wave in six groups of (3 xi) x (4 channel pairs): three accumulators in rotation, the
//     group's A words are three 16-byte loads (1 KB per wave-instruction, contiguous 3 KB per group; packed by
//     pg_conv2d_winograd4_pack_weight in exactly the order the wave walks), its B words six ds_read2st64_b32.
//   * the raw 10-row halo tile [16][10][72] arrives by 16-byte LDS-DMA (waves 0-3 issue, gather map kept in LDS; zero padding = the
//     buffer range check on a sentinel offset), requested for chunk k+1 when the transform of chunk k is done: single raw buffer,
//     single V buffer, two barriers per chunk.
//   * U words are inline-asm loads waited for with hand-counted vmcnt (two groups in flight; the DMA sits between them in the
//     queue for the issuing waves: vmcnt(3 + 12) for the first two groups of a chunk, vmcnt(3) after -- the third group's wait is
//     also what guarantees the DMA has landed before the next chunk's barrier).
//   * tail: all 36 xi of a (cout, tile) live in different waves, so 16 couts per round go through LDS (the V buffer, dead by then);
//     512 threads then own one (cout, tile) each: inverse transform 6x6 -> 4x4 (100 VALU), fused epilogue of conv2d_wino.h on four
//     16-byte row segments.  SPADE mode: thread = (channel, tile, row pair) combining the gamma and beta rows of its channel.
// Numerics: float32 throughout, weights transformed in float64 by the pack kernel and rounded once; tools/f43_error_probe.py
// measures the effect on the whole config-2 network (3.7e-5 max-abs on `img` against the direct float32 run, F(2x2): 1.0e-5).
#pragma once
#include <cstdlib>
#include "conv2d_wino.h"

#ifndef WINO4_EXP
#define WINO4_EXP 0      // dev ablations (results wrong by design): 1 no U loads, 2 no transform phase, 4 no tail, 8 no halo DMA
#endif

namespace pgconv {

constexpr int W4_KC = 16;                        // input channels per chunk
constexpr int W4_ROWS = 10;                      // halo rows of an 8-row output tile
constexpr int W4_LROW = 72;                      // floats per LDS halo row: global columns [ox0 - 4, ox0 + 68) = 18 aligned 16-byte words
constexpr int W4_CHF = W4_ROWS * W4_LROW;        // 720 floats per channel
constexpr int W4_NX = W4_KC * W4_CHF;            // 11520 staged floats per chunk
constexpr int W4_NWORDS = W4_NX / 4;             // 2880 sixteen-byte words
constexpr int W4_NDMA = 12;                      // requests per issuing thread (256 issuing threads): ceil(2880 / 256)
constexpr int W4_RAW = W4_NX + 16;               // + room for the 0..3 float shift that aligns the patches
constexpr int W4_V = 36 * 512;                   // V[xi][channel 16][tile 32]  /  exchange [xi][cout 16][tile 32]
constexpr int W4_UGROUP = 3 * 1024;              // bytes of one A-operand group (3 xi x 64 lanes x 16 B)
constexpr int W4_UCHUNK = 6 * W4_UGROUP;         // per (wave unit, chunk)

// 1-D transforms.  B^T rows (input), A^T rows (output) of F(4,3) with points 0, 1, -1, 2, -2, inf.
__device__ __forceinline__ void w4_bt(float d0, float d1, float d2, float d3, float d4, float d5,
                                      float& r0, float& r1, float& r2, float& r3, float& r4, float& r5) {
    const float t0 = fmaf(-4.f, d2, d4), t1 = fmaf(-4.f, d1, d3);      // d4 - 4 d2,  d3 - 4 d1
    const float t2 = d4 - d2, u = d3 - d1;
    r0 = fmaf(4.f, d0, fmaf(-5.f, d2, d4));
    r1 = t0 + t1;
    r2 = t0 - t1;
    r3 = fmaf(2.f, u, t2);
    r4 = fmaf(-2.f, u, t2);
    r5 = fmaf(4.f, d1, fmaf(-5.f, d3, d5));
}
__device__ __forceinline__ void w4_at(float m0, float m1, float m2, float m3, float m4, float m5, float& y0, float& y1, float& y2, float& y3) {
    const float s12 = m1 + m2, d12 = m1 - m2, s34 = m3 + m4, d34 = m3 - m4;
    y0 = m0 + s12 + s34;
    y1 = fmaf(2.f, d34, d12);
    y2 = fmaf(4.f, s34, s12);
    y3 = fmaf(8.f, d34, d12) + m5;
}

// MODE: 0 = plain input, 1 = per-(n, channel) input scale (modulated convolution).  (Pre-activation launches stay on conv2d_wino.h.)
template <int MODE>
__global__ __launch_bounds__(512, 2) void conv2d_wino4(ConvParams p) {
    extern __shared__ __attribute__((aligned(16))) float smem[];
    const int cin_loop = ((p.Cin + W4_KC - 1) / W4_KC) * W4_KC;
    const int nchunks = cin_loop / W4_KC;
    float* raw = smem;
    float* V = smem + W4_RAW;
    float* cs0 = V + W4_V;                       // prologue scale of two consecutive tiles [2][cin_loop]
    float* ep0 = cs0 + 2 * cin_loop;             // epilogue scale / bias of two consecutive tiles [2][64 + 64]
    unsigned* gmI = (unsigned*)(ep0 + 256);      // gather map of an interior tile [12][256] (byte offsets relative to the tile's first halo sample)
    unsigned* gmE = gmI + W4_NDMA * 256;         // gather map of the current edge tile (absolute in the image, sentinel outside)

    const int t = threadIdx.x, lane = t & 63;
    const int wave = __builtin_amdgcn_readfirstlane(t >> 6);
    const unsigned smem_b = __builtin_amdgcn_readfirstlane(lds_offset(smem));
    const int half = lane >> 5, l31 = lane & 31;
    const int HW = p.H * p.W;
    const int total = p.total_tiles;
    const int q8 = total >> 3, r8 = total & 7;
    const int q = wave >> 1, mt = wave & 1;      // GEMM role
    const int sh = p.pad_x & 3;                  // float shift of the staged tile: patch column 0 of tile_x lands on LDS column 4 tile_x + cbase
    const int cbase = 4 - p.pad_x + sh;          // 4 (pad 0..3) or 0 (pad 4): a multiple of 4 -> 16-byte aligned patch reads

    int n = 0, oy0 = 0, ox0 = 0, m0 = 0;
    bool edge = false;
    i32x4 xrsrc;

    // Interior gather map: tile-independent, computed once.  Word f of the buffer = (channel f / 180, row (f % 180) / 18, word f % 18).
    if (wave < 4) {
#pragma unroll
        for (int i = 0; i < W4_NDMA; i++) {
            const int f = i * 256 + t;
            const int c = f / 180, rem = f % 180;
            const int row = rem / 18, wd = rem % 18;
            gmI[i * 256 + t] = (unsigned)(c * HW + row * p.W + 4 * wd) * 4u;
        }
    }                                            // (read back by the writing thread only: no barrier needed)

    auto prep_tile = [&](int tile, float* cs) {
        const int xcd = tile & 7;
        int L = (xcd < r8 ? xcd * (q8 + 1) : r8 * (q8 + 1) + (xcd - r8) * q8) + (tile >> 3);
        const int mb = L % p.mblocks; L /= p.mblocks;
        const int tx = L % p.tilesX; L /= p.tilesX;
        const int ty = L % p.tilesY;
        n = L / p.tilesY;
        oy0 = ty * 8; ox0 = tx * 64; m0 = mb * 64;
        if (MODE != 0) {
            const float* in_scale = p.f.in_scale ? p.f.in_scale + (int64_t)n * p.Cin : nullptr;
            for (int c = t; c < cin_loop; c += 512) cs[c] = ((in_scale && c < p.Cin) ? ld_opaque(in_scale + c) : 1.f) * p.f.in_gain;
        }
        const int gy0 = oy0 - p.pad_y, gx0 = ox0 - 4;
        edge = !(gy0 >= 0 && gy0 + W4_ROWS <= p.H && gx0 >= 0 && gx0 + W4_LROW <= p.W);           // wave-uniform
        if (edge && wave < 4) {
            int tt = t;
            asm volatile("" : "+v"(tt));             // keep the index maths inside the tile loop
#pragma unroll
            for (int i = 0; i < W4_NDMA; i++) {
                const int f = i * 256 + tt;
                const int c = f / 180, rem = f % 180;
                const int row = rem / 18, wd = rem % 18;
                const int gy = gy0 + row, gx = gx0 + 4 * wd;
                const bool ok = gy >= 0 && gy < p.H && gx >= 0 && gx + 4 <= p.W;
                gmE[i * 256 + tt] = ok ? (unsigned)(c * HW + gy * p.W + gx) * 4u : 0x80000000u;
            }
        }
        const int shift = edge ? 0 : (gy0 * p.W + gx0) * 4;             // interior: offsets are relative to the tile's first halo sample
        const uint64_t base = (uint64_t)(uintptr_t)(p.x + (int64_t)n * p.Cin * HW) + (uint64_t)(int64_t)shift;
        xrsrc[0] = __builtin_amdgcn_readfirstlane((int)(unsigned)base);
        xrsrc[1] = __builtin_amdgcn_readfirstlane((int)(unsigned)(base >> 32) & 0xffff);
        xrsrc[2] = p.Cin * HW * 4 - shift;                               // same absolute end: channels beyond Cin read as zero
        xrsrc[3] = 0x00020000;
    };

    auto issue_chunk = [&](int c0) {
#if !(WINO4_EXP & 8)
        if (wave < 4) {
            const unsigned xs_b = smem_b + (unsigned)sh * 4u;
            const int soff = c0 * HW * 4;
            const unsigned* gm = edge ? gmE : gmI;
            unsigned off[W4_NDMA];
#pragma unroll
            for (int i = 0; i < W4_NDMA; i++) off[i] = gm[i * 256 + t];
#pragma unroll
            for (int i = 0; i < W4_NDMA - 1; i++) dma_dwordx4_buf(xrsrc, xs_b + (unsigned)(256 * i + 64 * wave) * 16u, off[i], soff);
            if (wave == 0) dma_dwordx4_buf(xrsrc, xs_b + (unsigned)(256 * (W4_NDMA - 1)) * 16u, off[W4_NDMA - 1], soff);    // words 2816..2879
            else asm volatile("s_nop 0");
        }
#endif
    };
    // requests the DMA-issuing waves put in the queue per chunk (wave 0 one more)
    // -> counted waits below use the per-wave number

    f32x16 acc[9];

    // A-operand stream of this wave: [m-block][mt][q][chunk][group 6][xi 3][lane 64][4 pairs] floats, walked strictly forwards
    // inside a tile; two groups in flight.
    unsigned pa;
    auto a_reset = [&](int m0_) { pa = (unsigned)((((m0_ >> 6) * 2 + mt) * 4 + q) * nchunks) * (unsigned)W4_UCHUNK + (unsigned)lane * 16u; };
    auto load_u = [&](f32x4 (&dst)[3]) {
#if !(WINO4_EXP & 1)
        asm volatile("global_load_dwordx4 %0, %1, %2" : "=v"(dst[0]) : "v"(pa), "s"(p.wp));
        asm volatile("global_load_dwordx4 %0, %1, %2 offset:1024" : "=v"(dst[1]) : "v"(pa), "s"(p.wp));
        asm volatile("global_load_dwordx4 %0, %1, %2 offset:2048" : "=v"(dst[2]) : "v"(pa), "s"(p.wp));
#endif
        pa += (unsigned)W4_UGROUP;
    };
    auto wait_u = [&](f32x4 (&g)[3], bool dma_younger) {
        // 3 younger ring words, plus (first two groups of a chunk, issuing waves) the chunk's DMA requests issued after them
        if (dma_younger) {
            if (wave == 0) asm volatile("s_waitcnt vmcnt(%0)" :: "n"(3 + W4_NDMA));
            else asm volatile("s_waitcnt vmcnt(%0)" :: "n"(3 + W4_NDMA - 1));
        } else {
            asm volatile("s_waitcnt vmcnt(3)");
        }
        asm volatile("" : "+v"(g[0]), "+v"(g[1]), "+v"(g[2]));
    };

    typedef const __attribute__((address_space(4))) ConvParams* kernarg_t;
    auto fresh_args = [&]() {
        kernarg_t a = (kernarg_t)__builtin_amdgcn_kernarg_segment_ptr();
        asm volatile("" : "+s"(a));
        return a;
    };

    // transform role: channel 2 * wave + half of the chunk, tile n = l31 -> (tile_x = n >> 1, tile_y = n & 1)
    const float* rb = raw + (2 * wave + half) * W4_CHF + (4 * (l31 & 1)) * W4_LROW + 4 * (l31 >> 1) + cbase;
    float* vw = V + wave * 64 + lane;                                   // V[xi][2 wave + half][l31]
    const float* vb = V + 9 * q * 512 + lane;                           // B operand base: V[9q + j][pair][half][l31]

    int tile = blockIdx.x;
    int par = 0;
    prep_tile(tile, cs0);
    f32x4 ur[2][3];
    a_reset(m0);
    load_u(ur[0]);
    load_u(ur[1]);
    issue_chunk(0);
    dma_wait_all();                                                     // (also the two U groups: they are home before the first chunk)
    while (true) {
#pragma unroll
        for (int j = 0; j < 9; j++)
#pragma unroll
            for (int k = 0; k < 16; k++) acc[j][k] = 0.f;

        int e_n = n, e_oy0 = oy0, e_ox0 = ox0, e_m0 = m0;
        bool has_next = false;
        int next = tile;
        const float* cs_cur = cs0 + par * cin_loop;
        float* ep_scale = ep0 + par * 128;
        float* ep_bias = ep_scale + 64;

#pragma unroll 1
        for (int k = 0; k < nchunks; k++) {
            __syncthreads();                                            // A: raw(k) has landed (each issuing wave waited for its own requests
                                                                        //    at its third U group), V is free (every wave is past GEMM(k-1) / the tail)
            // ---- transform phase: one 6x6 patch per thread -> 36 V values
#if !(WINO4_EXP & 2)
            {
                float d[6][6];
#pragma unroll
                for (int r = 0; r < 6; r++) {
                    const f32x4 lo = *(const f32x4*)(rb + r * W4_LROW), hi = *(const f32x4*)(rb + r * W4_LROW + 4);
                    d[r][0] = lo[0]; d[r][1] = lo[1]; d[r][2] = lo[2]; d[r][3] = lo[3]; d[r][4] = hi[0]; d[r][5] = hi[1];
                }
                float sc = 1.f;
                if (MODE != 0) sc = cs_cur[k * W4_KC + 2 * wave + half];
#pragma unroll
                for (int j = 0; j < 6; j++)                              // columns: over the patch rows
                    w4_bt(d[0][j], d[1][j], d[2][j], d[3][j], d[4][j], d[5][j], d[0][j], d[1][j], d[2][j], d[3][j], d[4][j], d[5][j]);
#pragma unroll
                for (int a = 0; a < 6; a++) {                            // rows: over the patch columns, then out to V[6a + b]
                    float v0, v1, v2, v3, v4, v5;
                    w4_bt(d[a][0], d[a][1], d[a][2], d[a][3], d[a][4], d[a][5], v0, v1, v2, v3, v4, v5);
                    if (MODE != 0) { v0 *= sc; v1 *= sc; v2 *= sc; v3 *= sc; v4 *= sc; v5 *= sc; }
                    vw[(6 * a + 0) * 512] = v0; vw[(6 * a + 1) * 512] = v1; vw[(6 * a + 2) * 512] = v2;
                    vw[(6 * a + 3) * 512] = v3; vw[(6 * a + 4) * 512] = v4; vw[(6 * a + 5) * 512] = v5;
                }
            }
#endif
            __syncthreads();                                            // B: V(k) complete, raw free

            // ---- request the next chunk (of this tile, or the first of the next tile)
            bool issued = true;
            if (k + 1 < nchunks) {
                issue_chunk((k + 1) * W4_KC);
            } else {
                e_n = n; e_oy0 = oy0; e_ox0 = ox0; e_m0 = m0;
                const auto& qa = *fresh_args();
                if (qa.f.spade_x) {
                    if (t < 32) {
                        const int ch = (e_m0 >> 1) + t;
                        ep_scale[t] = ld_opaque(qa.f.spade_mean + e_n * (qa.Cout >> 1) + ch);
                        ep_bias[t] = ld_opaque(qa.f.spade_rstd + e_n * (qa.Cout >> 1) + ch);
                    }
                } else if (t < 64) {
                    const int co = e_m0 + t;
                    const bool ok = co < qa.Cout;
                    const int cc = ok ? co : 0;
                    const float scv = qa.f.out_scale ? ld_opaque(qa.f.out_scale + (int64_t)e_n * qa.Cout + cc) : 1.f;
                    const float bi = qa.f.bias ? ld_opaque(qa.f.bias + cc) : 0.f;
                    ep_scale[t] = ok ? scv : 0.f;
                    ep_bias[t] = ok ? bi : 0.f;
                }
                next = tile + gridDim.x;
                has_next = next < total;
                if (has_next) {
                    prep_tile(next, cs0 + (par ^ 1) * cin_loop);
                    issue_chunk(0);
                } else {
                    issued = false;
                }
            }
            const bool dma_q = issued && wave < 4;                       // this wave put DMA requests behind its two ring groups

            // ---- GEMM phase: 6 groups of (3 xi) x (4 channel pairs)
#pragma unroll
            for (int g = 0; g < 6; g++) {
                const int jg = g >> 1, quad = g & 1;
                float b[3][4];
#pragma unroll
                for (int jj = 0; jj < 3; jj++)
#pragma unroll
                    for (int s = 0; s < 4; s++) b[jj][s] = vb[(3 * jg + jj) * 512 + (4 * quad + s) * 64];
                wait_u(ur[g & 1], g < 2 && dma_q);
#pragma unroll
                for (int s = 0; s < 4; s++)
#pragma unroll
                    for (int jj = 0; jj < 3; jj++)
                        acc[3 * jg + jj] = __builtin_amdgcn_mfma_f32_32x32x2f32(ur[g & 1][jj][s], b[jj][s], acc[3 * jg + jj], 0, 0, 0);
                if (g == 4 && k + 1 == nchunks) a_reset(m0);             // from here on: the next tile's first groups (m0 is already the next tile's)
                load_u(ur[g & 1]);                                       // refill the slot two groups ahead
                __builtin_amdgcn_sched_barrier(0);
            }
            if (!issued) dma_wait_all();                                 // last chunk of the last tile: nothing counted behind us
        }

        // ---- inverse transform + fused epilogue, 16 couts per round through LDS (the V buffer)
#if WINO4_EXP & 4
        { float sm = 0.f;
          for (int j = 0; j < 9; j++) for (int k = 0; k < 16; k++) sm += acc[j][k];
          if (sm == 12345.678f) p.y[t] = sm; }
        __syncthreads();
        if (!has_next) break;
        tile = next; par ^= 1;
        continue;
#endif
        const auto& qa = *fresh_args();
        const bool spade = qa.f.spade_x != nullptr;
        const float gain = qa.f.gain, slope = act_slope(qa.f.act, qa.f.alpha);
        const float cl = qa.f.clamp >= 0.f ? qa.f.clamp : __builtin_inff();
        const bool plain_tail = slope == 1.f && gain == 1.f && qa.f.clamp < 0.f;
        const bool vec_ok = qa.ys[3] == 1 && ((qa.ys[0] | qa.ys[1] | qa.ys[2] | qa.f.noise_batch_stride) & 3) == 0 && (qa.OW & 3) == 0 &&
                            ((((uintptr_t)qa.y) | ((uintptr_t)qa.f.noise) | ((uintptr_t)qa.f.residual) | ((uintptr_t)qa.f.spade_x)) & 15) == 0;      // 16-byte row segments everywhere
        const bool full = vec_ok && e_oy0 + 8 <= qa.OH && e_ox0 + 64 <= qa.OW && e_m0 + 64 <= qa.Cout;      // wave-uniform
        const int fn = t & 31, ftx = fn >> 1, fty = fn & 1;              // finishing role: tile
        const int oyb = e_oy0 + 4 * fty, oxb = e_ox0 + 4 * ftx;
        float* ex = V;
        auto act4 = [&](f32x4 v) {
            if (!plain_tail) {
#pragma unroll
                for (int e = 0; e < 4; e++) v[e] = __builtin_amdgcn_fmed3f((v[e] > 0.f ? v[e] : v[e] * slope) * gain, -cl, cl);
            }
            return v;
        };
        // one output row segment (4 pixels) of channel `ch` at row oy: load / store with the full-tile fast path or guarded scalars
        auto load4 = [&](const float* base, int ch, int oy, bool chan_ok) {
            f32x4 r = {0.f, 0.f, 0.f, 0.f};
            const int64_t o = (int64_t)e_n * qa.ys[0] + (int64_t)ch * qa.ys[1] + (int64_t)oy * qa.ys[2] + (int64_t)oxb * qa.ys[3];
            if (full) return *(const f32x4*)(base + o);
            if (chan_ok && oy < qa.OH) {
#pragma unroll
                for (int e = 0; e < 4; e++) if (oxb + e < qa.OW) r[e] = base[o + e * qa.ys[3]];
            }
            return r;
        };
        auto store4 = [&](int ch, int oy, f32x4 v, bool chan_ok) {
            const int64_t o = (int64_t)e_n * qa.ys[0] + (int64_t)ch * qa.ys[1] + (int64_t)oy * qa.ys[2] + (int64_t)oxb * qa.ys[3];
            if (full) { *(f32x4*)(qa.y + o) = v; return; }
            if (chan_ok && oy < qa.OH) {
#pragma unroll
                for (int e = 0; e < 4; e++) if (oxb + e < qa.OW) qa.y[o + e * qa.ys[3]] = v[e];
            }
        };
        auto noise4 = [&](int oy) {
            f32x4 r = {0.f, 0.f, 0.f, 0.f};
            if (!qa.f.noise) return r;
            const float* nzp = qa.f.noise + (int64_t)e_n * qa.f.noise_batch_stride + (int64_t)oy * qa.OW + oxb;
            if (full) return *(const f32x4*)nzp * qa.f.noise_gain;     // OW % 4 == 0 and oxb % 4 == 0; the noise tensor's base is checked by the host
            if (oy < qa.OH) {
#pragma unroll
                for (int e = 0; e < 4; e++) if (oxb + e < qa.OW) r[e] = nzp[e] * qa.f.noise_gain;
            }
            return r;
        };
#pragma unroll
        for (int rnd = 0; rnd < 4; rnd++) {
            // couts 8 rnd + 4 half + i of this wave's M-tile live in accumulator registers 4 rnd + i
#pragma unroll
            for (int j = 0; j < 9; j++)
#pragma unroll
                for (int i = 0; i < 4; i++) ex[(9 * q + j) * 512 + (mt * 8 + 4 * half + i) * 32 + l31] = acc[j][4 * rnd + i];
            __syncthreads();
            if (rnd == 0) {
                // the next tile's first two U groups (requested during the last chunk) must be home before this tile's stores enter
                // the queue: vmcnt counts stores too, and the counted waits of the next chunk assume only loads behind them
                asm volatile("s_waitcnt vmcnt(0)" : "+v"(ur[0][0]), "+v"(ur[0][1]), "+v"(ur[0][2]), "+v"(ur[1][0]), "+v"(ur[1][1]), "+v"(ur[1][2]));
            }
            if (!spade) {
                const int c16 = t >> 5;
                const int col = (c16 >> 3) * 32 + 8 * rnd + (c16 & 7);   // cout within the 64-block
                const int co = e_m0 + col;
                const bool chan_ok = co < qa.Cout;
                const int coc = chan_ok ? co : qa.Cout - 1;
                const float esc = ep_scale[col], ebi = ep_bias[col];
                const float* er = ex + t;                                // ex[xi][c16][fn] = ex[xi * 512 + t]
                float w[4][6];
#pragma unroll
                for (int b = 0; b < 6; b++)                              // over a, column b at a time (6 live inputs)
                    w4_at(er[(0 * 6 + b) * 512], er[(1 * 6 + b) * 512], er[(2 * 6 + b) * 512], er[(3 * 6 + b) * 512], er[(4 * 6 + b) * 512], er[(5 * 6 + b) * 512],
                          w[0][b], w[1][b], w[2][b], w[3][b]);
#pragma unroll
                for (int r = 0; r < 4; r++) {
                    float y0, y1, y2, y3;
                    w4_at(w[r][0], w[r][1], w[r][2], w[r][3], w[r][4], w[r][5], y0, y1, y2, y3);
                    const f32x4 y = {y0, y1, y2, y3};
                    f32x4 v = y * esc + (noise4(oyb + r) + ebi);
                    v = act4(v);
                    if (qa.f.residual) v += load4(qa.f.residual, coc, oyb + r, chan_ok);
                    store4(coc, oyb + r, v, chan_ok);
                }
            } else {
                // SPADE combine (networks.py:1715-1722): M-tile 0 rows are gamma, M-tile 1 rows beta of the same 32 channels;
                // thread = (channel c8, tile, row pair rh):  y = (x - mean) * rstd * (1 + gamma) + beta
                const int rh = t >> 8, c8 = (t >> 5) & 7;                // rh is wave-uniform
                const int chl = 8 * rnd + c8;
                const int ch = (e_m0 >> 1) + chl;
                const float mu = ep_scale[chl], rs = ep_bias[chl];
                f32x4 gb[2][2];                                          // [gamma | beta][row of the pair]
#pragma unroll
                for (int gbi = 0; gbi < 2; gbi++) {
                    const float* er = ex + (gbi * 8 + c8) * 32 + fn;
                    float z[6][4];
#pragma unroll
                    for (int a = 0; a < 6; a++)                          // over b, row a at a time
                        w4_at(er[(6 * a + 0) * 512], er[(6 * a + 1) * 512], er[(6 * a + 2) * 512], er[(6 * a + 3) * 512], er[(6 * a + 4) * 512], er[(6 * a + 5) * 512],
                              z[a][0], z[a][1], z[a][2], z[a][3]);
#pragma unroll
                    for (int c = 0; c < 4; c++) {                        // over a: only this thread's two output rows
                        const float s12 = z[1][c] + z[2][c], d12 = z[1][c] - z[2][c], s34 = z[3][c] + z[4][c], d34 = z[3][c] - z[4][c];
                        gb[gbi][0][c] = rh ? fmaf(4.f, s34, s12) : z[0][c] + s12 + s34;
                        gb[gbi][1][c] = rh ? fmaf(8.f, d34, d12) + z[5][c] : fmaf(2.f, d34, d12);
                    }
                }
#pragma unroll
                for (int rr = 0; rr < 2; rr++) {
                    const int oy = oyb + 2 * rh + rr;
                    const f32x4 x = load4(qa.f.spade_x, ch, oy, true);
                    f32x4 v = (x - mu) * rs * (gb[0][rr] + 1.f) + gb[1][rr];
                    v = act4(v);
                    store4(ch, oy, v, true);
                }
            }
            __syncthreads();                                             // the exchange area is rewritten by the next round / the next transform
        }
        if (!has_next) break;
        tile = next;
        par ^= 1;
    }
}

template <int MODE>
int launch_wino4_mode(const ConvParams& p0, hipStream_t s) {
    ConvParams p = p0;
    p.tilesX = (p.OW + 63) / 64;
    p.tilesY = (p.OH + 7) / 8;
    p.mblocks = p.CoutP / 64;
    const int64_t tiles = (int64_t)p.N * p.tilesX * p.tilesY * p.mblocks;
    if (tiles > 0x7fffffffLL) return PG_ERR_TOO_LARGE;
    p.total_tiles = (int)tiles;
    const int cin_loop = ((p.Cin + W4_KC - 1) / W4_KC) * W4_KC;
    const size_t lds = ((size_t)W4_RAW + W4_V + 2 * cin_loop + 256 + 2 * W4_NDMA * 256) * sizeof(float);
    if ((int64_t)36 * cin_loop * p.CoutP * 4 > 0x7fffffffLL) return PG_ERR_TOO_LARGE;      // the U stream uses 32-bit byte offsets
    if (lds > 160 * 1024) return PG_ERR_UNSUPPORTED;
    const int64_t blocks = tiles < (int64_t)num_cu() ? tiles : (int64_t)num_cu();
    static PerDeviceOnce lds_attr;
    const hipError_t e = lds_attr.run([] { return hipFuncSetAttribute((const void*)conv2d_wino4<MODE>, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024); });
    if (e != hipSuccess) return (int)e;
    hipLaunchKernelGGL((conv2d_wino4<MODE>), dim3((unsigned)blocks), dim3(512), lds, s, p);
    return launch_status();
}

}  // namespace pgconv
