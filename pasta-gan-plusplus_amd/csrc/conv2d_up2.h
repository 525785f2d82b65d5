// Stride-2 transposed 3x3 convolution for gfx950, all four output parities in ONE pass over the input, on
// v_mfma_f32_32x32x2_f32 -- the `up = 2` SynthesisLayer of the generator (reference: conv2d_resample.py:125-142 ->
// conv2d_gradfix.conv_transpose2d(stride=2), networks.py:73-94 for the modulation around it).
//
//     y[n, co, 2 iy + ky, 2 ix + kx] += x[n, ci, iy, ix] * w[co, ci, ky, kx]            (conv_transpose2d, padding 0)
// Gather form per output parity (a, b) at position (q, r):  y[2q + a, 2r + b] = sum over the taps with ky = a (mod 2), kx = b (mod 2):
//     a = 0: (ky 0, iy q), (ky 2, iy q - 1)    a = 1: (ky 1, iy q)      -- same for b / kx / ix / r
// i.e. 4 + 2 + 2 + 1 = 9 tap-products per position, no multiply spent on stuffed zeros, no scatter.
//
// conv2d_kernel.h runs this as four launches (2x2, 2x1, 1x2 and 1x1 taps); the three small ones are bound by the per-chunk
// structure (halo staging, barrier, operand reads), not by their MFMAs, and every launch re-reads the whole input.  Here a
// workgroup stages the halo of an 8 x 32 block of positions ONCE per K chunk and issues the 9 tap-products of all four parities
// from it: 18 MFMAs per channel pair and wave (2 position rows) behind 9 weight + 6 input operand reads.
//   * workgroup = 4 waves, 32 couts (one M-tile) x 8 x 32 positions; a wave owns 2 position rows x 4 parities = 8 accumulators
//     (128 VGPRs) -> 2 waves / SIMD; operands of channel pair cp + 1 are requested before the MFMAs of pair cp are issued;
//   * staging, persistent XCD-aware tile stream, double buffering, zero padding through the buffer range check: as conv2d_kernel.h;
//   * weights: the plain 3x3 pack of pg_conv2d_pack_weight ([Cin][9][CoutP]), tap = 3 ky + kx;
//   * epilogue: * out_scale[n, co] (demodulation); a lane holds both x-parities of a position, i.e. two ADJACENT output pixels:
//     8-byte stores when the output row pitch is even (the host allocates the (2H+1) x (2W+1) result with a padded pitch);
//   * the tiles cover the positions q <= H, r < W, i.e. every output row and the columns 0 .. 2W-1: a 33rd position column would
//     cost a whole extra 32-wide tile column (half of all tiles at W = 32).  The last output column (ox = 2W: position r = W, whose
//     only non-zero samples are x[q, W-1] and x[q-1, W-1], i.e. taps 2, 8 of parity (0,0) and tap 5 of parity (1,0)) is covered by EDGE
//     tiles in the same launch: 32 couts x 256 positions DOWN that column (the 32 MFMA columns of a wave are 32 consecutive q),
//     3 instead of 18 MFMAs per channel pair and position row, staging the input column W-1 only.  They run as a short pass BEFORE
//     the tile stream, one per workgroup from the END of the grid -- the workgroups the static tile stream gives one tile less.
//     (Until round 3 the host ran this column as two thin launches of the tiled kernel on a side stream: 0.3-0.7 ms of residency each.)
// Roofline: MFMA; algorithmic FLOPs 2 * N * Cout * Cin * 9 * H * W (+ the one-position border).
#pragma once
#include "conv2d_kernel.h"

namespace pgconv {

typedef float f32x2s __attribute__((ext_vector_type(2)));

constexpr int U_BM = 32, U_KC = 8;
// The 32 MFMA columns of a wave are 32 / TRW rows x TRW columns of positions: TRW = 32 for the wide layers, 16 / 8 for the 16^2 / 8^2
// ones (a 32-wide row of positions would be mostly empty there).  Workgroup tile: 8 * (32 / TRW) rows x TRW columns.
// VEC (round 4, TRW = 32 only): the halo rows are staged as ALIGNED 16-byte words -- global columns [r0 - 4, r0 + 36), 10 words per row -- with
// buffer_load_dwordx4 ... lds: 3 DMA instructions per thread and chunk instead of 10 four-byte ones (the request issue is what the multiplying
// waves pay for: ~100 cycles each beside 72 MFMAs); halo column c sits at LDS column c + 3.  Needs W % 4 == 0 and a 16-byte aligned x.
template <int TRW, bool VEC = false> struct UGeo {
    static constexpr int RG = 32 / TRW, TQ = 8 * RG, IH = TQ + 1, IW = VEC ? TRW + 8 : TRW + 1, PLANE = IH * IW;      // halo: rows q0-1 .., cols r0-1 .. (VEC: r0-4 ..)
    static constexpr int NX = U_KC * PLANE, COL0 = VEC ? 3 : 0;
    static constexpr int XPT = VEC ? (NX / 4 + 255) / 256 : (U_KC * 297 + 255) / 256;     // DMA instructions per thread and chunk
};
constexpr int U_XPT = (U_KC * 297 + 255) / 256;         // 9 x 33, 17 x 17, 33 x 9: at most 297 halo samples per channel
constexpr int U_NW4 = U_KC * 9 * U_BM / 4, U_WPT = (U_NW4 + 255) / 256;
constexpr int U_LDS_X = 3072, U_LDS_W = U_WPT * 256 * 4, U_LDS_BUF = U_LDS_X + U_LDS_W;      // (3072 floats: 10 x 256 four-byte slots, or 3 x 256 sixteen-byte ones)
static_assert(U_LDS_X >= U_XPT * 256, "staging buffer");

struct Up2Params {
    const float* x; const float* wp; float* y;
    const float* in_scale; const float* out_scale;
    int N, Cin, H, W, Cout, CoutP;
    int64_t ys[4];
    int tilesX, tilesY, mblocks, total_tiles;
    int etilesY, edge_tiles;                    // edge tiles of the last output column: N x etilesY x mblocks
    int retilesX, redge_tiles;                  // (round 5) edge tiles of the last output ROW 2H, when the main tiles stop at position row H - 1: N x retilesX x mblocks
    // split-K (round 5; the 8^2 / 16^2 layers: 128 / 256 tiles of 64 serial K chunks each): every tile (edge tiles too) exists `ksplit` times, share z reduces the
    // chunks [z * cpk, (z + 1) * cpk) into slice z of the workspace (y + z * ws_slice, laid out like y); up2_sum_slices adds the slices up.  ksplit = 1: y itself.
    int ksplit, cpk;
    int64_t ws_slice;
};
constexpr int U_EQ = 256, U_EPLANE = U_EQ + 1;  // positions per edge tile; staged samples per channel (rows q0-1 .. q0+255 of column W-1)

template <bool MOD, int TRW, bool VEC = false>
__global__ __launch_bounds__(256, 2) void conv2d_up2(Up2Params p) {
    typedef UGeo<TRW, VEC> G;
    static_assert(VEC ? (G::XPT * 1024 <= U_LDS_X && TRW == 32) : G::PLANE <= 297, "halo tile larger than the staging buffer");
    extern __shared__ __attribute__((aligned(16))) float smem[];
    const int cin_loop = ((p.Cin + U_KC - 1) / U_KC) * U_KC;
    const int nchunks = cin_loop / U_KC;
    float* cs0 = smem + 2 * U_LDS_BUF;          // input scale of two consecutive tiles [2][cin_loop]
    float* ep0 = cs0 + 2 * cin_loop;            // output scale of two consecutive tiles [2][32]

    const int t = threadIdx.x, lane = t & 63;
    const int wave = __builtin_amdgcn_readfirstlane(t >> 6);
    const unsigned smem_b = __builtin_amdgcn_readfirstlane(lds_offset(smem));
    const int half = lane >> 5, l31 = lane & 31;
    const int HW = p.H * p.W;
    const int total = p.total_tiles;
    const int q8 = total >> 3, r8 = total & 7;

    int n = 0, q0 = 0, r0 = 0, m0 = 0, kz = 0, zz = 0;      // kz: first chunk of this share, zz: its workspace slice
    unsigned xoff[U_XPT];
    i32x4 xrsrc;
    bool halo_vec = false;                       // VEC: false during the edge pass (its column staging stays on the four-byte path)

    auto prep_tile = [&](int tile, float* cs) {
        const int xcd = tile & 7;
        int L = (xcd < r8 ? xcd * (q8 + 1) : r8 * (q8 + 1) + (xcd - r8) * q8) + (tile >> 3);
        zz = L % p.ksplit; L /= p.ksplit;
        kz = zz * p.cpk;
        const int mb = L % p.mblocks; L /= p.mblocks;
        const int tx = L % p.tilesX; L /= p.tilesX;
        const int ty = L % p.tilesY;
        n = L / p.tilesY;
        q0 = ty * G::TQ; r0 = tx * TRW; m0 = mb * U_BM;
        if (MOD)
            for (int c = t; c < cin_loop; c += 256) cs[c] = c < p.Cin ? ld_opaque(p.in_scale + (int64_t)n * p.Cin + c) : 1.f;
        int tt = t;
        asm volatile("" : "+v"(tt));                 // keep the index maths inside the tile loop (see conv2d_kernel.h)
        if (VEC) {
#pragma unroll
            for (int i = 0; i < G::XPT; i++) {       // 16-byte words: (channel, row, word of the row); W % 4 == 0: a word is inside or outside the image as a whole
                const int e = tt + 256 * i;
                constexpr int WPR = G::IW / 4, WPC = G::PLANE / 4;
                const int c = e / WPC, rem = e % WPC;
                const int gy = q0 - 1 + rem / WPR, gx = r0 - 4 + 4 * (rem % WPR);
                const bool ok = e < G::NX / 4 && gy >= 0 && gy < p.H && gx >= 0 && gx + 4 <= p.W;
                xoff[i] = ok ? (unsigned)(c * HW + gy * p.W + gx) * 4u : 0x80000000u;
            }
        } else {
#pragma unroll
        for (int i = 0; i < U_XPT; i++) {
            const int e = tt + 256 * i;
            const int c = e / G::PLANE, rem = e % G::PLANE;
            const int gy = q0 - 1 + rem / G::IW, gx = r0 - 1 + rem % G::IW;
            const bool ok = e < G::NX && gy >= 0 && gy < p.H && gx >= 0 && gx < p.W;     // (gy <= q0 + 7 may exceed H - 1 in a ragged last tile: zero)
            xoff[i] = ok ? (unsigned)(c * HW + gy * p.W + gx) * 4u : 0x80000000u;
        }
        }
        const uint64_t base = (uint64_t)(uintptr_t)(p.x + (int64_t)n * p.Cin * HW);
        xrsrc[0] = __builtin_amdgcn_readfirstlane((int)(unsigned)base);
        xrsrc[1] = __builtin_amdgcn_readfirstlane((int)(unsigned)(base >> 32) & 0xffff);
        xrsrc[2] = p.Cin * HW * 4;
        xrsrc[3] = 0x00020000;
    };

    auto issue_chunk = [&](int c0, int buf) {
        const unsigned xs_b = smem_b + (unsigned)(buf * U_LDS_BUF + 64 * wave) * 4u;
        const unsigned ws_b = smem_b + (unsigned)(buf * U_LDS_BUF + U_LDS_X + 256 * wave) * 4u;
        const int soff = c0 * HW * 4;
        if (VEC && halo_vec) {
            const unsigned xs4_b = smem_b + (unsigned)(buf * U_LDS_BUF) * 4u + (unsigned)(64 * wave) * 16u;
#pragma unroll
            for (int i = 0; i < G::XPT; i++) dma_dwordx4_buf(xrsrc, xs4_b + 4096u * i, xoff[i], soff);
        } else {
#pragma unroll
            for (int i = 0; i < U_XPT; i++) dma_dword(xrsrc, xs_b + 1024u * i, xoff[i], soff);
        }
        const float* wb = p.wp + (int64_t)c0 * 9 * p.CoutP + m0;
#pragma unroll
        for (int i = 0; i < U_WPT; i++) {
            int e4 = t + 256 * i;
            if (U_NW4 % 256 != 0 && e4 >= U_NW4) e4 = U_NW4 - 1;       // clamp: the pad lanes copy a duplicate
            const int row = (e4 * 4) / U_BM, col = (e4 * 4) % U_BM;
            dma_dwordx4(wb + (int64_t)row * p.CoutP + col, ws_b + 4096u * i);
        }
    };

    // ---- edge pass: the last output column (see the header comment).  Plain double-buffered chunk loop, no cross-tile pipelining.
    static_assert(U_KC * U_EPLANE <= U_XPT * 256, "edge column larger than the staging buffer");
    // Round 5: the position row q = H (output row 2H: taps ky = 2 only, i.e. taps 6, 8 of parity (0,0) and tap 7 of parity (0,1) on the input row H - 1) is the
    // same kind of sliver.  Where it would open a whole extra row of main tiles (H % TQ == 0: 33 rows for a 32-row image in 8-row tiles, 17 in 16-row tiles) the
    // launcher stops the main tiles at q = H - 1 and this pass runs ROW tiles after the column tiles: 32 couts x 256 positions ALONG that row, the roles of q and r
    // swapped (staged: input row H - 1, columns r0 - 1 .. r0 + 255).  The corner (2H, 2W) belongs to the column tiles.
    for (int et = (int)gridDim.x - 1 - (int)blockIdx.x; et < p.edge_tiles + p.redge_tiles; et += gridDim.x) {
        const bool row_tile = et >= p.edge_tiles;
        int L = row_tile ? et - p.edge_tiles : et;
        const int ez = L % p.ksplit; L /= p.ksplit;
        const int ekz = ez * p.cpk;
        const int mb = L % p.mblocks; L /= p.mblocks;
        const int ety = row_tile ? p.retilesX : p.etilesY;
        const int ty = L % ety;
        n = L / ety; q0 = ty * U_EQ; m0 = mb * U_BM;              // (row tiles: q0 is the first COLUMN position r0 of the tile)
        if (MOD)
            for (int c = t; c < cin_loop; c += 256) cs0[c] = c < p.Cin ? ld_opaque(p.in_scale + (int64_t)n * p.Cin + c) : 1.f;
        if (t < U_BM) {
            const int co = m0 + t;
            ep0[t] = co < p.Cout ? (p.out_scale ? ld_opaque(p.out_scale + (int64_t)n * p.Cout + co) : 1.f) : 0.f;
        }
#pragma unroll
        for (int i = 0; i < U_XPT; i++) {
            const int e = t + 256 * i;
            const int c = e / U_EPLANE, gy = q0 - 1 + e % U_EPLANE;     // (row tiles: the column index)
            const bool ok = e < U_KC * U_EPLANE && gy >= 0 && gy < (row_tile ? p.W : p.H);
#if defined(UX_EXP) && (UX_EXP & 64)             // dev ablation (wrong by design): the column tiles read contiguous samples
            xoff[i] = ok ? (unsigned)(row_tile ? c * HW + (p.H - 1) * p.W + gy : c * HW + gy) * 4u : 0x80000000u;
#else
            xoff[i] = ok ? (unsigned)(row_tile ? c * HW + (p.H - 1) * p.W + gy : c * HW + gy * p.W + p.W - 1) * 4u : 0x80000000u;
#endif
        }
        const uint64_t base = (uint64_t)(uintptr_t)(p.x + (int64_t)n * p.Cin * HW);
        xrsrc[0] = __builtin_amdgcn_readfirstlane((int)(unsigned)base);
        xrsrc[1] = __builtin_amdgcn_readfirstlane((int)(unsigned)(base >> 32) & 0xffff);
        xrsrc[2] = p.Cin * HW * 4;
        xrsrc[3] = 0x00020000;
        issue_chunk(ekz * U_KC, 0);
        f32x16 ea[2][2];                             // [output row parity a][position row of the wave]
#pragma unroll
        for (int a = 0; a < 2; a++)
#pragma unroll
            for (int nt = 0; nt < 2; nt++)
#pragma unroll
                for (int k = 0; k < 16; k++) ea[a][nt][k] = 0.f;
        for (int k = 0; k < p.cpk; k++) {
            const int buf = k & 1;
            dma_wait_all();
            __syncthreads();
            if (k + 1 < p.cpk) issue_chunk((ekz + k + 1) * U_KC, buf ^ 1);
#pragma unroll
            for (int cp = 0; cp < U_KC / 2; cp++) {
                const float* ab = smem + buf * U_LDS_BUF + U_LDS_X + ((2 * cp + half) * 9) * U_BM + l31;
                const float* bb = smem + buf * U_LDS_BUF + (2 * cp + half) * U_EPLANE + (2 * wave) * 32 + l31;
                const float a2 = ab[(row_tile ? 6 : 2) * U_BM], a5 = ab[(row_tile ? 7 : 5) * U_BM], a8 = ab[8 * U_BM];      // row tiles: taps 6 (with x0), 7 (x0), 8 (xm)
                const float sc = MOD ? cs0[(ekz + k) * U_KC + 2 * cp + half] : 1.f;
#pragma unroll
                for (int nt = 0; nt < 2; nt++) {
                    const float xm = MOD ? bb[nt * 32] * sc : bb[nt * 32], x0 = MOD ? bb[nt * 32 + 1] * sc : bb[nt * 32 + 1];      // x[q - 1, W - 1], x[q, W - 1]
                    ea[0][nt] = __builtin_amdgcn_mfma_f32_32x32x2f32(a2, x0, ea[0][nt], 0, 0, 0);
                    ea[0][nt] = __builtin_amdgcn_mfma_f32_32x32x2f32(a8, xm, ea[0][nt], 0, 0, 0);
                    ea[1][nt] = __builtin_amdgcn_mfma_f32_32x32x2f32(a5, x0, ea[1][nt], 0, 0, 0);
                }
            }
        }
        // D col = lane & 31 = position q, row = cout (as below): output rows 2q and 2q + 1 of column 2W (row tiles: columns 2r and 2r + 1 of row 2H)
#pragma unroll
        for (int nt = 0; nt < 2; nt++) {
            const int q = q0 + (2 * wave + nt) * 32 + l31;
#pragma unroll
            for (int a = 0; a < 2; a++) {
                const bool row_ok = row_tile ? q < p.W : (a == 0 ? q <= p.H : q < p.H);
#pragma unroll
                for (int k = 0; k < 16; k++) {
                    const int rowc = (k & 3) + 8 * (k >> 2) + 4 * half;
                    const int co = m0 + rowc;
                    if (row_ok && co < p.Cout)
                        p.y[(int64_t)ez * p.ws_slice + (int64_t)n * p.ys[0] + (int64_t)co * p.ys[1] + (int64_t)(row_tile ? 2 * p.H : 2 * q + a) * p.ys[2] +
                            (int64_t)(row_tile ? 2 * q + a : 2 * p.W) * p.ys[3]] = ea[a][nt][k] * ep0[rowc];
                }
            }
        }
        __syncthreads();                             // the staging buffers and scales are rewritten next
    }
    if ((int)blockIdx.x >= p.total_tiles) return;    // a workgroup launched for an edge tile only
    halo_vec = VEC;

    f32x16 acc[4][2];                                // [parity 2a + b][position row of the wave]

    // operands of one channel pair: 9 taps of this lane's cout, and the 2 x 2 input samples each of its two positions touches
    struct Ops { float a[9]; float b[2][2][2]; };           // b[position row nt][input row q - 1 | q][input col r - 1 | r]
    const int lr = l31 / TRW, lc = l31 % TRW;
    auto fetch = [&](int buf, const float* cs, int c0, int cp, Ops& o) __attribute__((always_inline)) {
        const float* ab = smem + buf * U_LDS_BUF + U_LDS_X + ((2 * cp + half) * 9) * U_BM + l31;
        const float* bb = smem + buf * U_LDS_BUF + (2 * cp + half) * G::PLANE + ((2 * wave) * G::RG + lr) * G::IW + lc + G::COL0;
#pragma unroll
        for (int tp = 0; tp < 9; tp++) o.a[tp] = ab[tp * U_BM];
        const float sc = MOD ? cs[c0 + 2 * cp + half] : 1.f;
#pragma unroll
        for (int nt = 0; nt < 2; nt++)
#pragma unroll
            for (int dy = 0; dy < 2; dy++)
#pragma unroll
                for (int cc = 0; cc < 2; cc++) {
                    const float v = bb[(nt * G::RG + dy) * G::IW + cc];       // (RG = 1: the two position rows share a halo row; the compiler merges the reads)
                    o.b[nt][dy][cc] = MOD ? v * sc : v;
                }
    };
    auto mma = [&](const Ops& o) __attribute__((always_inline)) {
#pragma unroll
        for (int nt = 0; nt < 2; nt++) {
            const float xm_m = o.b[nt][0][0], xm_0 = o.b[nt][0][1], x0_m = o.b[nt][1][0], x0_0 = o.b[nt][1][1];
#define PG_MMA(ph, tap, v) acc[ph][nt] = __builtin_amdgcn_mfma_f32_32x32x2f32(o.a[tap], v, acc[ph][nt], 0, 0, 0)
            PG_MMA(0, 0, x0_0); PG_MMA(0, 2, x0_m); PG_MMA(0, 6, xm_0); PG_MMA(0, 8, xm_m);      // (0,0): ky,kx in {0,2}
            PG_MMA(1, 1, x0_0); PG_MMA(1, 7, xm_0);                                              // (0,1): ky in {0,2}, kx = 1
            PG_MMA(2, 3, x0_0); PG_MMA(2, 5, x0_m);                                              // (1,0): ky = 1, kx in {0,2}
            PG_MMA(3, 4, x0_0);                                                                  // (1,1): centre tap
#undef PG_MMA
        }
    };

    int tile = blockIdx.x, par = 0, g = 0;
    prep_tile(tile, cs0);
    issue_chunk(kz * U_KC, 0);
    dma_wait_all();
    __syncthreads();
    while (true) {
#pragma unroll
        for (int ph = 0; ph < 4; ph++)
#pragma unroll
            for (int nt = 0; nt < 2; nt++)
#pragma unroll
                for (int k = 0; k < 16; k++) acc[ph][nt][k] = 0.f;
        int e_n = n, e_q0 = q0, e_r0 = r0, e_m0 = m0;
        const int kz_cur = kz, z_cur = zz;           // (prep_tile of the NEXT tile overwrites kz / zz during this tile's last chunk)
        bool has_next = false;
        int next = tile;
        const float* cs_cur = cs0 + par * cin_loop;
        float* ep_scale = ep0 + par * U_BM;
        for (int k = 0; k < p.cpk; k++, g++) {
            const int buf = g & 1;
            if (k + 1 < p.cpk) {
                issue_chunk((kz_cur + k + 1) * U_KC, buf ^ 1);
            } else {
                e_n = n; e_q0 = q0; e_r0 = r0; e_m0 = m0;
                if (t < U_BM) {
                    const int co = e_m0 + t;
                    ep_scale[t] = co < p.Cout ? (p.out_scale ? ld_opaque(p.out_scale + (int64_t)e_n * p.Cout + co) : 1.f) : 0.f;
                }
                next = tile + gridDim.x;
                has_next = next < total;
                if (has_next) {
                    prep_tile(next, cs0 + (par ^ 1) * cin_loop);
                    issue_chunk(kz * U_KC, buf ^ 1);
                }
            }
            Ops cur, nxt;
            fetch(buf, cs_cur, (kz_cur + k) * U_KC, 0, cur);
#pragma unroll
            for (int cp = 0; cp < U_KC / 2; cp++) {
                if (cp + 1 < U_KC / 2) fetch(buf, cs_cur, (kz_cur + k) * U_KC, cp + 1, nxt);
                __builtin_amdgcn_sched_barrier(0);
                mma(cur);
                __builtin_amdgcn_sched_barrier(0);
                cur = nxt;
            }
            dma_wait_all();
            __syncthreads();
        }

        // ---- epilogue: D col = lane & 31 = position r, row = (reg & 3) + 8 * (reg >> 2) + 4 * half = cout.  A lane holds the two
        // x-parities of its position = two adjacent output pixels.
        const int r = e_r0 + lc;
        const bool pair_ok = (p.ys[2] & 1) == 0 && (p.ys[1] & 1) == 0 && (p.ys[0] & 1) == 0 && (((uintptr_t)p.y) & 7) == 0 && p.ys[3] == 1;
#pragma unroll
        for (int nt = 0; nt < 2; nt++) {
            const int q = e_q0 + (2 * wave + nt) * G::RG + lr;
#pragma unroll
            for (int a = 0; a < 2; a++) {
                const bool row_ok = a == 0 ? q <= p.H : q < p.H;
                const int oy = 2 * q + a;
                const bool ok0 = row_ok && r < p.W, ok1 = ok0;
#pragma unroll
                for (int k = 0; k < 16; k++) {
                    const int rowc = (k & 3) + 8 * (k >> 2) + 4 * half;
                    const int co = e_m0 + rowc;
                    const float sc = ep_scale[rowc];
                    const float v0 = acc[2 * a][nt][k] * sc, v1 = acc[2 * a + 1][nt][k] * sc;
                    if (co < p.Cout) {
                        float* dst = p.y + (int64_t)z_cur * p.ws_slice + (int64_t)e_n * p.ys[0] + (int64_t)co * p.ys[1] + (int64_t)oy * p.ys[2] + (int64_t)(2 * r) * p.ys[3];
                        if (pair_ok && ok1) {
                            *(f32x2s*)dst = (f32x2s){v0, v1};
                        } else {
                            if (ok0) dst[0] = v0;
                            if (ok1) dst[p.ys[3]] = v1;
                        }
                    }
                }
            }
        }
        if (!has_next) break;
        tile = next;
        par ^= 1;
    }
}

template <int TRW, bool VEC = false>
int launch_up2_t(const Up2Params& p0, hipStream_t s, bool edges_only = false) {
    typedef UGeo<TRW, VEC> G;
    Up2Params p = p0;
    p.tilesX = (p.W + TRW - 1) / TRW;
    static const bool redge_on = [] { const char* e = getenv("PG_UP2_ROW_EDGE"); return !e || atoi(e) != 0; }();      // A/B: 0 = the main tiles cover position row H
    const bool redge = redge_on && p.H % G::TQ == 0;          // row H would open a whole extra row of main tiles: it goes to the edge pass instead
    p.tilesY = redge ? p.H / G::TQ : (p.H + 1 + G::TQ - 1) / G::TQ;
    p.mblocks = p.CoutP / U_BM;
    const int cin_chunks = (p.Cin + U_KC - 1) / U_KC;
    if (p.ksplit < 1 || cin_chunks % p.ksplit != 0) return PG_ERR_INVALID_ARG;
    p.cpk = cin_chunks / p.ksplit;
    const int64_t tiles = (int64_t)p.N * p.tilesX * p.tilesY * p.mblocks * p.ksplit;
    p.etilesY = (p.H + 1 + U_EQ - 1) / U_EQ;
    const int64_t etiles = (int64_t)p.N * p.etilesY * p.mblocks * p.ksplit;
    if (tiles + etiles > 0x7fffffffLL) return PG_ERR_TOO_LARGE;
    p.total_tiles = edges_only ? 0 : (int)tiles;           // edges_only (conv2d_up2x3.h runs the main tiles): the edge pass, then every workgroup leaves
    p.edge_tiles = (int)etiles;
    p.retilesX = (p.W + U_EQ - 1) / U_EQ;
    const int64_t retiles = redge ? (int64_t)p.N * p.retilesX * p.mblocks * p.ksplit : 0;
    if (tiles + etiles + retiles > 0x7fffffffLL) return PG_ERR_TOO_LARGE;
    p.redge_tiles = (int)retiles;
#if defined(UX_EXP) && (UX_EXP & (128 | 256 | 512))   // dev ablations (wrong by design): no edge pass / no column tiles / no row tiles
    if (edges_only && (UX_EXP & 128)) return PG_OK;
    if (edges_only && (UX_EXP & 256)) p.edge_tiles = 0;
    if (edges_only && (UX_EXP & 512)) p.redge_tiles = 0;
#endif
    const int cin_loop = ((p.Cin + U_KC - 1) / U_KC) * U_KC;
    const size_t lds = ((size_t)2 * U_LDS_BUF + 2 * cin_loop + 2 * U_BM) * sizeof(float);
    if (lds > 160 * 1024) return PG_ERR_UNSUPPORTED;
    int per_cu = (int)((160 * 1024) / lds);
    if (per_cu > 2) per_cu = 2;                             // ~230 VGPRs x 4 waves per workgroup
    const int64_t work = (edges_only ? 0 : tiles) + p.edge_tiles + p.redge_tiles;
    const int64_t blocks = work < (int64_t)num_cu() * per_cu ? work : (int64_t)num_cu() * per_cu;
    if (p.in_scale) {
        static PerDeviceOnce a1;
        const hipError_t e = a1.run([] { return hipFuncSetAttribute((const void*)conv2d_up2<true, TRW, VEC>, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024); });
        if (e != hipSuccess) return (int)e;
        hipLaunchKernelGGL((conv2d_up2<true, TRW, VEC>), dim3((unsigned)blocks), dim3(256), lds, s, p);
    } else {
        static PerDeviceOnce a0;
        const hipError_t e = a0.run([] { return hipFuncSetAttribute((const void*)conv2d_up2<false, TRW, VEC>, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024); });
        if (e != hipSuccess) return (int)e;
        hipLaunchKernelGGL((conv2d_up2<false, TRW, VEC>), dim3((unsigned)blocks), dim3(256), lds, s, p);
    }
    return launch_status();
}

// Tiles of the main stream for a launch (what launch_up2 dispatches), for the split-K plan.
inline int64_t up2_tiles(int N, int H, int W, int Cout) {
    const int trw = W > 16 ? 32 : (W > 8 ? 16 : 8), tq = 8 * (32 / trw);
    return (int64_t)N * ((W + trw - 1) / trw) * (H % tq == 0 ? H / tq : (H + 1 + tq - 1) / tq) * ((Cout + U_BM - 1) / U_BM);
}

// The largest share count in {1, 2, 4, 8} that divides the K chunks, leaves >= 8 chunks per share and does not overfill the chip (two workgroups per CU): the
// layers whose tiles do not fill the chip -- 8^2 at N = 8: 128 tiles x 64 serial chunks of ~2.8 us, 202 -> 133 us with four shares.  PG_UP2_SPLITK=0: never (A/B).
inline int up2_splitk_plan(int N, int Cin, int H, int W, int Cout) {
    static const bool on = [] { const char* e = getenv("PG_UP2_SPLITK"); return !e || atoi(e) != 0; }();
    if (!on) return 1;
    const int chunks = (Cin + U_KC - 1) / U_KC;
    const int64_t tiles = up2_tiles(N, H, W, Cout);
    int best = 1;
    for (int k = 2; k <= 8; k *= 2)
        if (chunks % k == 0 && chunks / k >= 8 && tiles < (int64_t)num_cu() && tiles * k <= 2 * (int64_t)num_cu()) best = k;      // (tiles >= CUs: 16^2 at N = 8 measured 222 -> 232 us with two shares)
    return best;
}

__global__ __launch_bounds__(256) void up2_sum_slices(const float* __restrict__ ws, float* __restrict__ y, int ksplit, int64_t slice4) {
    typedef float f4 __attribute__((ext_vector_type(4)));
    for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < slice4; i += (int64_t)gridDim.x * 256) {
        f4 v = ((const f4*)ws)[i];
        for (int z = 1; z < ksplit; z++) v += ((const f4*)ws)[(int64_t)z * slice4 + i];      // fixed order: deterministic
        ((f4*)y)[i] = v;
    }
}

inline int launch_up2_edges_only(const Up2Params& p, hipStream_t s) {      // (the geometry conv2d_up2x3.h serves: 32-wide tiles, 16-byte staging)
    return launch_up2_t<32, true>(p, s, true);
}

inline int launch_up2(const Up2Params& p, hipStream_t s) {
    static const bool vec_on = [] { const char* e = getenv("PG_UP2_VEC"); return e ? atoi(e) != 0 : true; }();      // A/B switch: 0 = four-byte halo staging everywhere
    if (p.W > 16 && vec_on && p.W % 4 == 0 && (((uintptr_t)p.x) & 15) == 0) return launch_up2_t<32, true>(p, s);
    if (p.W > 16) return launch_up2_t<32>(p, s);
    if (p.W > 8) return launch_up2_t<16>(p, s);
    return launch_up2_t<8>(p, s);
}

}  // namespace pgconv
