// The fp32 `up = 2` layer (conv2d_up2.h: stride-2 transposed 3x3 convolution, all four output parities in one pass) with its multiplies on the bf16 matrix
// pipe (round 6): every float32 operand is the exact sum of three bf16 values (truncation split, 8 + 8 + 8 significand bits) and a float32 product is
// evaluated as the six largest of the nine plane products on v_mfma_f32_32x32x16_bf16 with float32 accumulation -- float32-class results (the dropped
// products are below 2^-24 of the product) at 6/16 of the fp32 MFMA's issue time.  Reference: conv2d_resample.py:125-142 -> conv2d_gradfix.conv_transpose2d
// (stride 2), networks.py:73-94 for the modulation around it.
//
// Why this pays here and barely did in the Winograd kernel (conv2d_wino4.h, X3 form: +8 %): there a transformed value meets 64 couts once; here an input sample
// meets 9 taps x 32 couts, so ONE split per staged sample (a pass over the halo tile in LDS: 44 VALU per 8 values) feeds 9 x 6 MFMAs, and the weights
// are split once per weight version on the host side of the launch (pg_conv2d_up2x3_pack_weight).
//
//   * workgroup = 4 waves = ONE per CU (LDS ~116 KB, up to 512 registers per lane), tile = 32 couts x (8 x 32 positions), wave = 2 position rows x 4 parities
//     = 8 accumulators; K chunk = 16 channels = one MFMA K.
//   * per chunk: (1) split phase -- the raw float32 halo tile [16][9][40] (16-byte LDS-DMA words, as conv2d_up2.h's VEC staging) -> three bf16 planes
//     [plane][halo pixel 9 x 33][k-half][8 channels], one 16-byte word per (pixel, k-half) = the B operand of one MFMA lane; the input scale of the modulated
//     convolution is applied before the split (the same float32 product the fp32 kernel forms).  (2) barrier; the next chunk's halo and weights are requested.
//     (3) 108 MFMAs per wave (9 taps x 2 position rows x 6 plane products) from 27 weight and 18 input fragments (ds_read_b128), ordered so that consecutive
//     MFMAs hit different accumulators (a dependent 32x32x16 link costs ~87 cycles, conv2d_wino4.h).  (4) barrier.
//   * weights: [m-block 32][chunk][tap 9][plane 3][k-half][32 couts][8 channels] bf16 -- one contiguous 27 KB slab per (m-block, chunk), double buffered.
//   * epilogue, tile stream, XCD mapping: conv2d_up2.h's.  The last output column / row (conv2d_up2.h's edge tiles: 3 instead of 18 MFMAs per channel pair)
//     are made by conv2d_up2_edges.h (fp32 MFMA, K split over the waves of a workgroup).
// Serves the layers conv2d_up2.h stages with 16-byte words (W > 16, W % 4 == 0, 16-byte aligned x) whose Cin is a multiple of 16, without split-K.
#pragma once
#include "conv2d_up2.h"
#include "conv2d_up2_edges.h"

#ifndef UX_EXP
#define UX_EXP 0         // dev ablations (results wrong by design; tools/up2x3_variants.py): 1 no MFMAs, 2 no operand split, 4 no halo loads, 8 no weight DMA, 16 no output stores, 64 edge columns read contiguously, 128 no edge pass
#endif

namespace pgconv {

constexpr int UX_KC = 16;
constexpr int UX_IH = 9, UX_IW = 40, UX_PLANE = UX_IH * UX_IW;       // raw halo tile per channel (floats): rows q0-1 .. q0+7, columns r0-4 .. r0+35
constexpr int UX_NXW = UX_KC * UX_PLANE / 4;                          // 1440 sixteen-byte words per chunk
constexpr int UX_XPT = (UX_NXW + 255) / 256;                          // 6 DMA instructions per thread
constexpr int UX_RAW = (512 + 3 * 320) * 4;                           // floats: the four producer waves' regions (conv2d_up2x3: [row of the wave][channel][10 words])
constexpr int UX_PW = 33, UX_NPIX = UX_IH * UX_PW;                    // halo pixels the MFMAs read: columns r0-1 .. r0+31
constexpr int UX_PLB = UX_NPIX * 32;                                  // bytes of one plane: [pixel][k-half][8 x bf16]
constexpr int UX_WW = 9 * 3 * 2 * 32;                                 // 1728 sixteen-byte words of weights per (m-block, chunk)
constexpr int UX_WPT = (UX_WW + 255) / 256;                           // 7
constexpr int UX_WBUF = UX_WPT * 256 * 16;                            // bytes
constexpr int UX_TASKS = UX_NPIX * 2;                                 // split tasks per chunk: (pixel, k-half)
constexpr int UX_TPT = (UX_TASKS + 255) / 256;                        // 3 per thread

typedef __bf16 ux_bf16x8 __attribute__((ext_vector_type(8)));
typedef unsigned ux_u32x4 __attribute__((ext_vector_type(4)));

__device__ __forceinline__ void ux_split_pair(float va, float vb, unsigned& p0, unsigned& p1, unsigned& p2) {
    const unsigned ua = __float_as_uint(va), ub = __float_as_uint(vb);
    const float ra = va - __uint_as_float(ua & 0xffff0000u), rb = vb - __uint_as_float(ub & 0xffff0000u);
    const unsigned ura = __float_as_uint(ra), urb = __float_as_uint(rb);
    const float la = ra - __uint_as_float(ura & 0xffff0000u), lb = rb - __uint_as_float(urb & 0xffff0000u);
    p0 = __builtin_amdgcn_perm(ub, ua, 0x07060302u);
    p1 = __builtin_amdgcn_perm(urb, ura, 0x07060302u);
    p2 = __builtin_amdgcn_perm(__float_as_uint(lb), __float_as_uint(la), 0x07060302u);
}

template <bool MOD>
__global__ __launch_bounds__(512, 1) void conv2d_up2x3(Up2Params p, const unsigned char* __restrict__ wx3, float* __restrict__ xcol) {
    extern __shared__ __attribute__((aligned(16))) unsigned char smem_u8[];
    float* raw = (float*)smem_u8;                                      // the four producer waves' regions of 16-byte halo words
    unsigned char* planes = smem_u8 + UX_RAW * 4;                      // [2 buffers][3 planes][2 k-halves][297 pixels] x 16 B
    unsigned char* wbuf = planes + 2 * 3 * UX_PLB;                     // [2 buffers][9][3][2][32] x 16 B
    const int cin_loop = p.Cin;                                        // (a multiple of 16: checked by the launcher)
    const int nchunks = cin_loop / UX_KC;
    float* cs0 = (float*)(wbuf + 2 * UX_WBUF);                         // input scale of two consecutive tiles [2][cin_loop]
    float* ep0 = cs0 + 2 * cin_loop;                                   // output scale of two consecutive tiles [2][32]

    const int wave = __builtin_amdgcn_readfirstlane((int)threadIdx.x >> 6);
    const int lane = (int)threadIdx.x & 63;
    const unsigned smem_b = __builtin_amdgcn_readfirstlane(lds_offset((const float*)smem_u8));
    const int HW = p.H * p.W;
    const int total = p.total_tiles;
    const int q8 = total >> 3, r8 = total & 7;
    const int my_tiles = (total - (int)blockIdx.x + (int)gridDim.x - 1) / (int)gridDim.x;
    const int my_chunks = my_tiles * nchunks;

    auto decode = [&](int tile, int& n, int& q0, int& r0, int& m0) __attribute__((always_inline)) {
        const int xcd = tile & 7;
        int L = (xcd < r8 ? xcd * (q8 + 1) : r8 * (q8 + 1) + (xcd - r8) * q8) + (tile >> 3);
        const int mb = L % p.mblocks; L /= p.mblocks;
        const int tx = L % p.tilesX; L /= p.tilesX;
        const int ty = L % p.tilesY;
        n = L / p.tilesY;
        q0 = ty * 8; r0 = tx * 32; m0 = mb * U_BM;
    };

    if (wave >= 4) {
        // ------------------------------------------------------------ producer waves.  A producer wave requests exactly the halo rows it splits itself (wave pw: rows pw, pw + 4,
        // and row 8 for wave 0) as 16-byte LDS-DMA words into its own region of `raw` ([row of the wave][channel 16][word 10], padded to whole 64-word requests), so it
        // depends on no other wave's requests: one barrier per round.  Round g: wait for the own rows of chunk g (requested at the end of round g - 1: the latency
        // runs while this wave sits at the barrier), request the weights of chunk g, split the rows into planes[g & 1], request the rows of chunk g + 1, wait for the
        // weights only (vector memory returns in order: the row requests are younger), barrier.
        // (Requesting the samples straight into registers one round ahead -- no `raw` -- was tried: 4-byte loads are 4x the requests of 16-byte words, -105 us of
        //  vector-memory issue on the 128 -> 64 layer: slower.)
        const int pw = wave - 4;
        const int nrows = pw == 0 ? 3 : 2;
        const int nreq = pw == 0 ? 8 : 5;                              // 16-byte requests per lane and chunk: 480 (padded to 512) or 320 words
        const int reg_w = pw == 0 ? 0 : 512 + (pw - 1) * 320;          // first word of this wave's region
        const int t = 64 * pw + lane;
        int n = 0, q0 = 0, r0 = 0, m0 = 0;
        unsigned xoff[8];
        i32x4 xrsrc;
        int c_tile = blockIdx.x, c_chunk = 0, c_par = 0;               // the chunk being REQUESTED
        int s_chunk = 0, s_m0 = 0, s_par = 0;                          // the chunk being SPLIT
        auto request_rows = [&]() __attribute__((always_inline)) {
            if (c_chunk == 0) {
                decode(c_tile, n, q0, r0, m0);
                int ll = lane;
                asm volatile("" : "+v"(ll));
#pragma unroll
                for (int i = 0; i < 8; i++) {                          // (row of the wave, channel, word of the row); W % 4 == 0: a word is inside or outside the image as a whole
                    const int f = i * 64 + ll;
                    const int ri = f / 160, rem = f % 160;
                    const int c = rem / 10, wd = rem % 10;
                    const int row = ri == 0 ? pw : (ri == 1 ? pw + 4 : 8);
                    const int gy = q0 - 1 + row, gx = r0 - 4 + 4 * wd;
                    const bool ok = ri < nrows && gy >= 0 && gy < p.H && gx >= 0 && gx + 4 <= p.W;
                    xoff[i] = ok ? (unsigned)(c * HW + gy * p.W + gx) * 4u : 0x80000000u;
                }
                const uint64_t base = (uint64_t)(uintptr_t)(p.x + (int64_t)n * p.Cin * HW);
                xrsrc[0] = (int)(unsigned)base;
                xrsrc[1] = (int)(unsigned)(base >> 32) & 0xffff;
                xrsrc[2] = p.Cin * HW * 4;
                xrsrc[3] = 0x00020000;
                if (MOD && pw == 0) {                                  // this tile's input scales (published by the barriers of the rounds before its chunk 0 is split;
                    float* cs = cs0 + c_par * cin_loop;                //  first tile: the prologue barrier).  Nothing of this wave is in flight here: ld_opaque's wait is its own.
                    for (int c = lane; c < cin_loop; c += 64) cs[c] = ld_opaque(p.in_scale + (int64_t)n * p.Cin + c);
                }
            }
            i32x4 rs;                                                  // (the descriptor is carried across the chunk loop: back onto the scalar unit for the DMA's operand)
#pragma unroll
            for (int i = 0; i < 4; i++) rs[i] = __builtin_amdgcn_readfirstlane(xrsrc[i]);
            const unsigned xs_b = smem_b + (unsigned)reg_w * 16u;
            const int soff = c_chunk * UX_KC * HW * 4;
#pragma unroll
            for (int i = 0; i < 8; i++)
                if (i < nreq && !(UX_EXP & 4)) dma_dwordx4_buf(rs, xs_b + 1024u * i, xoff[i], soff);
            if (++c_chunk == nchunks) { c_chunk = 0; c_tile += gridDim.x; c_par ^= 1; }
        };
        auto request_weights = [&](int g) __attribute__((always_inline)) {
            const unsigned ws_b = smem_b + (unsigned)(UX_RAW * 4 + 2 * 3 * UX_PLB + (g & 1) * UX_WBUF) + (unsigned)(64 * pw) * 16u;
            const unsigned char* wsrc = wx3 + ((int64_t)(s_m0 / U_BM) * nchunks + s_chunk) * (int64_t)(UX_WW * 16);
#pragma unroll
            for (int i = 0; i < UX_WPT; i++) {
                int e = t + 256 * i;
                if (e >= UX_WW) e = UX_WW - 1;                         // clamp: the pad lanes copy a duplicate
                if (!(UX_EXP & 8)) dma_dwordx4((const float*)(wsrc + (int64_t)e * 16), ws_b + 4096u * i);
            }
        };
        auto split_rows = [&](int g) __attribute__((always_inline)) {
            unsigned char* pl_b = planes + (size_t)(g & 1) * 3 * UX_PLB;
            const float* cs = cs0 + s_par * cin_loop + s_chunk * UX_KC;
#pragma unroll
            for (int i = 0; i < 4; i++) {
                const int task = lane + 64 * i;                        // (row of the wave, k-half, column): 66 per row
                if (task < nrows * 66) {
                    const int ri = task / 66, rem = task - ri * 66;
                    const int kh = rem >= 33 ? 1 : 0, col = rem - 33 * kh;
                    const int row = ri == 0 ? pw : (ri == 1 ? pw + 4 : 8);
                    const float* rp = raw + (reg_w + ri * 160 + (8 * kh) * 10) * 4 + col + 3;     // halo column c sits at column c + 3 of the 40-float row
                    ux_u32x4 pl[3];
#pragma unroll
                    for (int j = 0; j < 4; j++) {
                        float va = rp[(2 * j) * 40], vb = rp[(2 * j + 1) * 40];
                        if (MOD) { va *= cs[8 * kh + 2 * j]; vb *= cs[8 * kh + 2 * j + 1]; }
                        unsigned p0, p1, p2;
                        if (UX_EXP & 2) { p0 = __float_as_uint(va); p1 = __float_as_uint(vb); p2 = p0; }
                        else ux_split_pair(va, vb, p0, p1, p2);
                        pl[0][j] = p0; pl[1][j] = p1; pl[2][j] = p2;
                    }
                    unsigned char* dst = pl_b + (size_t)(kh * UX_NPIX + row * UX_PW + col) * 16;      // [k-half][pixel]
                    *(ux_u32x4*)(dst) = pl[0];
                    *(ux_u32x4*)(dst + UX_PLB) = pl[1];
                    *(ux_u32x4*)(dst + 2 * UX_PLB) = pl[2];
                }
            }
            if (++s_chunk == nchunks) { s_chunk = 0; s_par ^= 1; }
        };
        // Barriers: s_barrier counts all eight waves; every wave executes my_chunks + 2 of them (one prologue barrier -- the first tile's scales --, then round g = 0 ..
        // my_chunks).  The barrier of round g: planes[g & 1] / wbuf[g & 1] of chunk g are complete (producers), chunk g - 1 has been multiplied (multiplying waves).
        request_rows();                                                // chunk 0 (and the first tile's scales)
        s_m0 = m0;
        asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)\n\ts_barrier" ::: "memory");
        for (int g = 0; g < my_chunks; g++) {
            if (s_chunk == 0 && g > 0) { int dn, dq, dr; decode((int)blockIdx.x + (g / nchunks) * (int)gridDim.x, dn, dq, dr, s_m0); }
            request_weights(g);
            // the rows of chunk g: older than the weight requests just issued
            asm volatile("s_waitcnt vmcnt(%0)" :: "n"(UX_WPT) : "memory");
            if (xcol && m0 == 0 && r0 + 32 >= p.W && lane < nrows * 16) {
                // the tiles that hold input column W - 1 leave it behind as a dense [N][Cin][H] array for the edge kernel (conv2d_up2_edges.h: gathered from x the
                // column costs it 10 ... 34 us per launch, one 128-byte line per sample at a stride of one image row).  Halo rows 1 .. 8 = image rows q0 .. q0 + 7.
                const int ri = lane >> 4, c = lane & 15;
                const int row = ri == 0 ? pw : (ri == 1 ? pw + 4 : 8);
                const int gy = q0 - 1 + row;
                if (row >= 1 && gy < p.H)
                    xcol[((int64_t)n * p.Cin + s_chunk * UX_KC + c) * p.H + gy] = raw[(reg_w + ri * 160 + c * 10) * 4 + (p.W + 3 - r0)];
            }
            split_rows(g);
            asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");         // (the raw tile has been read: its region may be overwritten)
            if (g + 1 < my_chunks) {
                request_rows();
                if (pw == 0) asm volatile("s_waitcnt vmcnt(8)\n\ts_barrier" ::: "memory");      // the weights are home; the 8 / 5 row requests stay in flight
                else asm volatile("s_waitcnt vmcnt(5)\n\ts_barrier" ::: "memory");
            } else {
                asm volatile("s_waitcnt vmcnt(0)\n\ts_barrier" ::: "memory");
            }
        }
        asm volatile("s_barrier" ::: "memory");                       // round my_chunks: nothing left to produce
        return;
    }

    // ---------------------------------------------------------------- multiplying waves
    const int half = lane >> 5, l31 = lane & 31;
    f32x16 acc[4][2];                                                  // [parity 2a + b][position row of the wave]
    int tile = blockIdx.x, par = 0, g = 0;
    unsigned long long st_mm = 0, st_bar = 0, st_ep = 0, st_t0 = 0, st_rounds = 0;      // (UX_EXP & 32: cycle totals of this wave)
    asm volatile("s_barrier" ::: "memory");                           // prologue barrier (the first tile's scales are published to the producers)
    asm volatile("s_barrier" ::: "memory");                           // barrier of round 0: the producers have made chunk 0
    for (int ti = 0; ti < my_tiles; ti++, tile += gridDim.x, par ^= 1) {
#pragma unroll
        for (int ph = 0; ph < 4; ph++)
#pragma unroll
            for (int nt = 0; nt < 2; nt++)
#pragma unroll
                for (int k = 0; k < 16; k++) acc[ph][nt][k] = 0.f;
        int e_n, e_q0, e_r0, e_m0;
        decode(tile, e_n, e_q0, e_r0, e_m0);
        float* ep_scale = ep0 + par * U_BM;
        for (int k = 0; k < nchunks; k++, g++) {
            const int wb = g & 1;
            if (UX_EXP & 32) st_t0 = __builtin_amdgcn_s_memtime();
            if (k + 1 == nchunks && wave == 0 && lane < U_BM) {
                const int co = e_m0 + lane;
                ep_scale[lane] = co < p.Cout ? (p.out_scale ? ld_opaque(p.out_scale + (int64_t)e_n * p.Cout + co) : 1.f) : 0.f;
            }
            {
                const unsigned char* ab = wbuf + (size_t)wb * UX_WBUF + (size_t)(half * 32 + l31) * 16;
                const unsigned char* bb = planes + (size_t)wb * 3 * UX_PLB + (size_t)(half * UX_NPIX + (2 * wave) * UX_PW + l31) * 16;
                ux_u32x4 B[3][2][3];                                   // B[halo row of the wave 0..2][column l31 + cc][plane]
#pragma unroll
                for (int pl = 0; pl < 3; pl++)
#pragma unroll
                    for (int hr = 0; hr < 3; hr++)
#pragma unroll
                        for (int cc = 0; cc < 2; cc++) B[hr][cc][pl] = *(const ux_u32x4*)(bb + (size_t)pl * UX_PLB + (size_t)(hr * UX_PW + cc) * 16);
                // tap id 3 ky + kx: taps 0, 2, 6, 8 -> parity (0,0) on x[q][r], x[q][r-1], x[q-1][r], x[q-1][r-1]; taps 1, 7 -> (0,1) on x[q][r], x[q-1][r];
                // taps 3, 5 -> (1,0) on x[q][r], x[q][r-1]; tap 4 -> (1,1) on x[q][r].  Steps of three taps; within a step and across steps the same accumulator
                // is at least four MFMAs apart (a dependent 32x32x16 link costs ~87 cycles).
                constexpr int TAPS[9] = {0, 1, 3, 2, 7, 5, 6, 4, 8};
                constexpr int PAR[9] = {0, 1, 0, 2, 3, 2, 0, 1, 0}, DY[9] = {1, 1, 1, 1, 1, 1, 0, 0, 0}, CC[9] = {1, 1, 0, 1, 1, 0, 1, 1, 0};
                constexpr int PA[6] = {2, 1, 1, 0, 0, 0}, PB[6] = {0, 1, 0, 2, 1, 0};        // small products first
                auto a_frag = [&](int pr, int ti3, int j) __attribute__((always_inline)) {
                    return *(const ux_u32x4*)(ab + (size_t)((TAPS[3 * ti3 + j] * 3 + PA[pr]) * 64) * 16);
                };
                ux_u32x4 Ar[3][3];                                     // ring of three steps' A fragments: step st in slot st % 3, requested two steps ahead
#pragma unroll
                for (int j = 0; j < 3; j++) { Ar[0][j] = a_frag(0, 0, j); Ar[1][j] = a_frag(0, 1, j); }
#pragma unroll
                for (int st = 0; st < 18; st++) {
                    const int pr = st / 3, ti3 = st % 3;
                    if (st + 2 < 18) {
#pragma unroll
                        for (int j = 0; j < 3; j++) Ar[(st + 2) % 3][j] = a_frag((st + 2) / 3, (st + 2) % 3, j);
                    }
                    __builtin_amdgcn_sched_barrier(0);
#pragma unroll
                    for (int j = 0; j < 3; j++) {
                        const int tp = TAPS[3 * ti3 + j];
#pragma unroll
                        for (int nt = 0; nt < 2; nt++) {
                            if (!(UX_EXP & 1) || Ar[st % 3][j][0] == 0x12345678u) acc[PAR[tp]][nt] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(ux_bf16x8, Ar[st % 3][j]),
                                                                                       __builtin_bit_cast(ux_bf16x8, B[nt + DY[tp]][CC[tp]][PB[pr]]), acc[PAR[tp]][nt], 0, 0, 0);
                            __builtin_amdgcn_sched_barrier(0);         // (keep the issue order: the same accumulator at least four MFMAs apart)
                        }
                    }
                }
            }
            if (UX_EXP & 32) { const unsigned long long t1 = __builtin_amdgcn_s_memtime(); st_mm += t1 - st_t0; st_t0 = t1; }
            asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");      // barrier g + 1: chunk g + 1 is complete; this chunk's buffers are free
            if (UX_EXP & 32) { st_bar += __builtin_amdgcn_s_memtime() - st_t0; st_rounds++; }
        }

        // ---- epilogue (conv2d_up2.h's): D col = lane & 31 = position r, row = (reg & 3) + 8 * (reg >> 2) + 4 * half = cout; a lane holds two adjacent output pixels
        if (UX_EXP & 32) st_t0 = __builtin_amdgcn_s_memtime();
        const int r = e_r0 + l31;
        const bool pair_ok = (p.ys[2] & 1) == 0 && (p.ys[1] & 1) == 0 && (p.ys[0] & 1) == 0 && (((uintptr_t)p.y) & 7) == 0 && p.ys[3] == 1;
        // one 64-bit base per tile and lane, 32-bit offsets inside the image (the launcher checks that one image of y stays below 2^31 floats): the address arithmetic
        // of the 64 stores per lane was a third of the epilogue
        float* y_lane = p.y + (int64_t)e_n * p.ys[0] + (int64_t)(2 * r) * p.ys[3];
        const int cs_ = (int)p.ys[1], rs_ = (int)p.ys[2], xs_ = (int)p.ys[3];
        float scv[16];
#pragma unroll
        for (int k = 0; k < 16; k++) scv[k] = ep_scale[(k & 3) + 8 * (k >> 2) + 4 * half];
#pragma unroll
        for (int nt = 0; nt < 2; nt++) {
            const int q = e_q0 + 2 * wave + nt;
#pragma unroll
            for (int a = 0; a < 2; a++) {
                const bool row_ok = a == 0 ? q <= p.H : q < p.H;
                const bool ok = row_ok && r < p.W;
                const int row_off = (2 * q + a) * rs_;
#pragma unroll
                for (int k = 0; k < 16; k++) {
                    const int rowc = (k & 3) + 8 * (k >> 2) + 4 * half;
                    const int co = e_m0 + rowc;
                    const float v0 = acc[2 * a][nt][k] * scv[k], v1 = acc[2 * a + 1][nt][k] * scv[k];
                    if (co < p.Cout && ok && (!(UX_EXP & 16) || v0 == 12345.678f)) {
                        float* dst = y_lane + (co * cs_ + row_off);
                        if (pair_ok) {
                            *(f32x2s*)dst = (f32x2s){v0, v1};
                        } else {
                            dst[0] = v0;
                            dst[xs_] = v1;
                        }
                    }
                }
            }
        }
        if (UX_EXP & 32) st_ep += __builtin_amdgcn_s_memtime() - st_t0;
    }
    if ((UX_EXP & 32) && blockIdx.x == 0 && lane == 0) {
        unsigned long long* o = (unsigned long long*)p.y + wave * 4;
        o[0] = st_mm; o[1] = st_bar; o[2] = st_ep; o[3] = st_rounds;
    }
}

// wp32 = pg_conv2d_pack_weight's [CinP][9][CoutP] float32 pack (weight gain / flip folded in) -> the X3 slabs
__global__ __launch_bounds__(256) void up2x3_pack_kernel(const float* __restrict__ wp32, unsigned short* __restrict__ out, int Cin, int CoutP32, int CoutP) {
    const int nchunks = Cin / UX_KC;
    const int64_t total = (int64_t)Cin * 9 * CoutP32;
    for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < total; i += (int64_t)gridDim.x * 256) {
        const int co = (int)(i % CoutP32), tap = (int)((i / CoutP32) % 9), ci = (int)(i / ((int64_t)CoutP32 * 9));
        const float v = co < CoutP ? wp32[((int64_t)ci * 9 + tap) * CoutP + co] : 0.f;
        const unsigned u0 = __float_as_uint(v) & 0xffff0000u;
        const float r = v - __uint_as_float(u0);
        const unsigned u1 = __float_as_uint(r) & 0xffff0000u;
        const unsigned u2 = __float_as_uint(r - __uint_as_float(u1)) & 0xffff0000u;
        const int mb = co >> 5, m = co & 31, chunk = ci >> 4, kh = (ci >> 3) & 1, j = ci & 7;
        const int64_t dst = ((((((int64_t)mb * nchunks + chunk) * 9 + tap) * 3) * 2 + kh) * 32 + m) * 8 + j;      // plane 0; planes are 2 * 32 * 8 words apart
        out[dst] = (unsigned short)(u0 >> 16);
        out[dst + 512] = (unsigned short)(u1 >> 16);
        out[dst + 1024] = (unsigned short)(u2 >> 16);
    }
}

inline size_t up2x3_lds_bytes(int Cin) { return (size_t)UX_RAW * 4 + 2 * 3 * (size_t)UX_PLB + 2 * (size_t)UX_WBUF + ((size_t)2 * Cin + 2 * U_BM) * 4; }

inline bool up2x3_serves(const Up2Params& p) {
    return p.W > 16 && p.W % 4 == 0 && (((uintptr_t)p.x) & 15) == 0 && p.Cin % UX_KC == 0 && p.Cin >= 2 * UX_KC && p.ksplit == 1 && up2x3_lds_bytes(p.Cin) <= 160 * 1024;
}

// Main tiles on the bf16 pipe, the edge tiles (last output column / row) by the fp32 kernel's edge pass.
inline int launch_up2x3(const Up2Params& p0, const void* wx3, float* xcol, hipStream_t s) {
#ifdef UX_NOXCOL                                             // (dev A/B: the edge kernel gathers the column itself)
    xcol = nullptr;
#endif
    if (!up2x3_serves(p0) || !wx3) return PG_ERR_UNSUPPORTED;
    Up2Params p = p0;
    p.tilesX = (p.W + 31) / 32;
    static const bool redge_on = [] { const char* e = getenv("PG_UP2_ROW_EDGE"); return !e || atoi(e) != 0; }();
    const bool redge = redge_on && p.H % 8 == 0;
    p.tilesY = redge ? p.H / 8 : (p.H + 1 + 7) / 8;
    p.mblocks = p.CoutP / U_BM;
    p.cpk = p.Cin / UX_KC;
    const int64_t tiles = (int64_t)p.N * p.tilesX * p.tilesY * p.mblocks;
    if (tiles > 0x7fffffffLL) return PG_ERR_TOO_LARGE;
    if ((int64_t)(p.Cout - 1) * p.ys[1] + (int64_t)(2 * p.H) * p.ys[2] + (int64_t)(2 * p.W) * p.ys[3] >= 0x7fffffffLL) return PG_ERR_UNSUPPORTED;      // 32-bit offsets inside one image of y
    p.total_tiles = (int)tiles;
    p.edge_tiles = p.redge_tiles = 0;
    const size_t lds = up2x3_lds_bytes(p.Cin);
    const int64_t blocks = tiles < (int64_t)num_cu() ? tiles : (int64_t)num_cu();
    if (p.in_scale) {
        static PerDeviceOnce a1;
        const hipError_t e = a1.run([] { return hipFuncSetAttribute((const void*)conv2d_up2x3<true>, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024); });
        if (e != hipSuccess) return (int)e;
        hipLaunchKernelGGL((conv2d_up2x3<true>), dim3((unsigned)blocks), dim3(512), lds, s, p, (const unsigned char*)wx3, xcol);
    } else {
        static PerDeviceOnce a0;
        const hipError_t e = a0.run([] { return hipFuncSetAttribute((const void*)conv2d_up2x3<false>, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024); });
        if (e != hipSuccess) return (int)e;
        hipLaunchKernelGGL((conv2d_up2x3<false>), dim3((unsigned)blocks), dim3(512), lds, s, p, (const unsigned char*)wx3, xcol);
    }
    int st = launch_status();
    if (st != PG_OK) return st;
    if (UX_EXP & 128) return PG_OK;                            // (dev ablation: no edge pass)
    static const bool own_edges = [] { const char* e = getenv("PG_UP2_EDGES"); return !e || atoi(e) != 0; }();      // A/B: 0 = conv2d_up2.h's edge pass
    return own_edges ? launch_up2_edges(p0, redge, xcol, s) : launch_up2_edges_only(p0, s);
}

}  // namespace pgconv
