// Winograd F(2x2, 3x3) fp32 convolution for gfx950 on v_mfma_f32_32x32x2_f32: the stride-1 3x3 layers of the
// synthesis network (74 % of the step) with 2.25x fewer matrix FLOPs than the direct implicit GEMM of conv2d_kernel.h.
//
//   Y = A^T [ (G g G^T) .* (B^T d B) ] A      per 2x2 output tile and 4x4 input patch, summed over input channels
//   => 16 independent GEMMs  M_xi[co][tile] = sum_ci U_xi[co][ci] * V_xi[ci][tile],   xi = 4a + b
// (Lavin & Gray's matrices for cross-correlation, the operation F.conv2d / conv2d_gradfix.py:38 computes.)
//
// Mapping to the hardware
//   * workgroup = 512 threads = 8 waves; tile = 64 couts x (2 output rows x 64 output cols = 32 Winograd tiles);
//     wave w owns row-transform index a = w >> 1 with all four column indices b, for the 32-cout M-tile mt = w & 1:
//     4 accumulators of 16 VGPRs; MFMA lanes = the 32 tiles, K = input-channel pairs.
//   * B operand (V = B^T d B) is never materialised.  The raw 4-row halo tile sits in LDS in its natural column order
//     (rows stored 0,2,1,3 so that every a uses two ADJACENT LDS rows); a lane fetches its 4x2 samples with four 8-byte
//     LDS reads (conflict-free: lane t starts at column 2t) and builds V[a][0..3] with ~12 VALU ops on the idle vector
//     pipe.  The tile arrives by 16-byte LDS-DMA (`buffer_load_dwordx4 ... lds`; zero padding = the buffer range check
//     on a per-lane sentinel offset), double buffered, one barrier per 16 channels.  Ragged widths (W % 4 != 0) use the
//     4-byte DMA form of the same map.
//   * A operand (U = G g G^T, pre-transformed by pg_conv2d_winograd_pack_weight) is used by exactly one wave, so it is
//     NOT staged in LDS: the packed order [a][co/32][ci/2][ci&1][co&31][b] makes the four b values of a lane one 16-byte
//     word and every wave-instruction a contiguous 1 KB; each wave streams its slice from L2 through a 4-pair ring.
//   * vector-memory instructions are the scarce resource of this kernel (the texture-address path takes ~16 cycles per
//     wave-instruction whatever its width), hence 16-byte requests everywhere: 11 per wave per 16 channels.
//   * all global loads of the K loop are inline asm with hand-counted s_waitcnt (the compiler cannot count asm loads; its
//     own counts would make every chunk wait for the halo tile it has just requested).
//   * inverse transform: the column half (over b) is done in registers; the row half (over a) crosses waves, so 16 couts
//     per round go through a double-buffered LDS exchange (one barrier per round, 4 rounds); 512 threads then finish two
//     adjacent pixels each with the fused epilogue (demodulation, noise, bias, activation, gain, clamp, residual, or
//     the SPADE combine) and 8-byte stores.
//   * persistent tile stream with cross-tile prefetch, XCD-aware tile order: as conv2d_kernel.h.
// Numerics: exact fp32 products/sums (same MFMA), different summation order; measured max error ~2e-6 of the output
// scale, the same level as the direct kernel (tests/test_hip_parity.py::test_conv2d_winograd_*).
#pragma once
#include <cstdlib>
#include "conv2d_kernel.h"

#ifndef WINO_PK
#define WINO_PK 0        // 1 = the round-1 packed-fp32 form of the K-loop transform (A/B switch)
#endif
#ifndef WINO_PRIO
#define WINO_PRIO 1      // 1 = wave priority 2 for each pair's MFMA group, 1 for the rest of the K loop, 0 in the tail: a workgroup that multiplies
                         // wins issue arbitration over the co-resident workgroup's inverse transform / stores, whose VALU, LDS and store
                         // instructions then fill the gaps instead of delaying MFMAs (measured 1.5-2.7 % per launch; 0 = off, A/B)
#endif
#ifndef WINO_EXP
#define WINO_EXP 0       // dev ablations (tools/wino_variants.py; results wrong by design): 1 no U loads, 2 no LDS operand reads, 4 no tail,
                         // 8 no halo DMA, 128 no chunk barrier, 256 no transform VALU, 512 eight extra independent VALU ops per channel pair,
                         // 1024 no LDS exchange in the tail, 2048 no output stores, 32768 no wave priorities
#endif

namespace pgconv {

constexpr int W_KC = 16;                     // input channels per LDS chunk
constexpr int W_LROW = 72;                   // LDS halo row = global columns [ox0 - 4, ox0 + 68): 18 aligned 16-byte words
constexpr int W_CHF = 4 * W_LROW;            // floats per channel (4 halo rows)
constexpr int W_NX = W_KC * W_CHF;           // 4608 staged floats per chunk = 9 x 512
constexpr int W_BUFS = W_NX + 4;             // buffer stride: +1 float when the tile is shifted to make column pairs 8-byte aligned
constexpr int W_RING = 4;                    // U pairs in flight per wave
constexpr int W_EXCH = 4 * 2 * 16 * 32;      // exchange floats per round: [a][q][16 couts][32 tiles]

typedef float f32x2 __attribute__((ext_vector_type(2)));


// MODE: 0 = plain input, 1 = per-(n, channel) input scale (modulated convolution), 2 = scale + pre-activation (SPADE convs).
// VEC: 16-byte halo DMA (W % 4 == 0 and a 16-byte aligned x), issued by waves 0-3 only; otherwise dwords from all waves.
// Why one half issues the whole DMA: a wave's U words requested after its DMA cannot return before it (loads return in
// order), so a DMA that misses to HBM stalls its issuer 4 pairs later.  With all waves issuing, the whole workgroup stalls
// together; with waves 0-3 issuing, waves 4-7 (one per SIMD) keep the matrix pipe busy through that window.
// NDMA = requests per issuing thread per chunk.
template <int MODE, bool VEC>
__global__ __launch_bounds__(512, 4) void conv2d_wino(ConvParams p) {
    constexpr bool XF = MODE == 2;
    constexpr int NDMA = VEC ? 6 : 9;
    extern __shared__ __attribute__((aligned(16))) float smem[];
    const int cin_loop = ((p.Cin + W_KC - 1) / W_KC) * W_KC;
    const int nchunks = cin_loop / W_KC;
    float* ex0 = smem + 2 * W_BUFS;              // inverse-transform exchange, double buffered [2][W_EXCH]
    float* cs0 = ex0 + 2 * W_EXCH;               // prologue scale of two consecutive tiles [2][cin_loop]
    float* ep0 = cs0 + 2 * cin_loop;             // epilogue scale / bias of two consecutive tiles [2][64 + 64]
    unsigned* gm0 = (unsigned*)(ep0 + 256);      // VEC: the gather map of an interior tile, [NDMA][256 issuing threads] byte offsets

    const int t = threadIdx.x, lane = t & 63;
    const int wave = __builtin_amdgcn_readfirstlane(t >> 6);
    const unsigned smem_b = __builtin_amdgcn_readfirstlane(lds_offset(smem));
    const int half = lane >> 5, l31 = lane & 31;
    const int HW = p.H * p.W;
    const int total = p.total_tiles;
    const int q8 = total >> 3, r8 = total & 7;

    // This wave's transforms.  LDS rows hold halo rows 0,2,1,3; B^T row a = (LDS row rp) + s1 * (LDS row rp + 1):
    //   a=0: d0 - d2 (rp 0)   a=1: d2 + d1 (rp 1)   a=2: d2 - d1 (rp 1)   a=3: d1 - d3 (rp 2)
    const int ta = wave >> 1, mt = wave & 1;
    const int rp = ta == 0 ? 0 : (ta == 3 ? 2 : 1);
    const float s1 = ta == 1 ? 1.f : -1.f;
    // LDS column of global column gx is (gx - ox0 + 4) + sh; patch column j of lane t's tile is gx = ox0 + 2t + j - pad_x,
    // i.e. LDS column 2t + j + (4 - pad_x) + sh.  sh = (4 - pad_x) & 1 makes j = 0 even: the pairs (j0,j1), (j2,j3) are
    // 8-byte aligned LDS words.
    const int P = 4 - p.pad_x, sh = P & 1;
    const int b_lane = half * W_CHF + rp * W_LROW + 2 * l31 + P + sh;        // floats from the buffer start: this lane's (row rp, j = 0)

    int n = 0, oy0 = 0, ox0 = 0, m0 = 0;
    unsigned xoff[NDMA];
    i32x4 xrsrc;

    // Gather map of a tile whose halo lies inside the image, relative to the tile's first halo sample: it does not depend on
    // the tile, so it is computed once (the per-tile version costs every issuing wave ~150 VALU instructions -- matrix-pipe
    // time -- per tile) and kept in LDS; interior tiles read it back and move the tile origin into the descriptor base.
    if (VEC && wave < 4) {
#pragma unroll
        for (int i = 0; i < NDMA; i++) {
            const bool wide = i < 4;
            const int f = wide ? 4 * (t + 256 * i) : 4096 + t + 256 * (i - 4);
            const int c = f / W_CHF, rem = f % W_CHF;
            const int rl = rem / W_LROW, lc = rem % W_LROW;
            const int hr = rl == 1 ? 2 : (rl == 2 ? 1 : rl);
            gm0[i * 256 + t] = (unsigned)(c * HW + hr * p.W + lc) * 4u;
        }
    }                                            // (visible to its own thread only: no barrier needed)

    auto prep_tile = [&](int tile, float* cs) {
        const int xcd = tile & 7;
        int L = (xcd < r8 ? xcd * (q8 + 1) : r8 * (q8 + 1) + (xcd - r8) * q8) + (tile >> 3);
        const int mb = L % p.mblocks; L /= p.mblocks;
        const int tx = L % p.tilesX; L /= p.tilesX;
        const int ty = L % p.tilesY;
        n = L / p.tilesY;
        oy0 = ty * 2; ox0 = tx * 64; m0 = mb * 64;
        const float* in_scale = p.f.in_scale ? p.f.in_scale + (int64_t)n * p.Cin : nullptr;
        if (MODE != 0)                                 // the plain variant never reads the scale
            for (int c = t; c < cin_loop; c += 512) cs[c] = ((in_scale && c < p.Cin) ? ld_opaque(in_scale + c) : 1.f) * p.f.in_gain;     // host: in_gain defaults to 1
        int tt = t;
        asm volatile("" : "+v"(tt));                 // keep the index maths inside the tile loop (see conv2d_kernel.h)
        const int gy0 = oy0 - p.pad_y, gx0 = ox0 - 4;
        const bool interior = VEC && p.wino_gmap && gy0 >= 0 && gy0 + 4 <= p.H && gx0 >= 0 && gx0 + W_LROW <= p.W;      // wave-uniform
        // gather map: element f of the buffer = (channel f / 288, LDS row (f % 288) / 72, LDS column f % 72)
        if (interior) {
            if (wave < 4) {
#pragma unroll
                for (int i = 0; i < NDMA; i++) xoff[i] = gm0[i * 256 + tt];
            }
        } else if (!VEC || wave < 4) {
#pragma unroll
            for (int i = 0; i < NDMA; i++) {
                const bool wide = VEC && i < 4;      // VEC: four 16-byte requests (floats 4e .. 4e+3, e < 1024) + two dwords (floats 4096 ..)
                const int f = wide ? 4 * (tt + 256 * i) : (VEC ? 4096 + tt + 256 * (i - 4) : tt + 512 * i);
                const int c = f / W_CHF, rem = f % W_CHF;
                const int rl = rem / W_LROW, lc = rem % W_LROW;
                const int hr = rl == 1 ? 2 : (rl == 2 ? 1 : rl);             // LDS row -> halo row
                const int gy = oy0 - p.pad_y + hr, gx = ox0 - 4 + lc;
                const bool ok = gy >= 0 && gy < p.H && gx >= 0 && gx + (wide ? 4 : 1) <= p.W;
                xoff[i] = ok ? (unsigned)(c * HW + gy * p.W + gx) * 4u : 0x80000000u;
            }
        }
        const int shift = interior ? (gy0 * p.W + gx0) * 4 : 0;        // interior: offsets are relative to the tile's first halo sample
        const uint64_t base = (uint64_t)(uintptr_t)(p.x + (int64_t)n * p.Cin * HW) + (uint64_t)(int64_t)shift;
        xrsrc[0] = __builtin_amdgcn_readfirstlane((int)(unsigned)base);
        xrsrc[1] = __builtin_amdgcn_readfirstlane((int)(unsigned)(base >> 32) & 0xffff);
        xrsrc[2] = p.Cin * HW * 4 - shift;                               // same absolute end: channels beyond Cin still read as zero
        xrsrc[3] = 0x00020000;
    };

    auto issue_chunk = [&](int c0, int buf) {
        const unsigned xs_b = smem_b + (unsigned)(buf * W_BUFS + sh) * 4u;
        const int soff = c0 * HW * 4;
#if !(WINO_EXP & 8)
        if (VEC) {
            if (wave < 4) {
#pragma unroll
                for (int i = 0; i < 4; i++) dma_dwordx4_buf(xrsrc, xs_b + (unsigned)(256 * i + 64 * wave) * 16u, xoff[i], soff);
#pragma unroll
                for (int i = 4; i < 6; i++) dma_dword(xrsrc, xs_b + (unsigned)(4096 + 256 * (i - 4) + 64 * wave) * 4u, xoff[i], soff);
            }
        } else {
#pragma unroll
            for (int i = 0; i < NDMA; i++) dma_dword(xrsrc, xs_b + (unsigned)(512 * i + 64 * wave) * 4u, xoff[i], soff);
        }
#endif
    };

    const float in_slope = act_slope(p.f.in_act, p.f.in_alpha);
    const float in_cl = p.f.in_clamp >= 0.f ? p.f.in_clamp : __builtin_inff();

    f32x16 acc[4];                                   // [b]

    // A-operand stream: one 16-byte word per channel pair (the four b values), 1 KB per wave-instruction, advanced through
    // a ring of W_RING pairs that runs across tiles (the last pairs of a tile already fetch the first pairs of the next one).
    // The loads are inline asm, waited for by hand.  Loads return in order, so at the use of pair pp of a chunk the queue
    // holds, after the needed word, the 3 younger ring words plus -- for pp < 4, whose words were requested before this
    // chunk's halo DMA -- the NDMA DMA requests: vmcnt(3 + NDMA), else vmcnt(3); the latter also guarantees that the DMA
    // has landed before the chunk's barrier.  Other queue entries (epilogue stores, residual loads) only make it stricter;
    // the first four pairs of a tile need no wait at all (their words are waited for before the previous epilogue's stores).
    const int NP = cin_loop / 2;                                             // pairs per tile
    unsigned pa;                                                             // byte offset into the packed U (< 2^31: checked by the host)
    auto a_reset = [&]() { pa = (unsigned)((ta * (p.CoutP / 32) + (m0 >> 5) + mt) * NP) * 1024u + (unsigned)(half * 512 + l31 * 16); };
    auto load_a = [&](f32x4& dst, int slot) {                // slot = pair & 3: immediate offset; the base moves every 4th pair
        if (slot == 0) asm volatile("global_load_dwordx4 %0, %1, %2" : "=v"(dst) : "v"(pa), "s"(p.wp));
        else if (slot == 1) asm volatile("global_load_dwordx4 %0, %1, %2 offset:1024" : "=v"(dst) : "v"(pa), "s"(p.wp));
        else if (slot == 2) asm volatile("global_load_dwordx4 %0, %1, %2 offset:2048" : "=v"(dst) : "v"(pa), "s"(p.wp));
        else { asm volatile("global_load_dwordx4 %0, %1, %2 offset:3072" : "=v"(dst) : "v"(pa), "s"(p.wp)); pa += 4096u; }
    };
    const bool issues_dma = !VEC || wave < 4;
    auto wait_a = [&](f32x4& g, bool dma_younger, bool skip) {
        // the s_waitcnt itself carries no register operand (inside a branch it would make the compiler merge register copies);
        // the empty asm after it ties the ring word to the wait: volatile asms keep their order, every use of g comes after
        if (!skip) {
            if (dma_younger && issues_dma) asm volatile("s_waitcnt vmcnt(%0)" :: "n"(W_RING - 1 + NDMA));
            else asm volatile("s_waitcnt vmcnt(%0)" :: "n"(W_RING - 1));
        }
        asm volatile("" : "+v"(g));
    };

    // The tail's uniforms (pointers, strides, activation constants) are re-read from the kernel-argument segment where they
    // are used: kept live across the K loop they overflow the SGPR file, and every spilled SGPR comes back through a
    // v_readlane -- a VALU instruction, i.e. matrix-pipe time.  The opaque asm keeps the compiler from hoisting the loads.
    typedef const __attribute__((address_space(4))) ConvParams* kernarg_t;
    auto fresh_args = [&]() {
        kernarg_t a = (kernarg_t)__builtin_amdgcn_kernarg_segment_ptr();
        asm volatile("" : "+s"(a));
        return a;
    };

    int tile = blockIdx.x;
    int par = 0, g = 0;
    prep_tile(tile, cs0);
    f32x4 a_ring[W_RING];
    a_reset();
#pragma unroll
    for (int d = 0; d < W_RING; d++) load_a(a_ring[d], d);
    issue_chunk(0, 0);
    dma_wait_all();
    __syncthreads();
    while (true) {
#pragma unroll
        for (int b = 0; b < 4; b++)
#pragma unroll
            for (int k = 0; k < 16; k++) acc[b][k] = 0.f;

        int e_n = n, e_oy0 = oy0, e_ox0 = ox0, e_m0 = m0;
        bool has_next = false;
        int next = tile;
        const float* cs_cur = cs0 + par * cin_loop;
        float* ep_scale = ep0 + par * 128;           // per tile parity: the next tile's constants are written while slow
        float* ep_bias = ep_scale + 64;              // waves may still be in this tile's epilogue
#if WINO_PRIO && !(WINO_EXP & 32768)
        __builtin_amdgcn_s_setprio(1);
#endif
#pragma unroll 1                                     // unrolled by two, the ragged plain variant spills a VGPR whose reload brings a vmcnt(0) into the loop
        for (int k = 0; k < nchunks; k++, g++) {
            const int buf = g & 1;
            if (k + 1 < nchunks) {
                issue_chunk((k + 1) * W_KC, buf ^ 1);
            } else {
                e_n = n; e_oy0 = oy0; e_ox0 = ox0; e_m0 = m0;
                const auto& q = *fresh_args();
                if (q.f.spade_x) {                          // SPADE mode: per-(n, channel) mean / rstd of the normalised tensor
                    if (t < 32) {
                        const int ch = (e_m0 >> 1) + t;     // this tile's 32 output channels
                        ep_scale[t] = ld_opaque(q.f.spade_mean + e_n * (q.Cout >> 1) + ch);      // opaque: see conv2d_kernel.h
                        ep_bias[t] = ld_opaque(q.f.spade_rstd + e_n * (q.Cout >> 1) + ch);
                    }
                } else if (t < 64) {
                    const int co = e_m0 + t;
                    const bool ok = co < q.Cout;
                    const int cc = ok ? co : 0;
                    const float sc = q.f.out_scale ? ld_opaque(q.f.out_scale + (int64_t)e_n * q.Cout + cc) : 1.f;
                    const float bi = q.f.bias ? ld_opaque(q.f.bias + cc) : 0.f;
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
            // ---- multiply this chunk: 8 channel pairs x 4 positions per wave.  The samples (and the prologue scale) of
            // pair pp + 1 are requested from LDS before pair pp's MFMAs are issued.
            typedef const __attribute__((address_space(3))) float* lds_cptr;
            typedef const __attribute__((address_space(3))) f32x2* lds_cptr2;
            lds_cptr xc = (lds_cptr)smem + buf * W_BUFS + b_lane;            // this lane's (row rp, j = 0) of channel `half`
            const float* csb = cs_cur + k * W_KC + half;
            constexpr int PD = 1;                                            // LDS operand requests run PD pairs ahead (2 measured no faster)
            f32x2 bq[PD + 1][4];                                             // [row rp | rp + 1][columns (0,1) | (2,3)]
            float bs[PD + 1];
            auto read_b = [&](int pp, f32x2 (&dst)[4], float& sc) {
                // one VGPR base per pair (advanced by a single add) so that the four 8-byte reads fit ds_read2_b64's offsets
                asm volatile("" : "+v"(xc));
#if WINO_EXP & 2
                dst[0] = (f32x2){1.f, 2.f}; dst[1] = (f32x2){3.f, 4.f}; dst[2] = (f32x2){5.f, 6.f}; dst[3] = (f32x2){7.f, 8.f};
#else
                dst[0] = *(lds_cptr2)xc; dst[1] = *(lds_cptr2)(xc + 2);
                dst[2] = *(lds_cptr2)(xc + W_LROW); dst[3] = *(lds_cptr2)(xc + W_LROW + 2);
#endif
                if (MODE != 0) sc = csb[2 * pp];
                xc += 2 * W_CHF;
            };
#pragma unroll
            for (int d = 0; d < PD; d++) read_b(d, bq[d], bs[d]);
#pragma unroll
            for (int pp = 0; pp < W_KC / 2; pp++) {
                if (pp + PD < W_KC / 2) read_b(pp + PD, bq[(pp + PD) % (PD + 1)], bs[(pp + PD) % (PD + 1)]);
                __builtin_amdgcn_sched_barrier(0);                           // the requests go out BEFORE this pair's MFMAs
                // row transform q = (row rp) + s1 * (row rp + 1), scaled by the prologue scale.  Scalar fp32 on purpose: packed
                // fp32 VALU (v_pk_fma_f32 / v_pk_add_f32) is an anti-lever beside MFMAs on gfx950 (MI355X_MICROARCH.md price table);
                // this translation unit is also built with -fno-slp-vectorize so that the compiler does not re-pack it.
                const float sc = bs[pp % (PD + 1)];
                const f32x2 (&B)[4] = bq[pp % (PD + 1)];
                float q0, q1, q2, q3;
#if WINO_PK
                f32x2 q01, q23;
                if (XF) {
                    f32x2 d[4];
#pragma unroll
                    for (int i = 0; i < 4; i++)
#pragma unroll
                        for (int e = 0; e < 2; e++) {
                            const float v = B[i][e] * sc;
                            d[i][e] = __builtin_amdgcn_fmed3f(fmaxf(v, v * in_slope), -in_cl, in_cl);
                        }
                    q01 = d[2] * s1 + d[0]; q23 = d[3] * s1 + d[1];
                } else if (MODE == 1) {
                    const float ss = sc * s1;
                    q01 = B[2] * ss + B[0] * sc; q23 = B[3] * ss + B[1] * sc;
                } else {
                    q01 = B[2] * s1 + B[0]; q23 = B[3] * s1 + B[1];
                }
                q0 = q01[0]; q1 = q01[1]; q2 = q23[0]; q3 = q23[1];
#else
                if (XF) {
                    float d[4][2];
#pragma unroll
                    for (int i = 0; i < 4; i++)
#pragma unroll
                        for (int e = 0; e < 2; e++) {
                            const float v = B[i][e] * sc;
                            d[i][e] = __builtin_amdgcn_fmed3f(fmaxf(v, v * in_slope), -in_cl, in_cl);
                        }
                    q0 = fmaf(d[2][0], s1, d[0][0]); q1 = fmaf(d[2][1], s1, d[0][1]);
                    q2 = fmaf(d[3][0], s1, d[1][0]); q3 = fmaf(d[3][1], s1, d[1][1]);
                } else if (MODE == 1) {
                    q0 = fmaf(B[2][0], s1, B[0][0]) * sc; q1 = fmaf(B[2][1], s1, B[0][1]) * sc;
                    q2 = fmaf(B[3][0], s1, B[1][0]) * sc; q3 = fmaf(B[3][1], s1, B[1][1]) * sc;
                } else {
                    q0 = fmaf(B[2][0], s1, B[0][0]); q1 = fmaf(B[2][1], s1, B[0][1]);
                    q2 = fmaf(B[3][0], s1, B[1][0]); q3 = fmaf(B[3][1], s1, B[1][1]);
                }
#endif
#if WINO_EXP & 512
                { static_assert(true, ""); float dmy = sc;                        // timing probe: 8 independent VALU ops per pair
                  asm volatile("v_add_f32 %0, %0, %0\n\tv_add_f32 %0, %0, %0\n\tv_add_f32 %0, %0, %0\n\tv_add_f32 %0, %0, %0\n\t"
                               "v_add_f32 %0, %0, %0\n\tv_add_f32 %0, %0, %0\n\tv_add_f32 %0, %0, %0\n\tv_add_f32 %0, %0, %0" : "+v"(dmy)); }
#endif
#if WINO_EXP & 256
                const float v0 = bq[pp % (PD + 1)][0][0], v1 = bq[pp % (PD + 1)][0][1], v2 = bq[pp % (PD + 1)][1][0], v3 = bq[pp % (PD + 1)][1][1];   // timing probe: no transform
#else
                const float v0 = q0 - q2, v1 = q1 + q2, v2 = q2 - q1, v3 = q1 - q3;      // B^T d B, row a
#endif
                wait_a(a_ring[pp % W_RING], pp < W_RING, pp < W_RING && k == 0);   // first pairs of a tile: landed before the previous epilogue's stores
#if WINO_PRIO && !(WINO_EXP & 32768)
                __builtin_amdgcn_s_setprio(2);
#endif
                acc[0] = __builtin_amdgcn_mfma_f32_32x32x2f32(a_ring[pp % W_RING][0], v0, acc[0], 0, 0, 0);
                acc[1] = __builtin_amdgcn_mfma_f32_32x32x2f32(a_ring[pp % W_RING][1], v1, acc[1], 0, 0, 0);
                acc[2] = __builtin_amdgcn_mfma_f32_32x32x2f32(a_ring[pp % W_RING][2], v2, acc[2], 0, 0, 0);
                acc[3] = __builtin_amdgcn_mfma_f32_32x32x2f32(a_ring[pp % W_RING][3], v3, acc[3], 0, 0, 0);
#if WINO_PRIO && !(WINO_EXP & 32768)
                __builtin_amdgcn_s_setprio(1);
#endif
                if (pp == W_KC / 2 - W_RING && k + 1 == nchunks) a_reset();  // from here on: the next tile's first pairs
#if !(WINO_EXP & 1)
                load_a(a_ring[pp % W_RING], pp % W_RING);                    // refill the slot W_RING pairs ahead
#endif
                __builtin_amdgcn_sched_barrier(0);
            }
            // No vmcnt wait here: pair 4 already waited for a U word requested AFTER this chunk's halo DMA, so the DMA has
            // landed.  (A vmcnt(0) would also wait for the U refills issued a moment ago: one exposed L2 round trip per chunk.)
#if !(WINO_EXP & 128)
            __syncthreads();
#endif
        }

#if WINO_PRIO && !(WINO_EXP & 32768)
        __builtin_amdgcn_s_setprio(0);
#endif
        // ---- inverse transform + fused epilogue.  Column half in registers: Y'[a][q] = sum_b M[a][b] At[q][b]; the row half
        // (over a) crosses waves: 16 couts per round (8 of each M-tile) through the exchange buffer, one barrier per round.
#if WINO_EXP & 4
        { float sm = 0.f;
          for (int b = 0; b < 4; b++) for (int k = 0; k < 16; k++) sm += acc[b][k];
          if (sm == 12345.678f) p.y[t] = sm; }
        if (!has_next) break;
        tile = next; par ^= 1;
        continue;
#endif
        // Every VALU instruction here costs matrix-pipe time (they do not overlap on this hardware), so the tail is written for
        // instruction count: 32-bit byte offsets against uniform bases, packed fp32 math on the pixel pair, a predicate-free path
        // for interior tiles, and the activation chain skipped when it is the identity.
        const auto& q = *fresh_args();
        const int c_l = t >> 6, prow = (t >> 5) & 1, tcol = t & 31;          // this thread's outputs: cout (of 8), row of the 2x2, tile
        const int oy = e_oy0 + prow, ox = e_ox0 + 2 * tcol;
        const bool spade = q.f.spade_x != nullptr;
        const float gain = q.f.gain, slope = act_slope(q.f.act, q.f.alpha);
        const float cl = q.f.clamp >= 0.f ? q.f.clamp : __builtin_inff();
        const bool vec_store = q.ys[3] == 1 && ((q.ys[0] | q.ys[1] | q.ys[2]) & 1) == 0 && (((uintptr_t)q.y) & 7) == 0 && (q.OW & 1) == 0;
        const bool full = vec_store && e_oy0 + 2 <= q.OH && e_ox0 + 64 <= q.OW && e_m0 + 64 <= q.Cout;      // wave-uniform
        const bool row_ok = oy < q.OH;
        const int oyc = row_ok ? oy : q.OH - 1;
        const int ox0c = ox < q.OW ? ox : q.OW - 1, ox1c = ox + 1 < q.OW ? ox + 1 : q.OW - 1;
        const bool ok0 = row_ok && ox < q.OW, ok1 = row_ok && ox + 1 < q.OW;
        const unsigned cstride_b = (unsigned)q.ys[1] * 4u;
        const unsigned pix0_b = (unsigned)((int64_t)e_n * q.ys[0] + (int64_t)oyc * q.ys[2] + (int64_t)ox0c * q.ys[3]) * 4u;
        const unsigned pix1_b = (unsigned)((int64_t)e_n * q.ys[0] + (int64_t)oyc * q.ys[2] + (int64_t)ox1c * q.ys[3]) * 4u;
        const float sg = prow ? -1.f : 1.f;                                   // A^T row: prow 0 -> (+ + +), prow 1 -> (+ - -)
        const bool plain_tail = slope == 1.f && gain == 1.f && q.f.clamp < 0.f;                                // wave-uniform
        f32x2 nz = {0.f, 0.f};
        if (q.f.noise) {
            const float* nzp = q.f.noise + (int)(e_n * q.f.noise_batch_stride) + oyc * q.OW;
            nz[0] = nzp[ox0c] * q.f.noise_gain; nz[1] = nzp[ox1c] * q.f.noise_gain;
        }
        auto ld = [&](const float* base, unsigned off_b) { return *(const float*)((const char*)base + off_b); };
        auto ld2 = [&](const float* base, unsigned off_b) {                   // the pixel pair of one channel
            f32x2 r;
            if (full) r = *(const f32x2*)((const char*)base + off_b);
            else { r[0] = ld(base, off_b); r[1] = ld(base, off_b + (pix1_b - pix0_b)); }
            return r;
        };
        auto store2 = [&](unsigned off_b, f32x2 v, bool chan_ok) {
            if (WINO_EXP & 2048) {
                if (v[0] == 12345.678f) *(f32x2*)((char*)q.y + off_b) = v;
            } else if (full) {
                *(f32x2*)((char*)q.y + off_b) = v;
            } else if (vec_store) {
                if (ok0 && chan_ok) *(f32x2*)((char*)q.y + off_b) = v;
            } else {
                if (ok0 && chan_ok) *(float*)((char*)q.y + off_b) = v[0];
                if (ok1 && chan_ok) *(float*)((char*)q.y + off_b + (pix1_b - pix0_b)) = v[1];
            }
        };
        const float* exr0 = ex0 + (2 * prow) * 16 * 32 + c_l * 32 + tcol;
#pragma unroll
        for (int rnd = 0; rnd < 4; rnd++) {
            // couts 8 rnd + 4 half + j of this wave's M-tile live in accumulator registers 4 rnd + j
            float* ex = ex0 + (rnd & 1) * W_EXCH;                            // [a][q][16 couts: mt * 8 + 4 half + j][32 tiles]
#pragma unroll
            for (int j = 0; j < 4; j++) {
                const float m0v = acc[0][4 * rnd + j], m1v = acc[1][4 * rnd + j], m2v = acc[2][4 * rnd + j], m3v = acc[3][4 * rnd + j];
                float* dst = ex + ((2 * ta) * 16 + mt * 8 + 4 * half + j) * 32 + l31;
                const float m12 = m1v + m2v;
#if WINO_EXP & 1024
                if (m0v + m12 == 12345.678f) dst[0] = (m1v - m2v) - m3v;
#else
                dst[0] = m0v + m12;                                          // q = 0:  M0 + M1 + M2
                dst[16 * 32] = (m1v - m2v) - m3v;                            // q = 1:  M1 - M2 - M3
#endif
            }
            // the extra operand of this round's outputs, requested before the barrier
            const int chl = 8 * rnd + c_l;                                   // cout within an M-tile
            f32x2 r0 = {0.f, 0.f}, r1 = {0.f, 0.f};
            if (spade) {
                r0 = ld2(q.f.spade_x, pix0_b + (unsigned)((e_m0 >> 1) + chl) * cstride_b);
            } else if (q.f.residual) {
                const int c0 = (full || e_m0 + chl < q.Cout) ? e_m0 + chl : q.Cout - 1, c1 = (full || e_m0 + 32 + chl < q.Cout) ? e_m0 + 32 + chl : q.Cout - 1;
                r0 = ld2(q.f.residual, pix0_b + (unsigned)c0 * cstride_b);
                r1 = ld2(q.f.residual, pix0_b + (unsigned)c1 * cstride_b);
            }
            __syncthreads();
            if (rnd == 0) {
                // The U words of the next tile's first pairs (requested during the last chunk) must be home before this tile's
                // stores enter the queue: vmcnt counts stores as well, so the counted waits of the next tile's first pairs
                // would otherwise wait for these stores to reach memory.  They have had the whole round to arrive.
                asm volatile("s_waitcnt vmcnt(0)" : "+v"(a_ring[0]), "+v"(a_ring[1]), "+v"(a_ring[2]), "+v"(a_ring[3]));
            }
            // Y[prow][q] = Y'[a0][q] + sg * (Y'[a0 + 1][q] + Y'[a0 + 2][q]),  a0 = prow
            f32x2 yv[2];                                                     // [M-tile](q = 0, 1)
            const float* exr = exr0 + (rnd & 1) * W_EXCH;
#pragma unroll
            for (int kk = 0; kk < 2; kk++)
#pragma unroll
                for (int q = 0; q < 2; q++) {
                    const float* e3 = exr + (q * 16 + kk * 8) * 32;
#if WINO_EXP & 1024
                    yv[kk][q] = sg + (float)(kk + q);
#else
                    yv[kk][q] = fmaf(sg, e3[2 * 16 * 32] + e3[4 * 16 * 32], e3[0]);
#endif
                }
            if (spade) {
                // SPADE combine (networks.py:1715-1722): M-tile 0 rows are gamma, M-tile 1 rows the beta rows of the same 32
                // channels:  y = (x - mean) * rstd * (1 + gamma) + beta
                const float mu = ep_scale[chl], rs = ep_bias[chl];
                f32x2 v = (r0 - mu) * rs * (yv[0] + 1.f) + yv[1];
                if (!plain_tail) {                                           // the consumer's pre-activation (Spade_Conv2dLayer) folded in
#pragma unroll
                    for (int e = 0; e < 2; e++) v[e] = __builtin_amdgcn_fmed3f((v[e] > 0.f ? v[e] : v[e] * slope) * gain, -cl, cl);
                }
                store2(pix0_b + (unsigned)((e_m0 >> 1) + chl) * cstride_b, v, true);
            } else {
#pragma unroll
                for (int kk = 0; kk < 2; kk++) {
                    const int co = e_m0 + 32 * kk + chl;
                    const float esc = ep_scale[32 * kk + chl], ebi = ep_bias[32 * kk + chl];
                    f32x2 v = yv[kk] * esc + (nz + ebi);
                    if (!plain_tail) {
#pragma unroll
                        for (int e = 0; e < 2; e++) v[e] = __builtin_amdgcn_fmed3f((v[e] > 0.f ? v[e] : v[e] * slope) * gain, -cl, cl);
                    }
                    v += kk ? r1 : r0;
                    store2(pix0_b + (unsigned)((full || co < q.Cout) ? co : q.Cout - 1) * cstride_b, v, co < q.Cout);
                }
            }
        }
        if (!has_next) break;
        tile = next;
        par ^= 1;
    }
}

template <int MODE, bool VEC>
int launch_wino_xf(const ConvParams& p0, hipStream_t s) {
    ConvParams p = p0;
    { static const int gmap = [] { const char* e = getenv("PG_WINO_GMAP"); return e ? atoi(e) : 1; }(); p.wino_gmap = gmap; }
    p.tilesX = (p.OW + 63) / 64;
    p.tilesY = (p.OH + 1) / 2;
    p.mblocks = p.CoutP / 64;
    const int64_t tiles = (int64_t)p.N * p.tilesX * p.tilesY * p.mblocks;
    if (tiles > 0x7fffffffLL) return PG_ERR_TOO_LARGE;
    p.total_tiles = (int)tiles;
    const int cin_loop = ((p.Cin + W_KC - 1) / W_KC) * W_KC;
    const size_t lds = ((size_t)2 * W_BUFS + 2 * W_EXCH + 2 * cin_loop + 256 + (VEC ? 6 * 256 : 0)) * sizeof(float);
    if ((int64_t)16 * cin_loop * p.CoutP * 4 > 0x7fffffffLL) return PG_ERR_TOO_LARGE;      // the U stream uses 32-bit byte offsets
    if (lds > 160 * 1024) return PG_ERR_UNSUPPORTED;
    int per_cu = (int)((160 * 1024) / lds);
    if (per_cu > 2) per_cu = 2;                             // 128 VGPRs x 8 waves per workgroup
    if (per_cu < 1) per_cu = 1;
    const int64_t blocks = tiles < (int64_t)num_cu() * per_cu ? tiles : (int64_t)num_cu() * per_cu;
    static PerDeviceOnce lds_attr;
    const hipError_t e = lds_attr.run([] { return hipFuncSetAttribute((const void*)conv2d_wino<MODE, VEC>, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024); });
    if (e != hipSuccess) return (int)e;
    hipLaunchKernelGGL((conv2d_wino<MODE, VEC>), dim3((unsigned)blocks), dim3(512), lds, s, p);
    return launch_status();
}

}  // namespace pgconv
