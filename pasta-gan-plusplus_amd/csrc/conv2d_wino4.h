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
// Mapping (second version; the first had 8 waves x 9 xi and exchanged all 36 values per output tile: its tail was 19-26 % of a launch)
//   * workgroup = 768 threads = 12 waves, ONE per CU (LDS 118-150 KB), three waves per SIMD; tile = 64 couts x (8 output rows x 64
//     output cols) = 32 tiles of 4x4 outputs, MFMA column n = 2 * tile_x + tile_y (this interleave makes the 16-byte patch reads of
//     the transform phase conflict-free: tile_y adds 8 sixteen-byte slots mod 16).
//   * GEMM phase: wave w = (a = w >> 1, mt = w & 1) owns row a of the 6x6 transform domain, xi = 6a .. 6a + 5, for cout M-tile mt:
//     6 accumulators = 96 VGPRs.  Per chunk 48 MFMAs per wave in four groups of (3 xi) x (4 channel pairs): three accumulators in
//     rotation; a group's A words are three 16-byte loads (1 KB per wave-instruction, 3 KB contiguous per group, packed by
//     pg_conv2d_winograd4_pack_weight in exactly the order the wave walks), its B words six ds_read2st64_b32 of V.
//   * transform phase: waves 0-7 (16 channels x 32 tiles, one 6x6 patch per thread); waves 8-11 own the chores instead: they issue
//     the 16-byte LDS-DMA of the raw 10-row halo tile [16][10][72] for chunk k+1 as soon as the transform of chunk k is done (gather
//     map kept in LDS; zero padding = the buffer range check on a sentinel offset), compute edge-tile gather maps and fetch the
//     epilogue constants.  Single raw buffer, single V buffer, two barriers per chunk.
//   * U words are inline-asm loads waited for with hand-counted vmcnt (one group ahead; for the issuing waves the DMA sits in the
//     queue behind the first two groups' words: vmcnt(3 + NDMA) there, vmcnt(3) after -- the third group's wait is also what
//     guarantees the DMA has landed before the next chunk's barrier).
//   * tail: a wave holds all six b of its row a, so the column half of the inverse transform (6 -> 4) is done in registers and the
//     four results of a (cout, tile) leave as ONE 16-byte LDS word; 16 couts per round; waves 0-7 then own one (cout, tile) each:
//     six 16-byte reads, the row half (6 -> 4 per column), the fused epilogue of conv2d_wino.h on four 16-byte row segments
//     addressed by 32-bit offsets against per-image bases.  SPADE mode: thread = (channel, tile, row pair) combining gamma and beta.
// Numerics: float32 throughout, weights transformed in float64 by the pack kernel and rounded once; tools/f43_error_probe.py
// measures the effect on the whole config-2 network (3.7e-5 max-abs on `img` against the direct float32 run, F(2x2): 1.0e-5).
#pragma once
#include <cstdlib>
#include <type_traits>
#include "conv2d_wino.h"

#ifndef WINO4_EXP
#define WINO4_EXP 0      // dev ablations (results wrong by design): 1 no U loads, 2 no transform phase, 4 no tail, 8 no halo DMA, 16 no output stores,
                         // 32 no finishing (exchange writes and barriers stay), 64 no exchange writes, 128 cycle stamps of workgroup 0's second tile into y[0..];
                         // debugging switches that stay correct: 256 chore waves drain their queue at the end of every chunk, 512 every U wait is vmcnt(0)
#endif

#ifndef W4X_EXP
#define W4X_EXP 0        // dev switches of the X3 form: 512 no early cache-line touches of the next halo; ablations (results wrong by design): 4 no operand split (raw words as
                         // B operands), 8 no MFMAs.  (Tried and removed, all parity-green, tools/wino4_variants.py: start stagger of the three waves of a SIMD, fixed / per-block
                         // wave priorities, the halo requested after group 0 / 1 / 2, two groups across the transform phase, A planes re-requested as soon as their last
                         // MFMA is out, every wave requesting a share of the halo: each within +-2 %; v_pk_add_f32 in the split: 1.6x SLOWER.)
#endif

namespace pgconv {

constexpr bool W4_EX2 = (WINO4_EXP & 8192) != 0;  // timing experiment (round 4, parity- and stress-green): two exchange buffers in the tail, one barrier per round instead of two --
                                                // 434 | 472 | 237 | 358 us against 430 | 476 | 239 | 358 with the single buffer (128->128 256^2 | 64->64 512^2 | 64->128 256^2 | 256->256 128^2): +-1 %, not the barriers
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
constexpr int W4_UCHUNK = 4 * W4_UGROUP;         // per (wave unit = (m-block, mt, a), chunk): groups (jg, quad)
constexpr int W4_EX = 6 * 16 * 32 * 4;           // exchange floats per round: [a][cout 16][tile 32][4 columns]
constexpr int W4_UCHUNK_X3 = 6 * W4_UGROUP;      // X3 form: per (wave unit, chunk) six groups (b), each three planes x 64 lanes x 16 B (8 bf16 channels)
typedef __bf16 w4_bf16x8 __attribute__((ext_vector_type(8)));
typedef unsigned w4_u32x4 __attribute__((ext_vector_type(4)));

// X3 form: a float32 value is the exact sum of three bf16 values (8 + 8 + 8 significand bits, by truncation).  Two values of adjacent K slots -> the three
// packed operand words (plane 0 = leading bits, 1 = middle, 2 = trailing), low half = the first value: 4 v_and + 4 v_sub + 3 v_perm per pair.
__device__ __forceinline__ void w4_split_pair(float va, float vb, unsigned& p0, unsigned& p1, unsigned& p2) {
    const unsigned ua = __float_as_uint(va), ub = __float_as_uint(vb);
    const float ra = va - __uint_as_float(ua & 0xffff0000u), rb = vb - __uint_as_float(ub & 0xffff0000u);
    const unsigned ura = __float_as_uint(ra), urb = __float_as_uint(rb);
    const float la = ra - __uint_as_float(ura & 0xffff0000u), lb = rb - __uint_as_float(urb & 0xffff0000u);
    p0 = __builtin_amdgcn_perm(ub, ua, 0x07060302u);
    p1 = __builtin_amdgcn_perm(urb, ura, 0x07060302u);
    p2 = __builtin_amdgcn_perm(__float_as_uint(lb), __float_as_uint(la), 0x07060302u);
}

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

// Six V values of transform-domain row A (xi = 6A .. 6A + 5) of this wave's 64 (channel, tile) items, written with ds_write_addtid_b32:
// address = M0 + offset + 4 * lane with no address VGPR -- 128 B/clk/CU, twice the rate of ds_write_b32, which is what bounded the
// transform phase (36 writes per thread).  V[xi] planes are 2048 bytes apart; the instruction's offset field is 16 bits, so row 5 goes
// through a second base 16384 bytes up.  M0 is compiler-reserved: saved and restored inside the statement (cdna_hip_programming.md
// section 5.7); the writes are drained here because the compiler cannot count them for the barrier that follows.
template <int A>
__device__ __forceinline__ void w4_write_row(unsigned m0_base, float v0, float v1, float v2, float v3, float v4, float v5) {
    constexpr int ADJ = A == 5 ? 16384 : 0;
    unsigned keep;
    asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %1\n\ts_nop 0\n\t"
                 "ds_write_addtid_b32 %2 offset:%8\n\tds_write_addtid_b32 %3 offset:%9\n\tds_write_addtid_b32 %4 offset:%10\n\t"
                 "ds_write_addtid_b32 %5 offset:%11\n\tds_write_addtid_b32 %6 offset:%12\n\tds_write_addtid_b32 %7 offset:%13\n\t"
                 "s_mov_b32 m0, %0"
                 : "=&s"(keep)
                 : "s"(m0_base + ADJ), "v"(v0), "v"(v1), "v"(v2), "v"(v3), "v"(v4), "v"(v5),
                   "n"((6 * A + 0) * 2048 - ADJ), "n"((6 * A + 1) * 2048 - ADJ), "n"((6 * A + 2) * 2048 - ADJ),
                   "n"((6 * A + 3) * 2048 - ADJ), "n"((6 * A + 4) * 2048 - ADJ), "n"((6 * A + 5) * 2048 - ADJ)
                 : "memory");
}

// The lane index, recomputed where a phase needs it (two VALU instructions) instead of living in a VGPR across the whole kernel: with 96
// accumulator registers, a 12-register operand ring and the 36-value patch of the transform phase the budget of 168 has no room for
// loop-invariant address registers -- a spilled one comes back through a scratch load, i.e. a vmcnt(0) in the middle of the K loop.
// (volatile: not merged with other calls, not hoisted out of the loops)
__device__ __forceinline__ int w4_fresh_lane() {
    int l;
    asm volatile("v_mbcnt_lo_u32_b32 %0, -1, 0\n\tv_mbcnt_hi_u32_b32 %0, -1, %0" : "=v"(l));
    return l;
}

// MODE: 0 = plain input, 1 = per-(n, channel) input scale (modulated convolution).  (Pre-activation launches stay on conv2d_wino.h.)
// TAIL: which operands the fused epilogue has -- its own instantiation each, because with several tails in one kernel every edit of one
// moved the others' register allocation (and their time by 3-6 %), and because the specialised ones need far fewer scalar registers
// (13-26 spilled SGPRs instead of 50):
//   W4_TAIL_ANY    whatever pg_conv2d_fusion says, decided at run time (the fallback; it keeps ALL branches on purpose: compiled without the
//                  SPADE branch, hipcc's register allocation of what remains spills 112-144 VGPRs)
//   W4_TAIL_PLAIN  scale / bias / activation only (-3.6 ... -5.7 % per launch against W4_TAIL_ANY)
//   W4_TAIL_SPADE  the SPADE combine (-2.5 ... -4.5 %)
//   W4_TAIL_STATS  the plain tail + the output's instance-norm statistics (pg_conv2d_fusion::stats_partial) -- its own instantiation for the same reason:
//                  with the statistics as a run-time branch of the plain tail every plain launch paid for the extra scalar registers
//   W4_TAIL_RES / W4_TAIL_NOISE (residual only / noise only) compile cleanly too but measured +1 % / -1 ... +5 %: not instantiated.
enum { W4_TAIL_ANY = 0, W4_TAIL_PLAIN = 1, W4_TAIL_SPADE = 2, W4_TAIL_RES = 3, W4_TAIL_NOISE = 4, W4_TAIL_STATS = 5 };
// X3 (round 6): the Winograd-domain GEMM on v_mfma_f32_32x32x16_bf16 as six products of exact three-term operand splits (fp32 accumulation): U is split once per
// weight version by pg_conv2d_winograd4x3_pack_weight, the V values a lane reads from LDS are split in registers by the multiplying wave itself -- LDS keeps the
// float32 V, the transform phase and the tail are the fp32 form's.  Per (xi, 16-channel chunk): 44 VALU + 6 MFMAs of 32 cycles instead of 8 MFMAs of 64 cycles;
// the vector work of one wave runs beside the matrix work of the other two waves of its SIMD.
template <int MODE, int TAIL, bool X3 = false>
__global__ __launch_bounds__(768, 3) void conv2d_wino4(ConvParams p) {
    extern __shared__ __attribute__((aligned(16))) float smem[];
    const int cin_loop = ((p.Cin + W4_KC - 1) / W4_KC) * W4_KC;
    const int nchunks = cin_loop / W4_KC;
    float* raw = smem;
    float* V = smem + W4_RAW;
    unsigned* gmI = (unsigned*)(V + W4_V);       // gather map of an interior tile [12][256] (byte offsets relative to the tile's first halo sample)
    unsigned* gmE = gmI + W4_NDMA * 256;         // gather map of the current edge tile (absolute in the image, sentinel outside)
    float* cs0 = (float*)(gmE + W4_NDMA * 256);  // prologue scale of two consecutive tiles [2][cin_loop]
    float* ep0 = cs0 + 2 * cin_loop;             // epilogue scale / bias of two consecutive tiles [2][64 + 64]
    float* touch_sink = ep0 + 256;               // X3 form: 64 floats nobody reads (LDS target of the halo's cache-line touches)
    // (W4_EX2: the tail's SECOND exchange buffer is the last third of V plus both gather maps -- exactly W4_EX floats; the chore waves rebuild
    //  the maps during the next tile's first transform phase, which they sit out anyway)
    static_assert(W4_V - W4_EX + 2 * W4_NDMA * 256 == W4_EX, "second exchange buffer = tail of V + the two gather maps");

    const int wave = __builtin_amdgcn_readfirstlane((int)threadIdx.x >> 6);
    const unsigned smem_b = __builtin_amdgcn_readfirstlane(lds_offset(smem));
    if (smem_b > 1024u) __builtin_trap();        // ds_write_addtid_b32 bases are 16-bit: the V buffer must start below 64 KB (dynamic LDS is this kernel's only LDS: 0)
    const int HW = p.H * p.W;
    const int total = p.total_tiles;
    const int q8 = total >> 3, r8 = total & 7;
    const int ta = wave >> 1, mt = wave & 1;     // GEMM role: transform-domain row a, cout M-tile
    const bool chore = wave >= 8;                // waves 8-11: DMA issue, gather maps, epilogue constants (they sit out the transform and the finish)
    const int cw = wave - 8;                     // chore wave 0..3; chore thread index tc = 64 cw + lane
    const int sh = p.pad_x & 3;                  // float shift of the staged tile: patch column 0 of tile_x lands on LDS column 4 tile_x + cbase
    constexpr int cbase = 4;                     // = 4 - pad_x + sh for the paddings the launcher admits (1, 3): a multiple of 4 -> 16-byte aligned patch reads

    int n = 0, oy0 = 0, ox0 = 0, m0 = 0;
    bool edge = false;
    i32x4 xrsrc;

    // Interior gather map: tile-independent.  Word f of the buffer = (channel f / 180, row (f % 180) / 18, word f % 18).
    auto build_gmI = [&]() {
        if (chore) {
            const int tc = 64 * cw + w4_fresh_lane();
#pragma unroll
            for (int i = 0; i < W4_NDMA; i++) {
                const int f = i * 256 + tc;
                const int c = f / 180, rem = f % 180;
                const int row = rem / 18, wd = rem % 18;
                gmI[i * 256 + tc] = (unsigned)(c * HW + row * p.W + 4 * wd) * 4u;
            }
        }                                        // (read back by the writing thread only: no barrier needed)
    };
    build_gmI();
    auto build_gmE = [&](int gy0, int gx0) {     // gather map of an edge tile whose first halo sample is (gy0, gx0)
        if (chore) {
            const int tt = 64 * cw + w4_fresh_lane();
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
    };

    auto prep_tile = [&](int tile, float* cs) {
        const int xcd = tile & 7;
        int L = (xcd < r8 ? xcd * (q8 + 1) : r8 * (q8 + 1) + (xcd - r8) * q8) + (tile >> 3);
        const int mb = L % p.mblocks; L /= p.mblocks;
        const int tx = L % p.tilesX; L /= p.tilesX;
        const int ty = L % p.tilesY;
        n = L / p.tilesY;
        oy0 = ty * 8; ox0 = tx * 64; m0 = mb * 64;
        if (MODE == 1) {
            const float* in_scale = p.f.in_scale ? p.f.in_scale + (int64_t)n * p.Cin : nullptr;
            for (int c = 64 * wave + w4_fresh_lane(); c < cin_loop; c += 768) cs[c] = ((in_scale && c < p.Cin) ? ld_opaque(in_scale + c) : 1.f) * p.f.in_gain;
        }
        const int gy0 = oy0 - p.pad_y, gx0 = ox0 - 4;
        edge = !(gy0 >= 0 && gy0 + W4_ROWS <= p.H && gx0 >= 0 && gx0 + W4_LROW <= p.W);           // wave-uniform
        if (edge) build_gmE(gy0, gx0);
        const int shift = edge ? 0 : (gy0 * p.W + gx0) * 4;             // interior: offsets are relative to the tile's first halo sample
        const uint64_t base = (uint64_t)(uintptr_t)(p.x + (int64_t)n * p.Cin * HW) + (uint64_t)(int64_t)shift;
        xrsrc[0] = __builtin_amdgcn_readfirstlane((int)(unsigned)base);
        xrsrc[1] = __builtin_amdgcn_readfirstlane((int)(unsigned)(base >> 32) & 0xffff);
        xrsrc[2] = p.Cin * HW * 4 - shift;                               // same absolute end: channels beyond Cin read as zero
        xrsrc[3] = 0x00020000;
    };

    // part < 0: the whole chunk; part 0 / 1 / 2 (W4_SPREAD): a third of its requests (4 + 4 + 4, wave 8 one more in the last part)
    auto issue_chunk = [&](int c0, const int part = -1) __attribute__((always_inline)) {
#if !(WINO4_EXP & 8)
        if (chore) {
            const unsigned xs_b = smem_b + (unsigned)sh * 4u;
            const int soff = c0 * HW * 4;
            const unsigned* gm = edge ? gmE : gmI;
            const int tc = 64 * cw + w4_fresh_lane();
            const int i0 = part < 0 ? 0 : 4 * part, i1 = part < 0 ? W4_NDMA - 1 : (4 * part + 4 < W4_NDMA - 1 ? 4 * part + 4 : W4_NDMA - 1);
            unsigned off[W4_NDMA];
#pragma unroll
            for (int i = 0; i < W4_NDMA; i++)
                if ((i >= i0 && i < i1) || (i == W4_NDMA - 1 && (part < 0 || part == 2))) off[i] = gm[i * 256 + tc];
#pragma unroll
            for (int i = 0; i < W4_NDMA - 1; i++)
                if (i >= i0 && i < i1) dma_dwordx4_buf(xrsrc, xs_b + (unsigned)(256 * i + 64 * cw) * 16u, off[i], soff);
            if (cw == 0 && (part < 0 || part == 2)) dma_dwordx4_buf(xrsrc, xs_b + (unsigned)(256 * (W4_NDMA - 1)) * 16u, off[W4_NDMA - 1], soff);    // words 2816..2879
        }
#endif
    };

    // X3 form: the halo of the chunk that the NEXT GEMM phase will request, touched one transform phase early -- one dword per 128-byte line (4 per 288-byte row:
    // words 0, 8, 16, 17) as LDS-DMA into a sink, 3 requests per chore thread (640 touches, the last 128 slots repeat).  Why: the CU's vector memory path returns in
    // order, so the HBM misses of the halo DMA (~2 k cycles) used to hold back every wave's A words queued behind them, once per chunk, in the phase that lives on
    // that stream; touched during the transform phase -- when no wave waits for vector memory -- the lines are in L2 by the time the DMA asks for them.
    constexpr int W4_NTOUCH = 3;
    auto touch_chunk = [&](int c0) __attribute__((always_inline)) {
        if (chore) {
            const int soff = c0 * HW * 4;
            const unsigned* gm = edge ? gmE : gmI;
            const int tc = 64 * cw + w4_fresh_lane();
            const unsigned sink_b = smem_b + (unsigned)((touch_sink - smem) * 4);
#pragma unroll
            for (int i = 0; i < W4_NTOUCH; i++) {
                int idx = i * 256 + tc;
                idx = idx < 640 ? idx : idx - 128;
                const int row = idx >> 2, part = idx & 3;
                const int f = (row / 10) * 180 + (row % 10) * 18 + (part == 3 ? 17 : 8 * part);
                const unsigned off = gm[f];
                unsigned keep;
                asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %1\n\ts_nop 0\n\tbuffer_load_dword %2, %3, %4 offen lds\n\ts_mov_b32 m0, %0"
                             : "=&s"(keep) : "s"(sink_b), "v"(off), "s"(xrsrc), "s"(soff) : "memory");
            }
        }
    };

    f32x16 acc[6];                                   // [b]; never zeroed: see `chunk`
    const f32x16 zero16 = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};

    // A-operand stream of this wave: [m-block][mt][a][chunk][group = (jg, quad)][xi 3][lane 64][4 pairs] floats, walked strictly
    // forwards inside a tile, one group ahead.
    unsigned pa;
    auto a_reset = [&](int m0_) { pa = (unsigned)((((m0_ >> 6) * 2 + mt) * 6 + ta) * nchunks) * (unsigned)(X3 ? W4_UCHUNK_X3 : W4_UCHUNK) + (unsigned)w4_fresh_lane() * 16u; };
    auto load_u = [&](f32x4 (&dst)[3]) {
#if !(WINO4_EXP & 1)
        asm volatile("global_load_dwordx4 %0, %1, %2" : "=v"(dst[0]) : "v"(pa), "s"(p.wp));
        asm volatile("global_load_dwordx4 %0, %1, %2 offset:1024" : "=v"(dst[1]) : "v"(pa), "s"(p.wp));
        asm volatile("global_load_dwordx4 %0, %1, %2 offset:2048" : "=v"(dst[2]) : "v"(pa), "s"(p.wp));
#endif
        pa += (unsigned)W4_UGROUP;
    };
    // wait for the three words of one group; `younger` = vector-memory operations issued after them that may still be in flight
    auto wait_u = [&](f32x4 (&g)[3], bool dma_younger, const int spread_g = -1) {
        if (WINO4_EXP & 512) {
            asm volatile("s_waitcnt vmcnt(0)");
        } else if (dma_younger && spread_g >= 0) {
            // W4_SPREAD: behind group g's words sit (g = 1) G2 + P0 = 7, (g = 2) P0 + G3 + P1 = 11, (g = 3) P1 + nG0 + P2 = 10 (wave 8: 11) younger operations
            if (spread_g == 0) asm volatile("s_waitcnt vmcnt(3)");
            else if (spread_g == 1) asm volatile("s_waitcnt vmcnt(7)");
            else if (spread_g == 2) asm volatile("s_waitcnt vmcnt(11)");
            else if (cw == 0) asm volatile("s_waitcnt vmcnt(11)");
            else asm volatile("s_waitcnt vmcnt(10)");
        } else if (dma_younger) {
            if (cw == 0) asm volatile("s_waitcnt vmcnt(%0)" :: "n"(3 + W4_NDMA));
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


#if WINO4_EXP & 128
    // stamps go straight to LDS (wave w: stamp_lds[40 w + i], low 32 bits of s_memtime): held in registers they pushed the kernel into scratch spills
    unsigned* stamp_lds = (unsigned*)(touch_sink + 64);
    int tiles_done = 0;
    auto put_stamp = [&](int i) __attribute__((always_inline)) {
        const unsigned t = (unsigned)__builtin_amdgcn_s_memtime();
        if (w4_fresh_lane() == 0) stamp_lds[40 * wave + i] = t;
    };
#define W4_STAMP(i) do { if (tiles_done == 1) put_stamp(i); } while (0)
#define W4_XSTAMP(i) do { if (tiles_done == 1 && k == 2) put_stamp(16 + (i)); } while (0)
#else
#define W4_XSTAMP(i) do { } while (0)
#define W4_STAMP(i) do { } while (0)
#endif
    int tile = blockIdx.x;
    int par = 0;
    int tiles_run = 0;
    prep_tile(tile, cs0);
    f32x4 ur[2][3];
    a_reset(m0);
    load_u(ur[0]);                                                      // group 0 of the first chunk
    issue_chunk(0);
    dma_wait_all();
    while (true) {
        int e_n = n, e_oy0 = oy0, e_ox0 = ox0, e_m0 = m0;
        bool has_next = false;
        int next = tile;
        const float* cs_cur = cs0 + par * cin_loop;
        float* ep_scale = ep0 + par * 128;
        float* ep_bias = ep_scale + 64;

        W4_STAMP(0);
        // One K chunk.  FIRST (chunk 0 of a tile) is a second copy of the body in which the first MFMA into each accumulator reads the inline
        // constant 0 as its C operand: the 96 v_mov per wave and tile that zeroed the accumulators (1.1 k issue cycles per SIMD and tile with
        // nothing else running) are gone.
        auto chunk = [&](const int k, auto first_tag) __attribute__((always_inline)) {
            constexpr bool FIRST = decltype(first_tag)::value;
            __syncthreads();                                            // A: raw(k) has landed (the issuing waves waited for their requests at
                                                                        //    their third U group), V is free (every wave is past GEMM(k-1) / the tail)
            if (k == 2) W4_STAMP(1);
            bool touched = false;
            if (X3 && !(W4X_EXP & 512) && nchunks >= 2 && (k + 1 < nchunks || has_next)) {      // (k + 1 == nchunks: the state is already the next tile's, see request_next)
                touch_chunk(k + 1 < nchunks ? (k + 1) * W4_KC : 0);
                touched = true;
            }
            if (W4_EX2 && k == 0 && tiles_run > 0) {     // the previous tile's tail used the maps' LDS as exchange space: the chore waves (idle in this phase) rebuild them
                build_gmI();
                if (edge) build_gmE(oy0 - p.pad_y, ox0 - 4);
            }
            // ---- transform phase (waves 0-7): one 6x6 patch per thread -> 36 V values
#if !(WINO4_EXP & 2)
            if (!chore) {
                // transform role (waves 0-7): channel 2 * wave + half of the chunk, tile n = l31 -> (tile_x = n >> 1, tile_y = n & 1)
                const int lane = w4_fresh_lane(), half = lane >> 5, l31 = lane & 31;
                const float* rb = raw + (2 * wave + half) * W4_CHF + (4 * (l31 & 1)) * W4_LROW + 4 * (l31 >> 1) + cbase;
                float d[6][6];
#pragma unroll
                for (int r = 0; r < 6; r++) {
                    const f32x4 lo = *(const f32x4*)(rb + r * W4_LROW), hi = *(const f32x4*)(rb + r * W4_LROW + 4);
                    d[r][0] = lo[0]; d[r][1] = lo[1]; d[r][2] = lo[2]; d[r][3] = lo[3]; d[r][4] = hi[0]; d[r][5] = hi[1];
                }
                float sc = 1.f;
                if (MODE == 1) sc = cs_cur[k * W4_KC + 2 * wave + half];
#pragma unroll
                for (int j = 0; j < 6; j++)                              // columns: over the patch rows
                    w4_bt(d[0][j], d[1][j], d[2][j], d[3][j], d[4][j], d[5][j], d[0][j], d[1][j], d[2][j], d[3][j], d[4][j], d[5][j]);
                // rows: over the patch columns, then out to V[6a + b]
                const unsigned vbase = smem_b + (unsigned)(W4_RAW * 4 + wave * 256);      // bytes: V + 64 floats per transform wave (< 65536: checked by the host)
#define W4_ROW(A) { float v0, v1, v2, v3, v4, v5; \
                    w4_bt(d[A][0], d[A][1], d[A][2], d[A][3], d[A][4], d[A][5], v0, v1, v2, v3, v4, v5); \
                    if (MODE == 1) { v0 *= sc; v1 *= sc; v2 *= sc; v3 *= sc; v4 *= sc; v5 *= sc; } \
                    w4_write_row<A>(vbase, v0, v1, v2, v3, v4, v5); }
                W4_ROW(0) W4_ROW(1) W4_ROW(2) W4_ROW(3) W4_ROW(4) W4_ROW(5)
#undef W4_ROW
                asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
            }
#endif
            if (k == 2) W4_STAMP(2);
            __syncthreads();                                            // B: V(k) complete, raw free
            if (k == 2) W4_STAMP(3);

            if (k == 0) {
                // epilogue constants of THIS tile (wave 8 only, before anything new enters its queue: ld_opaque waits for vmcnt(0))
                const auto& qa = *fresh_args();
                const int tc = wave == 8 ? w4_fresh_lane() : 64;              // wave 8 only
                if (TAIL == W4_TAIL_SPADE || (TAIL == W4_TAIL_ANY && qa.f.spade_x)) {
                    if (tc < 32) {
                        const int ch = (m0 >> 1) + tc;
                        ep_scale[tc] = ld_opaque(qa.f.spade_mean + n * (qa.Cout >> 1) + ch);
                        ep_bias[tc] = ld_opaque(qa.f.spade_rstd + n * (qa.Cout >> 1) + ch);
                    }
                } else if (tc < 64) {
                    const int co = m0 + tc;
                    const bool ok = co < qa.Cout;
                    const int cc = ok ? co : 0;
                    const float scv = qa.f.out_scale ? ld_opaque(qa.f.out_scale + (int64_t)n * qa.Cout + cc) : 1.f;
                    const float bi = qa.f.bias ? ld_opaque(qa.f.bias + cc) : 0.f;
                    // linear / relu / lrelu are positively homogeneous and the host guarantees gain > 0:  act(v * s + b) * gain = act(v * (s * gain) + b * gain),
                    // so the gain rides in the per-cout constants (and in the noise gain) and the tail's activation is max(w, w * slope) alone
                    ep_scale[tc] = ok ? scv * qa.f.gain : 0.f;
                    ep_bias[tc] = ok ? bi * qa.f.gain : 0.f;
                }
            }
            load_u(ur[1]);                                               // group 1 of this chunk (BEFORE the DMA: its wait must not depend on it)
            // ---- request the next chunk (of this tile, or the first of the next tile).  The next tile is prepared (coordinates, edge
            // gather map, prologue scales) one chunk EARLY -- right after this tile's last chunk has been requested -- so that the last
            // chunk only has to issue: the chore waves would otherwise reach the tail thousands of cycles after everybody else.
            bool issued = true;
            auto request_next = [&]() __attribute__((always_inline)) {
                const bool prep_now = nchunks == 1 ? true : k + 2 == nchunks;
                if (k + 1 < nchunks) issue_chunk((k + 1) * W4_KC);
                if (prep_now) {
                    e_n = n; e_oy0 = oy0; e_ox0 = ox0; e_m0 = m0;
                    next = tile + gridDim.x;
                    has_next = next < total;
                    if (has_next) prep_tile(next, cs0 + (par ^ 1) * cin_loop);
                }
                if (k + 1 == nchunks) {
                    if (has_next) issue_chunk(0);
                    else issued = false;
                }
            };
            if constexpr (X3) {
                // ---- GEMM phase, X3 form: six groups = the six xi of this wave's row; group g: A = three 16-byte words (planes 0..2 of the lane's 8 channels
                // 2 j + h), B = the lane's 8 float32 V values split into three packed planes.  The six MFMAs of a group are ONE dependent chain (same accumulator;
                // stamps: ~87 cycles per link for a wave alone on its SIMD), so the wave's own issue slots between them are free: the split of the NEXT group's V
                // values (4 pairs x 11 VALU) sits in the first four gaps, and the V values of the group after that are requested in the fourth (-2.5 % per launch
                // against split-then-multiply).  A ring: group g in slot g & 1; group 1 requested above, group g + 2 after the MFMAs of group g (g = 4: group 0 of
                // the next chunk / tile; g = 5: nothing -- two groups crossing the transform phase measured +-0 and their 12 live registers spill in the SPADE tail).
                request_next();
                const bool dma_q = issued && chore;
                const float* vb = V + 6 * ta * 512 + w4_fresh_lane();
                float br[8];
                auto read_b8 = [&](int b, float (&dst)[8]) {
#pragma unroll
                    for (int j = 0; j < 8; j++) dst[j] = vb[b * 512 + j * 64];
                };
                w4_u32x4 bp[2][3];
                read_b8(0, br);
#pragma unroll
                for (int q = 0; q < 4; q++) {
                    unsigned p0, p1, p2;
                    w4_split_pair(br[2 * q], br[2 * q + 1], p0, p1, p2);
                    bp[0][0][q] = p0; bp[0][1][q] = p1; bp[0][2][q] = p2;
                }
                read_b8(1, br);
                W4_XSTAMP(0);
#pragma unroll
                for (int g = 0; g < 6; g++) {
                    W4_XSTAMP(1 + 3 * g);
                    if (g == 0 && dma_q && touched) {
                        if (cw == 0) asm volatile("s_waitcnt vmcnt(%0)" :: "n"(3 + W4_NTOUCH + W4_NDMA));
                        else asm volatile("s_waitcnt vmcnt(%0)" :: "n"(3 + W4_NTOUCH + W4_NDMA - 1));
                        asm volatile("" : "+v"(ur[0][0]), "+v"(ur[0][1]), "+v"(ur[0][2]));
                    } else wait_u(ur[g & 1], g < 2 && dma_q);
                    W4_XSTAMP(2 + 3 * g);
                    __builtin_amdgcn_sched_barrier(0);
                    // small products first:  u2 v0,  u1 v1,  u1 v0,  u0 v2,  u0 v1,  u0 v0
                    constexpr int PA[6] = {2, 1, 1, 0, 0, 0}, PB[6] = {0, 1, 0, 2, 1, 0};
#pragma unroll
                    for (int t = 0; t < 6; t++) {
                        if (!(W4X_EXP & 8) || bp[0][0][0] == 0x12345678u) acc[g] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(w4_bf16x8, ur[g & 1][PA[t]]), __builtin_bit_cast(w4_bf16x8, bp[g & 1][PB[t]]),
                                                                         (FIRST && t == 0) ? zero16 : acc[g], 0, 0, 0);
                        __builtin_amdgcn_sched_barrier(0);
                        if (g + 1 < 6 && t < 4) {
                            unsigned p0, p1, p2;
                            if (W4X_EXP & 4) { p0 = __float_as_uint(br[2 * t]); p1 = __float_as_uint(br[2 * t + 1]); p2 = p0; }
                            else w4_split_pair(br[2 * t], br[2 * t + 1], p0, p1, p2);
                            bp[(g + 1) & 1][0][t] = p0; bp[(g + 1) & 1][1][t] = p1; bp[(g + 1) & 1][2][t] = p2;
                            if (t == 3 && g + 2 < 6) read_b8(g + 2, br);
                            __builtin_amdgcn_sched_barrier(0);
                        }
                    }
                    W4_XSTAMP(3 + 3 * g);
                    if (g == 4 && k + 1 == nchunks) a_reset(m0);
                    if (g < 5) load_u(ur[g & 1]);
                    __builtin_amdgcn_sched_barrier(0);
                }
                if (k == 2) W4_STAMP(4);
                if (!issued) dma_wait_all();
                if ((WINO4_EXP & 256) && chore) dma_wait_all();
            } else {
            constexpr bool LATE = (WINO4_EXP & 1024) != 0;               // timing experiment: the chore waves request the next chunk after group 1's MFMAs instead of before group 0's
            // W4_SPREAD (timing experiment, bit 16384): a third of the chunk's requests behind each of the first three MFMA groups, so that a request's issue
            // overlaps the wave's own MFMA in flight instead of delaying its first one; the next tile is then prepared after the last third.  Measured (round 4,
            // results identical): 4-6 % SLOWER on every shape (427 -> 446 us at 128->128 256^2) -- all requests up front stays the best placement for this kernel.
            const bool spread = (WINO4_EXP & 16384) != 0 && nchunks >= 2;
            int pend_c0 = -1;
            if (spread) {
                if (k + 1 < nchunks) pend_c0 = (k + 1) * W4_KC;
                else if (k + 1 == nchunks) {                             // (the next tile was prepared during the previous chunk)
                    if (has_next) pend_c0 = 0; else issued = false;
                }
            } else if (!LATE) request_next();
            bool dma_q = issued && chore;                                // this wave put DMA requests behind the words of groups 0 and 1 (LATE: of groups 2 and 3)

            // ---- GEMM phase: 4 groups of (3 xi) x (4 channel pairs).  Ring: group g in slot g & 1; requests: group 1 above, group g + 2
            // after the MFMAs of group g (g = 2: group 0 of the next chunk / tile; g = 3: nothing -- one group crosses the transform phase).
            if ((WINO4_EXP & 2048) && chore) __builtin_amdgcn_s_setprio(1);      // timing experiments: the chore waves (the youngest of each SIMD, with the DMA issue on top) /
            if ((WINO4_EXP & 4096) && !chore) __builtin_amdgcn_s_setprio(1);     // the other eight waves at priority 1 during the GEMM phase
            const float* vb = V + 6 * ta * 512 + w4_fresh_lane();                // B operand base: V[6a + b][pair][half][l31]
            // B words are read half a group (3 xi x 2 pairs) ahead of the MFMAs that use them: 6 MFMAs = 384 matrix-pipe cycles cover
            // the LDS latency, so a wave only ever waits for LDS at the first half group of a chunk.
            float bw[2][3][2];
            auto read_b = [&](int hg, float (&dst)[3][2]) {                  // hg = half group 0..7: (jg, quad, pair half)
                const int jg = hg >> 2, quad = (hg >> 1) & 1, ph = hg & 1;
#pragma unroll
                for (int jj = 0; jj < 3; jj++)
#pragma unroll
                    for (int s = 0; s < 2; s++) dst[jj][s] = vb[(3 * jg + jj) * 512 + (4 * quad + 2 * ph + s) * 64];
            };
            read_b(0, bw[0]);
#pragma unroll
            for (int g = 0; g < 4; g++) {
                const int jg = g >> 1;
                if (spread) wait_u(ur[g & 1], dma_q, g);
                else wait_u(ur[g & 1], (LATE ? g >= 2 : g < 2) && dma_q);
#pragma unroll
                for (int ph = 0; ph < 2; ph++) {
                    const int hg = 2 * g + ph;
                    if (hg + 1 < 8) read_b(hg + 1, bw[(hg + 1) & 1]);
                    __builtin_amdgcn_sched_barrier(0);                       // the reads go out BEFORE this half group's MFMAs
#pragma unroll
                    for (int s = 0; s < 2; s++)
#pragma unroll
                        for (int jj = 0; jj < 3; jj++)
                            acc[3 * jg + jj] = __builtin_amdgcn_mfma_f32_32x32x2f32(ur[g & 1][jj][2 * ph + s], bw[hg & 1][jj][s],
                                                                                    (FIRST && g == 2 * jg && ph == 0 && s == 0) ? zero16 : acc[3 * jg + jj], 0, 0, 0);
                }
                if (g == 2 && k + 1 == nchunks) a_reset(m0);             // from here on: the next tile's first group (m0 is already the next tile's)
                if (g < 3) load_u(ur[g & 1]);
                if (LATE && g == 1) { request_next(); dma_q = issued && chore; }
                if (spread && g < 3) {
                    if (pend_c0 >= 0) issue_chunk(pend_c0, g);
                    if (g == 2 && k + 2 == nchunks) {                    // prepare the next tile (after this tile's last chunk has been requested in full)
                        e_n = n; e_oy0 = oy0; e_ox0 = ox0; e_m0 = m0;
                        next = tile + gridDim.x;
                        has_next = next < total;
                        if (has_next) prep_tile(next, cs0 + (par ^ 1) * cin_loop);
                    }
                }
                __builtin_amdgcn_sched_barrier(0);
            }
            if (WINO4_EXP & (2048 | 4096)) __builtin_amdgcn_s_setprio(0);
            if (k == 2) W4_STAMP(4);
            if (spread && dma_q) asm volatile("s_waitcnt vmcnt(0)" ::: "memory");   // the last third must have landed before the next barrier A
            if (LATE && dma_q) asm volatile("s_waitcnt vmcnt(3)" ::: "memory");   // the halo must have landed before the next barrier A; only the next chunk's first U group is younger
            if (!issued) dma_wait_all();                                 // last chunk of the last tile: nothing counted behind us
            if ((WINO4_EXP & 256) && chore) dma_wait_all();
            }
        };
        chunk(0, std::true_type{});
#pragma unroll 1
        for (int k = 1; k < nchunks; k++) chunk(k, std::false_type{});
        W4_STAMP(5);

        // ---- inverse transform + fused epilogue, 16 couts per round through LDS (the V buffer)
#if WINO4_EXP & 4
        { float sm = 0.f;
          for (int j = 0; j < 6; j++) for (int k = 0; k < 16; k++) sm += acc[j][k];
          if (sm == 12345.678f) p.y[w4_fresh_lane()] = sm; }
        __syncthreads();
        if (!has_next) break;
        tile = next; par ^= 1;
        continue;
#endif
        // Everything the rounds need is read from the kernel-argument segment ONCE here (a scalar load per use inside the rounds --
        // what re-reading through an opaque pointer turns into -- cost ~4 us per tile: with one workgroup per CU nothing hides it).
        const auto& qa = *fresh_args();
        const bool spade = TAIL == W4_TAIL_SPADE ? true : (TAIL == W4_TAIL_ANY ? qa.f.spade_x != nullptr : false);
        const float gain = qa.f.gain, slope = act_slope(qa.f.act, qa.f.alpha);
        const float cl = qa.f.clamp >= 0.f ? qa.f.clamp : __builtin_inff();
        const bool plain_tail = slope == 1.f && gain == 1.f && qa.f.clamp < 0.f;
        const int OHv = qa.OH, OWv = qa.OW, Coutv = qa.Cout;
        const float ngain = qa.f.noise_gain * (spade ? 1.f : gain);      // (gain folded: see the epilogue constants)
        // (the host only launches this kernel with 16-byte addressable outputs: unit x stride, strides and OW multiples of 4, aligned bases)
        const bool full = e_oy0 + 8 <= OHv && e_ox0 + 64 <= OWv && e_m0 + 64 <= Coutv;      // wave-uniform: no predicates needed
        const bool seg_rows_full = full;
        // per-image bases (uniform) + 32-bit byte offsets inside the image (the host checks that one image of y stays below 4 GB)
        const int64_t img_off = (int64_t)e_n * qa.ys[0];
        float* y_n = qa.y + img_off;
        const float* res_n = ((TAIL == W4_TAIL_ANY || TAIL == W4_TAIL_RES) && qa.f.residual) ? qa.f.residual + img_off : nullptr;
        const float* spx_n = spade ? qa.f.spade_x + img_off : nullptr;
        const float* nz_n = ((TAIL == W4_TAIL_ANY || TAIL == W4_TAIL_NOISE) && qa.f.noise) ? qa.f.noise + (int64_t)e_n * qa.f.noise_batch_stride : nullptr;
        const unsigned cstride_b = (unsigned)qa.ys[1] * 4u, rstride_b = (unsigned)qa.ys[2] * 4u, nzrow_b = (unsigned)OWv * 4u;
        f32x4* ex4 = (f32x4*)V;
        const f32x4 zero4 = {0.f, 0.f, 0.f, 0.f};
        // act(v) * gain = max(v * gain, v * (slope * gain)) for 0 <= slope <= 1, gain > 0 (the host guarantees both): two multiplies and a
        // max instead of multiply / compare / select / multiply
        const float g_pos = gain, g_neg = gain * slope;
        auto act4 = [&](f32x4 v) {                                       // SPADE tail: the gain cannot ride in per-cout constants there
            if (!plain_tail) {
#pragma unroll
                for (int e = 0; e < 4; e++) v[e] = __builtin_amdgcn_fmed3f(fmaxf(v[e] * g_pos, v[e] * g_neg), -cl, cl);
            }
            return v;
        };
        const bool need_act = slope != 1.f, need_clamp = qa.f.clamp >= 0.f;      // wave-uniform
        auto act4n = [&](f32x4 v) {                                      // every other tail: gain already applied (see the epilogue constants)
            if (need_act) {
#pragma unroll
                for (int e = 0; e < 4; e++) v[e] = fmaxf(v[e], v[e] * slope);
            }
            if (need_clamp) {
#pragma unroll
                for (int e = 0; e < 4; e++) v[e] = __builtin_amdgcn_fmed3f(v[e], -cl, cl);
            }
            return v;
        };
        auto load4 = [&](const float* base, unsigned off_b, bool ok) {
            return (seg_rows_full || ok) ? *(const f32x4*)((const char*)base + off_b) : zero4;
        };
        auto store4 = [&](unsigned off_b, f32x4 v, bool ok) {
            if (WINO4_EXP & 16) { if (v[0] == 12345.678f) *(f32x4*)((char*)y_n + off_b) = v; return; }
            if (seg_rows_full || ok) *(f32x4*)((char*)y_n + off_b) = v;
        };
        // Finishing role of a thread in round `rnd`, rebuilt from the lane index where it is used (a handful of VALU instructions; kept
        // live -- or precomputed for all four rounds, as the compiler prefers -- it costs registers that the rounds do not have):
        // non-SPADE thread = (cout c16 of 16, tile fn); SPADE thread = (channel c8 of 8, tile fn, row pair rh).
        struct Role { int t, fn, oyb, rh, c8, col; unsigned ob, nzb; bool cok; };
        auto role_of = [&](int rnd) {
            Role g;
            const int lane = w4_fresh_lane();
            g.t = 64 * wave + lane;
            g.fn = g.t & 31;
            const int ftx = g.fn >> 1, fty = g.fn & 1, c16 = g.t >> 5;
            g.oyb = e_oy0 + 4 * fty;
            const int oxb = e_ox0 + 4 * ftx;
            g.rh = wave >> 2; g.c8 = c16 & 7;                            // (rh: waves 0-3 | 4-7 -- wave-uniform, kept on the scalar unit)
            g.col = (c16 >> 3) * 32 + 8 * rnd + (c16 & 7);
            const int ch = spade ? (e_m0 >> 1) + 8 * rnd + g.c8 : e_m0 + g.col;
            const bool chan_ok = spade || ch < Coutv;
            g.cok = chan_ok && oxb < OWv;                               // OW % 4 == 0: a 4-pixel segment is inside or outside as a whole
            g.ob = (unsigned)g.oyb * rstride_b + (unsigned)oxb * 4u + (unsigned)(chan_ok ? ch : Coutv - 1) * cstride_b + (spade ? (unsigned)(2 * g.rh) * rstride_b : 0u);
            g.nzb = ((unsigned)g.oyb * (unsigned)OWv + (unsigned)oxb) * 4u;
            return g;
        };
        // Output statistics (pg_conv2d_fusion::stats_partial; plain tail only): sum and M2 (squared deviations from the tile's own mean) of this tile's in-image outputs per cout
        float* stats_p = TAIL == W4_TAIL_STATS ? qa.f.stats_partial : nullptr;                                             // (wave-uniform)
        const int stats_T = qa.tilesX * qa.tilesY, stats_t = (e_oy0 >> 3) * qa.tilesX + (e_ox0 >> 6);
        const bool op_is_noise = !spade && !res_n && nz_n;
        const bool late_noise = !spade && res_n && nz_n;
        const bool has_operand = spade || res_n || nz_n;
        auto request = [&](int rnd, f32x4 (&dst)[4]) {
            if (!has_operand) return;                                    // (wave-uniform; `dst` keeps the zeros it was declared with)
#pragma unroll
            for (int r = 0; r < 4; r++) dst[r] = zero4;
            if (chore || (WINO4_EXP & 32)) return;
            const Role g = role_of(rnd);
            if (spade) {
#pragma unroll
                for (int rr = 0; rr < 2; rr++) dst[rr] = load4(spx_n, g.ob + (unsigned)rr * rstride_b, g.cok && g.oyb + 2 * g.rh + rr < OHv);
            } else if (res_n) {
#pragma unroll
                for (int r = 0; r < 4; r++) dst[r] = load4(res_n, g.ob + (unsigned)r * rstride_b, g.cok && g.oyb + r < OHv);
            } else if (nz_n) {
#pragma unroll
                for (int r = 0; r < 4; r++) dst[r] = load4(nz_n, g.nzb + (unsigned)r * nzrow_b, g.cok && g.oyb + r < OHv);
            }
        };
        // Column half of the inverse transform (over b) in registers, for all 16 accumulator rows at once: 96 live accumulator registers
        // become 64 before the rounds start.
        f32x4 yy[16];
#pragma unroll
        for (int r = 0; r < 16; r++) {
            float y0, y1, y2, y3;
            w4_at(acc[0][r], acc[1][r], acc[2][r], acc[3][r], acc[4][r], acc[5][r], y0, y1, y2, y3);
            yy[r] = (f32x4){y0, y1, y2, y3};
        }
        W4_STAMP(6);
        f32x4 op[4] = {zero4, zero4, zero4, zero4};
        request(0, op);
        __syncthreads();                                                 // the exchange area IS the V buffer: every wave must be past its last GEMM reads before the first word is written
        // couts 8 rnd + 4 half + i of this wave's M-tile come from accumulator registers 4 rnd + i; the four column results of
        // (a, cout, tile) leave as one 16-byte word.  W4_EX2: round rnd goes through buffer rnd & 1, and round rnd + 1 is written BEFORE round rnd is
        // finished -- one barrier per round (it orders "rnd + 1 written" before its readers and "rnd read" before rnd + 2 overwrites it).
        auto write_ex = [&](const int rnd) __attribute__((always_inline)) {
            const int lane = w4_fresh_lane();
            f32x4* exw = ex4 + (W4_EX2 ? (rnd & 1) * (W4_EX / 4) : 0) + (ta * 16 + mt * 8 + 4 * (lane >> 5)) * 32 + (lane & 31);
#pragma unroll
            for (int i = 0; i < 4; i++)
                if (!(WINO4_EXP & 64) || yy[4 * rnd + i][0] == 12345.678f) exw[i * 32] = yy[4 * rnd + i];
        };
        if (W4_EX2) { write_ex(0); __syncthreads(); }
#pragma unroll
        for (int rnd = 0; rnd < 4; rnd++) {
            const f32x4* exr = ex4 + (W4_EX2 ? (rnd & 1) * (W4_EX / 4) : 0);
            if (W4_EX2) { if (rnd < 3) write_ex(rnd + 1); }
            else write_ex(rnd);
            if (rnd == 0) W4_STAMP(7);
            if (!W4_EX2) __syncthreads();
            if (rnd == 0) W4_STAMP(8);
            if (rnd == 0) {
                // the next tile's first U group (requested during the last chunk) must be home before this tile's stores enter the queue:
                // vmcnt counts stores too, and the counted waits of the next chunk assume only loads behind the ring words
                asm volatile("s_waitcnt vmcnt(0)" : "+v"(ur[0][0]), "+v"(ur[0][1]), "+v"(ur[0][2]));
            }
            if (rnd == 0) W4_STAMP(9);
            if (!chore && !(WINO4_EXP & 32)) {
                const Role g = role_of(rnd);
                f32x4 v[4];
                if (!spade) {
                    const float esc = ep_scale[g.col], ebi = ep_bias[g.col];
                    {
                        f32x4 z[6];
#pragma unroll
                        for (int a = 0; a < 6; a++) z[a] = exr[(a * 16) * 32 + g.t];   // ex[a][c16][fn]
#pragma unroll
                        for (int c = 0; c < 4; c++) {                        // row half (over a), one column at a time
                            float y0, y1, y2, y3;
                            w4_at(z[0][c], z[1][c], z[2][c], z[3][c], z[4][c], z[5][c], y0, y1, y2, y3);
                            v[0][c] = y0; v[1][c] = y1; v[2][c] = y2; v[3][c] = y3;
                        }
                    }
#pragma unroll
                    for (int r = 0; r < 4; r++) {
                        f32x4 w;
                        if (op_is_noise || late_noise) {                     // wave-uniform
                            const f32x4 nz = late_noise ? load4(nz_n, g.nzb + (unsigned)r * nzrow_b, g.cok && g.oyb + r < OHv) : op[r];
#pragma unroll
                            for (int e = 0; e < 4; e++) w[e] = fmaf(v[r][e], esc, fmaf(nz[e], ngain, ebi));
                        } else {
#pragma unroll
                            for (int e = 0; e < 4; e++) w[e] = fmaf(v[r][e], esc, ebi);
                        }
                        w = act4n(w);
                        if (res_n) w += op[r];
                        v[r] = w;
                    }
                    if (rnd < 3) request(rnd + 1, op);                       // before this round's stores
#pragma unroll
                    for (int r = 0; r < 4; r++) store4(g.ob + (unsigned)r * rstride_b, v[r], g.cok && g.oyb + r < OHv);
                    if (TAIL == W4_TAIL_STATS && stats_p) {
                        // this thread's 16 outputs of (cout, tile), then the 32 tiles of the workgroup tile (the 32 lanes of a wave half share the cout)
                        // (sum, M2 = sum of squared deviations from the set's own mean) pairs, merged pairwise with Chan's formula: E[x^2] - E[x]^2 on float32
                        // sums of raw squares (round 4) loses the variance of a plane whose |mean| >> std; the separate statistics pass is
                        // shifted-data too.  The count of a set follows from the geometry (whole 4-wide segments, rows inside the image).
                        float cnt = 0.f, s1 = 0.f;
                        bool okr[4];
#pragma unroll
                        for (int r = 0; r < 4; r++) {
                            okr[r] = seg_rows_full || (g.cok && g.oyb + r < OHv);
                            cnt += okr[r] ? 4.f : 0.f;
#pragma unroll
                            for (int e = 0; e < 4; e++) s1 += okr[r] ? v[r][e] : 0.f;
                        }
                        const float mt = cnt > 0.f ? s1 / cnt : 0.f;
                        float m2 = 0.f;
#pragma unroll
                        for (int r = 0; r < 4; r++)
#pragma unroll
                            for (int e = 0; e < 4; e++) { const float d = okr[r] ? v[r][e] - mt : 0.f; m2 = fmaf(d, d, m2); }
#pragma unroll
                        for (int m = 1; m < 32; m <<= 1) {
                            const float cb = __shfl_xor(cnt, m, 64), sb = __shfl_xor(s1, m, 64), qb = __shfl_xor(m2, m, 64);
                            const float ct = cnt + cb;
                            // delta = mean_b - mean_a, as (sb * cnt - s1 * cb) / (cnt * cb); M2 += delta^2 * cnt * cb / ct  (both sets non-empty)
                            const float num = sb * cnt - s1 * cb;
                            const float add = (cnt > 0.f && cb > 0.f) ? num * num / (cnt * cb * ct) : 0.f;
                            m2 = m2 + qb + add; s1 += sb; cnt = ct;
                        }
                        const int co = e_m0 + g.col;
                        if (g.fn == 0 && co < Coutv) {
                            float* dst = stats_p + (((int64_t)e_n * Coutv + co) * stats_T + stats_t) * 2;
                            dst[0] = s1; dst[1] = m2;
                        }
                    }
                } else {
                    // SPADE combine (networks.py:1715-1722): M-tile 0 rows are gamma, M-tile 1 rows beta of the same 32 channels;
                    // thread = (channel c8, tile, row pair rh):  y = (x - mean) * rstd * (1 + gamma) + beta
                    const int chl = 8 * rnd + g.c8;
                    const int rh = wave >> 2;                                // == g.rh, but on the scalar unit: the row selects below are branches, not v_cndmask pairs
                    const float mu = ep_scale[chl], rsd = ep_bias[chl];
                    f32x4 gb[2][2];                                          // [gamma | beta][row of the pair]
#pragma unroll
                    for (int gbi = 0; gbi < 2; gbi++) {
                        f32x4 z[6];
#pragma unroll
                        for (int a = 0; a < 6; a++) z[a] = exr[(a * 16 + gbi * 8 + g.c8) * 32 + g.fn];
#pragma unroll
                        for (int c = 0; c < 4; c++) {                        // over a: only this thread's two output rows
                            const float s12 = z[1][c] + z[2][c], d12 = z[1][c] - z[2][c], s34 = z[3][c] + z[4][c], d34 = z[3][c] - z[4][c];
                            if (rh) { gb[gbi][0][c] = fmaf(4.f, s34, s12); gb[gbi][1][c] = fmaf(8.f, d34, d12) + z[5][c]; }
                            else { gb[gbi][0][c] = z[0][c] + s12 + s34; gb[gbi][1][c] = fmaf(2.f, d34, d12); }
                        }
                    }
                    const float nmr = -mu * rsd;
#pragma unroll
                    for (int rr = 0; rr < 2; rr++) {
                        f32x4 w;
#pragma unroll
                        for (int e = 0; e < 4; e++) w[e] = fmaf(fmaf(op[rr][e], rsd, nmr), gb[0][rr][e] + 1.f, gb[1][rr][e]);
                        v[rr] = act4(w);
                    }
                    if (rnd < 3) request(rnd + 1, op);                       // before this round's stores
#pragma unroll
                    for (int rr = 0; rr < 2; rr++) store4(g.ob + (unsigned)rr * rstride_b, v[rr], g.cok && g.oyb + 2 * rh + rr < OHv);
                }
            }
            if (rnd == 0) W4_STAMP(10);
            if (!W4_EX2 || rnd < 3) __syncthreads();                     // single buffer: the exchange area is rewritten by the next round / the next transform;
                                                                         // two buffers: see write_ex (after round 3 the next tile's barrier A does it)
            W4_STAMP(11 + rnd);
        }
#if WINO4_EXP & 128
        if (tiles_done == 1 && blockIdx.x == 0 && (w4_fresh_lane() == 0)) {
            for (int i = 0; i < 15; i++) ((unsigned*)p.y)[wave * 16 + i] = stamp_lds[40 * wave + i] - stamp_lds[40 * wave];
            if (X3) for (int i = 0; i < 19; i++) ((unsigned*)p.y)[192 + wave * 20 + i] = stamp_lds[40 * wave + 16 + i] - stamp_lds[40 * wave];
        }
        tiles_done++;
#endif
        if (!has_next) break;
        tile = next;
        par ^= 1;
        tiles_run++;
    }
}

template <int MODE, int TAIL, bool X3 = false>
int launch_wino4_mode(const ConvParams& p0, hipStream_t s) {
    ConvParams p = p0;
    p.tilesX = (p.OW + 63) / 64;
    p.tilesY = (p.OH + 7) / 8;
    p.mblocks = p.CoutP / 64;
    const int64_t tiles = (int64_t)p.N * p.tilesX * p.tilesY * p.mblocks;
    if (tiles > 0x7fffffffLL) return PG_ERR_TOO_LARGE;
    p.total_tiles = (int)tiles;
    const int cin_loop = ((p.Cin + W4_KC - 1) / W4_KC) * W4_KC;
    const size_t lds = ((size_t)W4_RAW + W4_V + 2 * cin_loop + 256 + 64 + ((WINO4_EXP & 128) ? 12 * 40 : 0) + 2 * W4_NDMA * 256) * sizeof(float);
    if ((int64_t)36 * cin_loop * p.CoutP * (X3 ? 6 : 4) > 0x7fffffffLL) return PG_ERR_TOO_LARGE;      // the U stream uses 32-bit byte offsets
    {   // the tail addresses one image of y / residual / spade_x with 32-bit byte offsets
        const int64_t ext = 1 + (int64_t)(p.f.spade_x ? p.Cout / 2 - 1 : p.Cout - 1) * p.ys[1] + (int64_t)(p.OH - 1) * p.ys[2] + (int64_t)(p.OW - 1) * p.ys[3];
        if (ext * 4 > 0xffffffffLL || (int64_t)p.OH * p.OW * 4 > 0xffffffffLL) return PG_ERR_TOO_LARGE;
    }
    if (lds > 160 * 1024) return PG_ERR_UNSUPPORTED;
    if (p.pad_x != 1 && p.pad_x != 3) return PG_ERR_UNSUPPORTED;    // W % 4 == 0 and OW % 4 == 0 together leave odd paddings only; 0 / 4 had their own (never exercised) staging arithmetic: removed in round 5
    // the tail moves 16-byte row segments: unit x stride, every other stride and OW a multiple of 4, 16-byte aligned bases
    if (p.ys[3] != 1 || ((p.ys[0] | p.ys[1] | p.ys[2] | p.f.noise_batch_stride) & 3) != 0 || (p.OW & 3) != 0 ||
        ((((uintptr_t)p.y) | ((uintptr_t)p.f.noise) | ((uintptr_t)p.f.residual) | ((uintptr_t)p.f.spade_x)) & 15) != 0) return PG_ERR_UNSUPPORTED;
    const int64_t blocks = tiles < (int64_t)num_cu() ? tiles : (int64_t)num_cu();
    static PerDeviceOnce lds_attr;
    const hipError_t e = lds_attr.run([] { return hipFuncSetAttribute((const void*)conv2d_wino4<MODE, TAIL, X3>, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024); });
    if (e != hipSuccess) return (int)e;
    hipLaunchKernelGGL((conv2d_wino4<MODE, TAIL, X3>), dim3((unsigned)blocks), dim3(768), lds, s, p);
    return launch_status();
}

}  // namespace pgconv
