// Winograd F(4x4, 3x3) fp32 convolution for gfx950, TWO workgroups per CU (round 4): the same algorithm, transforms and numerics as
// conv2d_wino4.h on v_mfma_f32_16x16x4_f32 instead of v_mfma_f32_32x32x2_f32 (same 64 FLOP/clk/SIMD, a quarter of the accumulator
// tile).  Why: conv2d_wino4.h is ONE 12-wave workgroup per CU whose phases are separated by barriers -- nothing multiplies during its
// transform phase (13 % of a K chunk), its tail (10-24 % of a tile) and its barriers: 0.56 of the fp32 matrix peak.  Co-execution of
// vector and matrix work across waves is real on this part (tools/probes/mfma_valu_coexec.hip: an MFMA wave is not slowed by a
// partner's VALU stream) but one workgroup cannot use it: every attempt to pipeline roles inside it lost to the slowest role
// (DESIGN.md section 3.0).  Two independent workgroups per CU can: one's transform / tail / barrier waits run beside the other's GEMM
// phase -- what conv2d_wino.h (F(2x2), two workgroups per CU) gets 18-22 % from.  That needs <= 128 VGPRs and <= 80 KB of LDS per
// workgroup, which the 32x32 accumulator tiles of conv2d_wino4.h (96 of 168 registers) cannot give.
//
// Mapping
//   * workgroup = 512 threads = 8 waves, 128 VGPRs, 77-81 KB LDS -> two per CU, four waves per SIMD; tile = 64 couts x (8 output rows x
//     32 output cols) = 16 tiles of 4x4 outputs, MFMA column n = 2 * tile_x + tile_y; input transform amortised over 64 couts as before.
//   * GEMM phase: wave w = (cb = w & 3, ah = w >> 2) owns cout block cb (16 couts) and transform-domain rows a = 3 ah .. 3 ah + 2, i.e.
//     18 xi values = 18 accumulators of 4 registers.  Per 16-channel chunk and xi: ONE 16-byte A word from the pre-transformed weight
//     stream (lane (m, kq) holds U[xi][cout m][channels 4 j + kq], j = 0..3) and ONE 16-byte B word from V in LDS (lane (kq, n) holds
//     V[xi][channels 4 j + kq][tile n]) feed the four K steps j of v_mfma_f32_16x16x4_f32; two xi are interleaved so that no MFMA
//     waits for its own accumulator (40-cycle dependent latency, 32-cycle issue).  72 MFMAs, 18 global + 18 LDS 16-byte reads per wave
//     and chunk.  The price of the 16-wide tile: twice the weight stream per output of conv2d_wino4.h (it comes from L2).
//   * transform phase: waves 0-3 (wave = kq, lane = 4 * tile + j -> channel 4 j + kq): one 6x6 patch per thread from the raw halo
//     tile [16][10 rows][40 cols] (channel stride 404 floats: conflict-free ds_read_b128), 36 ds_write_addtid_b32 into
//     V[xi][kq][tile][j] -- exactly the B word layout, 256 contiguous bytes per wave-instruction.  Waves 4-7 have nothing to do then:
//     the other workgroup of the CU has.
//   * halo: 16-byte LDS-DMA, 26 wave-instructions per chunk spread over all eight waves (3-4 each), requested right after the
//     transform has released the raw buffer; per-thread byte offsets in an LDS map that is rebuilt only when a tile is (or follows)
//     an edge tile; zero padding = the buffer range check on a sentinel offset.
//   * A words: inline-asm loads two xi pairs ahead with hand-counted vmcnt (the compiler cannot count the asm DMA).
//   * tail: column half of the inverse transform in registers (a wave holds all six b of its three a), one exchange with the partner
//     wave (cb, 1 - ah) through the V area in two rounds (a lane finishes couts 2 ah and 2 ah + 1 of its four), row half, fused
//     epilogue of conv2d_wino4.h on 16-byte row segments.  SPADE mode: gamma / beta rows are packed as adjacent cout pairs
//     (pack_spade_gamma_beta(winograd=3)), so round 0 finishes gamma and round 1 beta of the SAME channel in the same lane.
#pragma once
#include <cstdlib>
#include <type_traits>
#include "conv2d_wino4.h"

#ifndef WINO4B_EXP
#define WINO4B_EXP 0     // dev ablations (results wrong by design): 1 no U loads, 2 no transform, 4 no tail, 8 no halo DMA; timing experiments that stay correct:
                         // 16 s_setprio 1 around the GEMM phase, 32 / 64 only waves 4-7 / 0-3 request the halo, 128 the halo is requested EARLY (right after the transform),
                         // 256 two A-word pairs ahead instead of three, 1024 / 2048 s_setprio 2 in the transform / the tail
#endif
#ifndef WINO4B_LOADERS
#define WINO4B_LOADERS ((WINO4B_EXP & 32) ? 1 : (WINO4B_EXP & 64) ? 2 : 0)      // which waves request the halo: 0 all eight, 1 waves 4-7, 2 waves 0-3
#endif

namespace pgconv {

constexpr int B4_KC = 16;                          // input channels per chunk
constexpr int B4_ROWS = 10;                        // halo rows of an 8-row output tile
constexpr int B4_LROW = 40;                        // floats per LDS halo row: global columns [ox0 - 4, ox0 + 36)
constexpr int B4_CHW = B4_ROWS * B4_LROW / 4 + 1;  // 101 sixteen-byte words per channel: one pad word makes the channel stride 404 floats = 4 (mod 16)
constexpr int B4_CHF = 4 * B4_CHW;
constexpr int B4_NWORDS = B4_KC * B4_CHW;          // 1616 words per chunk
constexpr int B4_NINST = (B4_NWORDS + 63) / 64;    // 26 DMA wave-instructions per chunk: instruction q = NL i + lw for loader wave lw of NL
constexpr int B4_NL = WINO4B_LOADERS == 0 ? 8 : 4;  // loader waves
constexpr int B4_NI = (B4_NINST + B4_NL - 1) / B4_NL;      // instructions per loader wave (the last round is partial)
// Measured defaults (round 4, same-box A/B of the switches above): the halo is requested LATE in the GEMM phase, behind the chunk's last A
// words, and waited for at the top of the next chunk -- requested right after the transform its requests sit in front of the A words in the
// in-order vector-memory queue and every wave stalls on an HBM round trip in the middle of its MFMAs (-7 %); three A-word pairs ahead (-2 %).
constexpr int B4_AT = (WINO4B_EXP & 128) ? -1 : (WINO4B_EXP & 65536) ? 1 : (WINO4B_EXP & 131072) ? 3 : 8;      // the halo is requested after the MFMAs of pair AT (-1: right after the transform; 8: behind the last A words)
constexpr bool B4_LATE = B4_AT == 8;                       // bit 128: request the halo right after the transform instead; 65536 / 131072: after pair 1 / 3
constexpr int B4_DP = (WINO4B_EXP & 512) ? 4 : (WINO4B_EXP & 256) ? 2 : 3;          // A-word pairs requested ahead of the one being multiplied (bit 256: two)
constexpr int B4_RAW = B4_NINST * 256 + 16;        // floats (+ room for the 0..3 float shift that aligns the patches)
constexpr int B4_V = 36 * 256;                     // V[xi][kq 4][tile 16][j 4]  /  exchange [wave 8][a' 3][lane 64][4]
constexpr int B4_UWAVE = 18 * 1024;                // bytes of A words per (wave, chunk)
constexpr unsigned B4_SENT = 0x80000000u;

typedef float f32x4v __attribute__((ext_vector_type(4)));

template <int A>
__device__ __forceinline__ void w4b_write_row(unsigned m0_base, float v0, float v1, float v2, float v3, float v4, float v5) {
    unsigned keep;
    asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %1\n\ts_nop 0\n\t"
                 "ds_write_addtid_b32 %2 offset:%8\n\tds_write_addtid_b32 %3 offset:%9\n\tds_write_addtid_b32 %4 offset:%10\n\t"
                 "ds_write_addtid_b32 %5 offset:%11\n\tds_write_addtid_b32 %6 offset:%12\n\tds_write_addtid_b32 %7 offset:%13\n\t"
                 "s_mov_b32 m0, %0"
                 : "=&s"(keep)
                 : "s"(m0_base), "v"(v0), "v"(v1), "v"(v2), "v"(v3), "v"(v4), "v"(v5),
                   "n"((6 * A + 0) * 1024), "n"((6 * A + 1) * 1024), "n"((6 * A + 2) * 1024),
                   "n"((6 * A + 3) * 1024), "n"((6 * A + 4) * 1024), "n"((6 * A + 5) * 1024)
                 : "memory");
}

template <int MODE, int TAIL>
__global__ __launch_bounds__(512, 4) void conv2d_wino4b(ConvParams p) {
    extern __shared__ __attribute__((aligned(16))) float smem[];
    const int cin_loop = ((p.Cin + B4_KC - 1) / B4_KC) * B4_KC;
    const int nchunks = cin_loop / B4_KC;
    float* raw = smem;
    float* V = smem + B4_RAW;
    float* cs0 = V + B4_V;                       // prologue scale of two consecutive tiles [2][cin_loop]
    float* ep0 = cs0 + 2 * cin_loop;             // epilogue scale / bias of two consecutive tiles [2][64 + 64]
    unsigned* gm = (unsigned*)(ep0 + 256);       // halo gather map of the current tile [NI][NL * 64] (byte offsets; interior tiles: relative to the first halo sample)

    const int wave = __builtin_amdgcn_readfirstlane((int)threadIdx.x >> 6);
    const unsigned smem_b = __builtin_amdgcn_readfirstlane(lds_offset(smem));
    if (smem_b + (unsigned)(B4_RAW + B4_V) * 4u > 65536u) __builtin_trap();   // ds_write_addtid_b32: 16-bit base + 16-bit offset
    const int HW = p.H * p.W;
    const int total = p.total_tiles;
    const int q8 = total >> 3, r8 = total & 7;
    const int cb = wave & 3, ah = wave >> 2;     // GEMM role: cout block, transform-domain row half
    const bool loader = WINO4B_LOADERS == 0 ? true : (WINO4B_LOADERS == 1 ? wave >= 4 : wave < 4);
    const int lw = WINO4B_LOADERS == 0 ? wave : (wave & 3);                       // loader wave index
    const int nd = !loader ? 0 : (B4_NL * (B4_NI - 1) + lw < B4_NINST ? B4_NI : B4_NI - 1);      // halo DMA instructions of this wave per chunk
    const int sh = p.pad_x & 3;                  // float shift of the staged tile: patch column 0 of tile_x lands on LDS column 4 tile_x + cbase
    constexpr int cbase = 4;                     // = 4 - pad_x + sh for the paddings the launcher admits (1, 3): a multiple of 4 -> 16-byte aligned patch reads

    int n = 0, oy0 = 0, ox0 = 0, m0 = 0;
    bool edge = false, map_is_interior = false;
    i32x4 xrsrc;

    auto prep_tile = [&](int tile, float* cs) {
        const int xcd = tile & 7;
        int L = (xcd < r8 ? xcd * (q8 + 1) : r8 * (q8 + 1) + (xcd - r8) * q8) + (tile >> 3);
        const int mb = L % p.mblocks; L /= p.mblocks;
        const int tx = L % p.tilesX; L /= p.tilesX;
        const int ty = L % p.tilesY;
        n = L / p.tilesY;
        oy0 = ty * 8; ox0 = tx * 32; m0 = mb * 64;
        if (MODE == 1) {
            const float* in_scale = p.f.in_scale ? p.f.in_scale + (int64_t)n * p.Cin : nullptr;
            for (int c = (int)threadIdx.x; c < cin_loop; c += 512) cs[c] = ((in_scale && c < p.Cin) ? ld_opaque(in_scale + c) : 1.f) * p.f.in_gain;
        }
        const int gy0 = oy0 - p.pad_y, gx0 = ox0 - 4;
        edge = !(gy0 >= 0 && gy0 + B4_ROWS <= p.H && gx0 >= 0 && gx0 + B4_LROW <= p.W);           // wave-uniform
        if ((edge || !map_is_interior) && loader) {          // (an interior tile after an interior tile: the map is tile-independent)
            const int lane = w4_fresh_lane();
#pragma unroll
            for (int i = 0; i < B4_NI; i++) {
                const int q = B4_NL * i + lw;
                if (q < B4_NINST) {
                    const int f = 64 * q + lane;
                    const int c = f / B4_CHW, rem = f - c * B4_CHW;
                    const int row = rem / 10, wd = rem - row * 10;
                    const bool pad = rem == B4_CHW - 1 || f >= B4_NWORDS;
                    unsigned off;
                    if (edge) {
                        const int gy = gy0 + row, gx = gx0 + 4 * wd;
                        const bool ok = !pad && gy >= 0 && gy < p.H && gx >= 0 && gx + 4 <= p.W;
                        off = ok ? (unsigned)(c * HW + gy * p.W + gx) * 4u : B4_SENT;
                    } else {
                        off = pad ? B4_SENT : (unsigned)(c * HW + row * p.W + 4 * wd) * 4u;
                    }
                    gm[i * (B4_NL * 64) + lw * 64 + lane] = off;      // read back by the writing thread only: no barrier needed
                }
            }
        }
        if (edge || !map_is_interior) map_is_interior = !edge;
        const int shift = edge ? 0 : (gy0 * p.W + gx0) * 4;
        const uint64_t base = (uint64_t)(uintptr_t)(p.x + (int64_t)n * p.Cin * HW) + (uint64_t)(int64_t)shift;
        xrsrc[0] = __builtin_amdgcn_readfirstlane((int)(unsigned)base);
        xrsrc[1] = __builtin_amdgcn_readfirstlane((int)(unsigned)(base >> 32) & 0xffff);
        xrsrc[2] = p.Cin * HW * 4 - shift;                               // same absolute end: channels beyond Cin read as zero
        xrsrc[3] = 0x00020000;
    };

    auto issue_chunk = [&](int c0) {
#if !(WINO4B_EXP & 8)
        if (loader) {
            const unsigned xs_b = smem_b + (unsigned)sh * 4u;
            const int soff = c0 * HW * 4;
            const int lt = lw * 64 + w4_fresh_lane();
            unsigned off[B4_NI];
#pragma unroll
            for (int i = 0; i < B4_NI - 1; i++) off[i] = gm[i * (B4_NL * 64) + lt];
            if (nd == B4_NI) off[B4_NI - 1] = gm[(B4_NI - 1) * (B4_NL * 64) + lt];
#pragma unroll
            for (int i = 0; i < B4_NI - 1; i++) dma_dwordx4_buf(xrsrc, xs_b + (unsigned)(B4_NL * i + lw) * 1024u, off[i], soff);
            if (nd == B4_NI) dma_dwordx4_buf(xrsrc, xs_b + (unsigned)(B4_NL * (B4_NI - 1) + lw) * 1024u, off[B4_NI - 1], soff);
        }
#endif
    };

    f32x4v acc[18];                                  // [6 a' + b]; never zeroed: chunk 0 of a tile feeds the constant 0 as C
    const f32x4v zero4 = {0.f, 0.f, 0.f, 0.f};

    // A-operand stream of this wave: [m-block][cb][ah][chunk][e 18][lane 64][j 4] floats, walked strictly forwards inside a tile
    unsigned pa = 0;
    auto a_reset = [&](int m0_) { pa = (unsigned)((((m0_ >> 6) * 4 + cb) * 2 + ah) * nchunks) * (unsigned)B4_UWAVE + (unsigned)w4_fresh_lane() * 16u; };
    auto load_pair = [&](f32x4v (&dst)[2]) {
#if !(WINO4B_EXP & 1)
        asm volatile("global_load_dwordx4 %0, %1, %2" : "=v"(dst[0]) : "v"(pa), "s"(p.wp));
        asm volatile("global_load_dwordx4 %0, %1, %2 offset:1024" : "=v"(dst[1]) : "v"(pa), "s"(p.wp));
#endif
        if (!(WINO4B_EXP & 4096)) pa += 2048u;                           // (ablation 4096: every A load hits the same 2 KB -- the stream without its L2 traffic)
    };
    // wait for a pair's two words; `base` = loads issued after them that may still be in flight, plus this wave's DMA when `dma`
    // wait for a pair's two words; `base` = A loads issued after them that may still be in flight (0, 2 or 4), plus this wave's DMA when `dma`
    auto wait_pair = [&](f32x4v (&g)[2], const int base, const bool dma, const int nd_now) __attribute__((always_inline)) {
#define B4_WAIT(N) asm volatile("s_waitcnt vmcnt(%0)" :: "n"(N))
#define B4_WAIT_BASE(EXTRA) { if (base == 0) B4_WAIT(0 + (EXTRA)); else if (base == 2) B4_WAIT(2 + (EXTRA)); else B4_WAIT(4 + (EXTRA)); }
        if (!dma || nd_now == 0) B4_WAIT_BASE(0)
        else if (nd_now == B4_NI) B4_WAIT_BASE(B4_NI)
        else B4_WAIT_BASE(B4_NI - 1)
#undef B4_WAIT_BASE
#undef B4_WAIT
        asm volatile("" : "+v"(g[0]), "+v"(g[1]));
    };

    typedef const __attribute__((address_space(4))) ConvParams* kernarg_t;
    auto fresh_args = [&]() {
        kernarg_t a = (kernarg_t)__builtin_amdgcn_kernarg_segment_ptr();
        asm volatile("" : "+s"(a));
        return a;
    };

#if WINO4B_EXP & 32768
    // dev build: s_memtime stamps of this workgroup's third tile, per wave, kept in LDS and dumped into y (host side: tools/wino4b_stamps.py)
    unsigned long long* stamps_lds = (unsigned long long*)(gm + B4_NI * B4_NL * 64);      // [8 waves][64]
    int tiles_done = 0;
#define B4_STAMP(i) do { if (tiles_done == 2 && w4_fresh_lane() == 0) stamps_lds[wave * 64 + (i)] = __builtin_amdgcn_s_memtime(); } while (0)
#else
#define B4_STAMP(i) do { } while (0)
#endif
    int tile = blockIdx.x;
    int par = 0;
    prep_tile(tile, cs0);
    issue_chunk(0);
    dma_wait_all();
    {   // Stagger: the two workgroups of a CU start together and run the same program -- left alone they stay in lockstep (both transform, both
        // multiply, both finish at the same time: measured, every phase cost its stand-alone time).  The one whose LDS allocation does not start at
        // 0 (HW_REG_LDS_ALLOC.LDS_BASE) begins a fraction of a chunk period later, so that one's vector phases meet the other's matrix phase.
        constexpr int SLEEP = ((WINO4B_EXP & 8192) ? 40 : 0) + ((WINO4B_EXP & 16384) ? 80 : 0);
        if (SLEEP) {
            const unsigned lds_base = __builtin_amdgcn_s_getreg((7 << 11) | (0 << 6) | 6);
            if (lds_base != 0) __builtin_amdgcn_s_sleep(SLEEP);
        }
    }
    while (true) {
        int e_n = n, e_oy0 = oy0, e_ox0 = ox0, e_m0 = m0;
        bool has_next = false;
        int next = tile;
        const float* cs_cur = cs0 + par * cin_loop;
        float* ep_scale = ep0 + par * 128;
        float* ep_bias = ep_scale + 64;
        a_reset(m0);

        auto chunk = [&](const int k, auto first_tag) __attribute__((always_inline)) {
            constexpr bool FIRST = decltype(first_tag)::value;
            if (!(WINO4B_EXP & 262144)) {  // The two workgroups of a CU take turns at being the favoured one (every 65 k cycles): issue arbitration is by priority, then AGE,
                                           // and the stamps showed the older workgroup finishing a tile in 100 k cycles against 152 k for the younger one (-2.5 ... -3.3 % per launch)
                const unsigned second = __builtin_amdgcn_s_getreg((7 << 11) | (0 << 6) | 6) != 0;
                if ((((unsigned)(__builtin_amdgcn_s_memtime() >> 16)) ^ second) & 1) __builtin_amdgcn_s_setprio(1); else __builtin_amdgcn_s_setprio(0);
            }
            if (k < 8) B4_STAMP(6 * k + 0);
            if (B4_LATE) dma_wait_all();                                 // (late requests: nothing else waited for them)
            if (k < 8) B4_STAMP(6 * k + 1);
            __syncthreads();                                            // A: raw(k) has landed (every wave waited for its own requests in the
                                                                        //    previous GEMM phase), V is free (everybody is past GEMM(k-1) / the tail)
            f32x4v ua[B4_DP][2];
            if (k < 8) B4_STAMP(6 * k + 2);
#if !(WINO4B_EXP & 2)
            if (wave < 4) {
                if (WINO4B_EXP & 1024) __builtin_amdgcn_s_setprio(2);
                // transform role: kq = wave, lane = 4 * tile + j -> channel 4 j + kq; tile n = 2 tile_x + tile_y
                const int lane = w4_fresh_lane(), nt = lane >> 2, j = lane & 3;
                const int c = 4 * j + wave;
                const float* rb = raw + c * B4_CHF + (4 * (nt & 1)) * B4_LROW + 4 * (nt >> 1) + cbase;
                float d[6][6];
#pragma unroll
                for (int r = 0; r < 6; r++) {
                    const f32x4v lo = *(const f32x4v*)(rb + r * B4_LROW), hi = *(const f32x4v*)(rb + r * B4_LROW + 4);
                    d[r][0] = lo[0]; d[r][1] = lo[1]; d[r][2] = lo[2]; d[r][3] = lo[3]; d[r][4] = hi[0]; d[r][5] = hi[1];
                }
                float sc = 1.f;
                if (MODE == 1) sc = cs_cur[k * B4_KC + c];
#pragma unroll
                for (int jj = 0; jj < 6; jj++)                           // columns: over the patch rows
                    w4_bt(d[0][jj], d[1][jj], d[2][jj], d[3][jj], d[4][jj], d[5][jj], d[0][jj], d[1][jj], d[2][jj], d[3][jj], d[4][jj], d[5][jj]);
                const unsigned vbase = smem_b + (unsigned)(B4_RAW * 4 + wave * 256);
#define B4_ROW(A) { float v0, v1, v2, v3, v4, v5; \
                    w4_bt(d[A][0], d[A][1], d[A][2], d[A][3], d[A][4], d[A][5], v0, v1, v2, v3, v4, v5); \
                    if (MODE == 1) { v0 *= sc; v1 *= sc; v2 *= sc; v3 *= sc; v4 *= sc; v5 *= sc; } \
                    w4b_write_row<A>(vbase, v0, v1, v2, v3, v4, v5); }
                B4_ROW(0) B4_ROW(1) B4_ROW(2) B4_ROW(3) B4_ROW(4) B4_ROW(5)
#undef B4_ROW
                asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
                if (WINO4B_EXP & 1024) __builtin_amdgcn_s_setprio(0);
            } else if (k == 0 && wave == 7) {
                // epilogue constants of THIS tile (before anything enters this wave's queue: ld_opaque waits for vmcnt(0))
                const auto& qa = *fresh_args();
                const int tc = w4_fresh_lane();
                if (TAIL == W4_TAIL_SPADE || (TAIL == W4_TAIL_ANY && qa.f.spade_x)) {
                    if (tc < 32) {
                        const int ch = (m0 >> 1) + tc;
                        ep_scale[tc] = ld_opaque(qa.f.spade_mean + n * (qa.Cout >> 1) + ch);
                        ep_bias[tc] = ld_opaque(qa.f.spade_rstd + n * (qa.Cout >> 1) + ch);
                    }
                } else {
                    const int co = m0 + tc;
                    const bool ok = co < qa.Cout;
                    const int cc = ok ? co : 0;
                    const float scv = qa.f.out_scale ? ld_opaque(qa.f.out_scale + (int64_t)n * qa.Cout + cc) : 1.f;
                    const float bi = qa.f.bias ? ld_opaque(qa.f.bias + cc) : 0.f;
                    ep_scale[tc] = ok ? scv * qa.f.gain : 0.f;           // gain folded: see conv2d_wino4.h
                    ep_bias[tc] = ok ? bi * qa.f.gain : 0.f;
                }
            }
#endif
#pragma unroll
            for (int d = 0; d < B4_DP; d++) load_pair(ua[d]);            // the first pairs of this chunk (before the DMA: their waits count it)
            if (k < 8) B4_STAMP(6 * k + 3);
            __syncthreads();                                            // B: V(k) complete, raw free
            if (k < 8) B4_STAMP(6 * k + 4);

            // ---- request the next chunk (of this tile, or the first of the next tile)
            int nd_now = nd;
            auto request_next = [&]() __attribute__((always_inline)) {
                if (k + 1 < nchunks) {
                    issue_chunk((k + 1) * B4_KC);
                } else {
                    e_n = n; e_oy0 = oy0; e_ox0 = ox0; e_m0 = m0;
                    next = tile + gridDim.x;
                    has_next = next < total;
                    if (has_next) { prep_tile(next, cs0 + (par ^ 1) * cin_loop); issue_chunk(0); }
                    else nd_now = 0;
                }
            };
            if (B4_AT < 0) request_next();

            // ---- GEMM phase: 9 pairs of xi, four K steps each
            if (WINO4B_EXP & 16) __builtin_amdgcn_s_setprio(1);
            const f32x4v* vp = (const f32x4v*)(V + 18 * ah * 256) + w4_fresh_lane();
            f32x4v vb[2][2];
            vb[0][0] = vp[0]; vb[0][1] = vp[64];
#pragma unroll
            for (int ep = 0; ep < 9; ep++) {
                const int s = ep & 1, su = ep % B4_DP;
                if (ep + 1 < 9) { vb[s ^ 1][0] = vp[(2 * ep + 2) * 64]; vb[s ^ 1][1] = vp[(2 * ep + 3) * 64]; }
                // issued so far: pairs 0 .. min(ep + DP - 1, 8); the DMA sits between pair DP - 1 and pair DP (LATE: behind pair 8)
                wait_pair(ua[su], 2 * ((ep + B4_DP - 1 < 8 ? ep + B4_DP - 1 : 8) - ep), B4_LATE ? ep + B4_DP - 1 >= 8 : (ep > B4_AT && ep <= B4_AT + B4_DP), nd_now);
                __builtin_amdgcn_sched_barrier(0);
#pragma unroll
                for (int j = 0; j < 4; j++) {
                    acc[2 * ep] = __builtin_amdgcn_mfma_f32_16x16x4f32(ua[su][0][j], vb[s][0][j], (FIRST && j == 0) ? zero4 : acc[2 * ep], 0, 0, 0);
                    acc[2 * ep + 1] = __builtin_amdgcn_mfma_f32_16x16x4f32(ua[su][1][j], vb[s][1][j], (FIRST && j == 0) ? zero4 : acc[2 * ep + 1], 0, 0, 0);
                }
                if (ep + B4_DP < 9) load_pair(ua[su]);
                if (B4_LATE && ep + B4_DP == 8) request_next();          // behind the chunk's last A words
                if (!B4_LATE && B4_AT >= 0 && ep == B4_AT) request_next();
                __builtin_amdgcn_sched_barrier(0);
            }
            if (WINO4B_EXP & 16) __builtin_amdgcn_s_setprio(0);
            if (k < 8) B4_STAMP(6 * k + 5);
            if (WINO4B_EXP & 1) dma_wait_all();                          // (ablation builds: the counted waits assume the A loads exist)
        };
        chunk(0, std::true_type{});
#pragma unroll 1
        for (int k = 1; k < nchunks; k++) chunk(k, std::false_type{});

#if WINO4B_EXP & 4
        { float sm = 0.f;
          for (int j = 0; j < 18; j++) for (int k = 0; k < 4; k++) sm += acc[j][k];
          if (sm == 12345.678f) p.y[w4_fresh_lane()] = sm; }
        if (!has_next) break;
        tile = next; par ^= 1;
        continue;
#endif
        // ---- inverse transform + fused epilogue
        const auto& qa = *fresh_args();
        const bool spade = TAIL == W4_TAIL_SPADE ? true : (TAIL == W4_TAIL_ANY ? qa.f.spade_x != nullptr : false);
        const float gain = qa.f.gain, slope = act_slope(qa.f.act, qa.f.alpha);
        const float cl = qa.f.clamp >= 0.f ? qa.f.clamp : __builtin_inff();
        const bool need_act = slope != 1.f, need_clamp = qa.f.clamp >= 0.f;      // wave-uniform
        const int OHv = qa.OH, OWv = qa.OW, Coutv = qa.Cout;
        const float ngain = qa.f.noise_gain * (spade ? 1.f : gain);
        const int64_t img_off = (int64_t)e_n * qa.ys[0];
        float* y_n = qa.y + img_off;
        const float* res_n = ((TAIL == W4_TAIL_ANY || TAIL == W4_TAIL_RES) && qa.f.residual) ? qa.f.residual + img_off : nullptr;
        const float* spx_n = spade ? qa.f.spade_x + img_off : nullptr;
        const float* nz_n = ((TAIL == W4_TAIL_ANY || TAIL == W4_TAIL_NOISE) && qa.f.noise) ? qa.f.noise + (int64_t)e_n * qa.f.noise_batch_stride : nullptr;
        const unsigned cstride_b = (unsigned)qa.ys[1] * 4u, rstride_b = (unsigned)qa.ys[2] * 4u, nzrow_b = (unsigned)OWv * 4u;
        auto act4n = [&](f32x4v v) {                                     // gain already applied through the per-cout constants
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
        auto act4g = [&](f32x4v v) {                                     // SPADE tail: gain here
            const float g_pos = gain, g_neg = gain * slope;
            if (need_act || need_clamp || gain != 1.f) {
#pragma unroll
                for (int e = 0; e < 4; e++) v[e] = __builtin_amdgcn_fmed3f(fmaxf(v[e] * g_pos, v[e] * g_neg), -cl, cl);
            }
            return v;
        };

        // column half (over b) in registers: 72 accumulator registers become 48
        f32x4v mp[3][4];                                                 // [a'][cout i of the lane's four] = the four output columns
#pragma unroll
        for (int a = 0; a < 3; a++)
#pragma unroll
            for (int i = 0; i < 4; i++) {
                float y0, y1, y2, y3;
                w4_at(acc[6 * a + 0][i], acc[6 * a + 1][i], acc[6 * a + 2][i], acc[6 * a + 3][i], acc[6 * a + 4][i], acc[6 * a + 5][i], y0, y1, y2, y3);
                mp[a][i] = (f32x4v){y0, y1, y2, y3};
            }
        B4_STAMP(48);
        __syncthreads();                                                 // the exchange area IS the V buffer: every wave must be past its last GEMM reads
        B4_STAMP(49);
        f32x4v* ex = (f32x4v*)V;
        const int lane = w4_fresh_lane();
        const int q = lane >> 4, nt = lane & 15;
        const int oyb = e_oy0 + 4 * (nt & 1), oxb = e_ox0 + 4 * (nt >> 1);
        const bool full = e_oy0 + 8 <= OHv && e_ox0 + 32 <= OWv && e_m0 + 64 <= Coutv;      // wave-uniform: no predicates needed
        // (the row half `ah` of this wave as a compile-time constant: two copies of the rounds instead of ~100 v_cndmask per tile)
        auto finish = [&](auto ah_tag) __attribute__((always_inline)) {
        constexpr int AH = decltype(ah_tag)::value;
        f32x4v gam[4];                                                   // SPADE: round 0's result (gamma) waits for round 1's (beta)
#pragma unroll
        for (int r = 0; r < 2; r++) {
            // a lane finishes couts 2 ah + r of its four; the three a' rows of cout 2 (1 - ah) + r go to the partner wave (cb, 1 - ah)
#pragma unroll
            for (int a = 0; a < 3; a++) ex[(wave * 3 + a) * 64 + lane] = AH ? mp[a][r] : mp[a][2 + r];
            __syncthreads();
            f32x4v z[6];
#pragma unroll
            for (int a = 0; a < 3; a++) {
                const f32x4v got = ex[((wave ^ 4) * 3 + a) * 64 + lane];
                const f32x4v own = AH ? mp[a][2 + r] : mp[a][r];
                z[a] = AH ? got : own;                                   // rows a = 0..2 come from the ah = 0 wave, 3..5 from the ah = 1 wave
                z[3 + a] = AH ? own : got;
            }
            if (r == 0) __syncthreads();                                 // round 1 overwrites the area
            f32x4v v[4];
#pragma unroll
            for (int c = 0; c < 4; c++) {                                // row half (over a), one column at a time
                float y0, y1, y2, y3;
                w4_at(z[0][c], z[1][c], z[2][c], z[3][c], z[4][c], z[5][c], y0, y1, y2, y3);
                v[0][c] = y0; v[1][c] = y1; v[2][c] = y2; v[3][c] = y3;
            }
            const int col = 16 * cb + 4 * q + 2 * AH + r;                // row of the 64-cout block
            if (!spade) {
                const int co = e_m0 + col;
                const bool cok = co < Coutv && oxb < OWv;               // OW % 4 == 0: a 4-pixel segment is inside or outside as a whole
                const unsigned ob = (unsigned)oyb * rstride_b + (unsigned)oxb * 4u + (unsigned)(co < Coutv ? co : Coutv - 1) * cstride_b;
                const unsigned nzb = ((unsigned)oyb * (unsigned)OWv + (unsigned)oxb) * 4u;
                const float esc = ep_scale[col], ebi = ep_bias[col];
#pragma unroll
                for (int rr = 0; rr < 4; rr++) {
                    const bool ok = full || (cok && oyb + rr < OHv);
                    f32x4v w;
                    if (nz_n) {                                          // wave-uniform
                        const f32x4v nz = ok ? *(const f32x4v*)((const char*)nz_n + nzb + (unsigned)rr * nzrow_b) : zero4;
#pragma unroll
                        for (int e = 0; e < 4; e++) w[e] = fmaf(v[rr][e], esc, fmaf(nz[e], ngain, ebi));
                    } else {
#pragma unroll
                        for (int e = 0; e < 4; e++) w[e] = fmaf(v[rr][e], esc, ebi);
                    }
                    w = act4n(w);
                    if (res_n) { if (ok) w += *(const f32x4v*)((const char*)res_n + ob + (unsigned)rr * rstride_b); }
                    if (ok) *(f32x4v*)((char*)y_n + ob + (unsigned)rr * rstride_b) = w;
                }
            } else if (r == 0) {
#pragma unroll
                for (int rr = 0; rr < 4; rr++) gam[rr] = v[rr];
            } else {
                // SPADE combine (networks.py:1715-1722): cout rows (2 c', 2 c' + 1) of the block are (gamma, beta) of channel c':
                //   y = (x - mean) * rstd * (1 + gamma) + beta
                const int chl = 8 * cb + 2 * q + AH;                     // channel within the block's 32
                const int ch = (e_m0 >> 1) + chl;
                const bool cok = oxb < OWv;
                const unsigned ob = (unsigned)oyb * rstride_b + (unsigned)oxb * 4u + (unsigned)ch * cstride_b;
                const float mu = ep_scale[chl], rsd = ep_bias[chl];
                const float nmr = -mu * rsd;
#pragma unroll
                for (int rr = 0; rr < 4; rr++) {
                    const bool ok = full || (cok && oyb + rr < OHv);
                    const f32x4v xv = ok ? *(const f32x4v*)((const char*)spx_n + ob + (unsigned)rr * rstride_b) : zero4;
                    f32x4v w;
#pragma unroll
                    for (int e = 0; e < 4; e++) w[e] = fmaf(fmaf(xv[e], rsd, nmr), gam[rr][e] + 1.f, v[rr][e]);
                    w = act4g(w);
                    if (ok) *(f32x4v*)((char*)y_n + ob + (unsigned)rr * rstride_b) = w;
                }
            }
        }
        };
        if (WINO4B_EXP & 2048) __builtin_amdgcn_s_setprio(2);
        if (ah) finish(std::integral_constant<int, 1>{}); else finish(std::integral_constant<int, 0>{});
        if (WINO4B_EXP & 2048) __builtin_amdgcn_s_setprio(0);
        B4_STAMP(50);
#if WINO4B_EXP & 32768
        if (tiles_done == 2 && w4_fresh_lane() == 0) {
            unsigned long long* out = (unsigned long long*)p.y + ((size_t)blockIdx.x * 8 + wave) * 64;
            for (int i = 0; i < 51; i++) out[i] = stamps_lds[wave * 64 + i];
            out[62] = (unsigned long long)__builtin_amdgcn_s_getreg((3 << 11) | (0 << 6) | 20);      // HW_REG_XCC_ID[3:0]
            out[63] = (unsigned long long)__builtin_amdgcn_s_getreg((31 << 11) | (0 << 6) | 4);      // HW_REG_HW_ID
        }
        tiles_done++;
#endif
        if (!has_next) break;
        tile = next;
        par ^= 1;
    }
}

template <int MODE, int TAIL>
int launch_wino4b_mode(const ConvParams& p0, hipStream_t s) {
    ConvParams p = p0;
    p.tilesX = (p.OW + 31) / 32;
    p.tilesY = (p.OH + 7) / 8;
    p.mblocks = p.CoutP / 64;
    const int64_t tiles = (int64_t)p.N * p.tilesX * p.tilesY * p.mblocks;
    if (tiles > 0x7fffffffLL) return PG_ERR_TOO_LARGE;
    p.total_tiles = (int)tiles;
    const int cin_loop = ((p.Cin + B4_KC - 1) / B4_KC) * B4_KC;
    const size_t lds = ((size_t)B4_RAW + B4_V + 2 * cin_loop + 256 + B4_NI * B4_NL * 64) * sizeof(float) + ((WINO4B_EXP & 32768) ? 4096 : 0);
    if ((int64_t)36 * cin_loop * p.CoutP * 4 > 0x7fffffffLL) return PG_ERR_TOO_LARGE;      // the U stream uses 32-bit byte offsets
    {   // the tail addresses one image of y / residual / spade_x with 32-bit byte offsets
        const int64_t ext = 1 + (int64_t)(p.f.spade_x ? p.Cout / 2 - 1 : p.Cout - 1) * p.ys[1] + (int64_t)(p.OH - 1) * p.ys[2] + (int64_t)(p.OW - 1) * p.ys[3];
        if (ext * 4 > 0xffffffffLL || (int64_t)p.OH * p.OW * 4 > 0xffffffffLL) return PG_ERR_TOO_LARGE;
    }
    if (p.pad_x != 1 && p.pad_x != 3) return PG_ERR_UNSUPPORTED;    // W % 4 == 0 and OW % 4 == 0 together leave odd paddings only; 0 / 4 had their own (never exercised) staging arithmetic: removed in round 5
    if (lds > 80 * 1024) return PG_ERR_UNSUPPORTED;                                          // two workgroups per CU
    if (p.ys[3] != 1 || ((p.ys[0] | p.ys[1] | p.ys[2] | p.f.noise_batch_stride) & 3) != 0 || (p.OW & 3) != 0 ||
        ((((uintptr_t)p.y) | ((uintptr_t)p.f.noise) | ((uintptr_t)p.f.residual) | ((uintptr_t)p.f.spade_x)) & 15) != 0) return PG_ERR_UNSUPPORTED;
    static const int per_cu = [] { const char* e = getenv("PG_WINO4B_PER_CU"); return (e && atoi(e) == 1) ? 1 : 2; }();      // dev A/B: 1 = one workgroup per CU (LDS request padded past 80 KB)
    const int64_t slots = (int64_t)num_cu() * per_cu;
    const int64_t blocks = tiles < slots ? tiles : slots;
    const size_t lds_req = per_cu == 1 ? (size_t)100 * 1024 : lds;
    static PerDeviceOnce lds_attr;
    const hipError_t e = lds_attr.run([] { return hipFuncSetAttribute((const void*)conv2d_wino4b<MODE, TAIL>, hipFuncAttributeMaxDynamicSharedMemorySize, 100 * 1024); });
    if (e != hipSuccess) return (int)e;
    hipLaunchKernelGGL((conv2d_wino4b<MODE, TAIL>), dim3((unsigned)blocks), dim3(512), lds_req, s, p);
    return launch_status();
}

}  // namespace pgconv
