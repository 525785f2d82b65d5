// 16-bit (bf16 / fp16) channels-last convolution for gfx950 as an implicit GEMM on v_mfma_f32_32x32x16_{bf16,f16}
// with fp32 accumulation and the StyleGAN2 elementwise tail (demodulation scale, noise, bias, activation, gain, clamp,
// residual) folded into its epilogue.
//
// What it replaces in the reference: the cuDNN call behind conv2d_gradfix.conv2d / conv_transpose2d
// (torch_utils/ops/conv2d_gradfix.py:35-43) for half-precision tensors -- the discriminator's fp16 blocks
// (training/networks.py:444-523, `use_fp16`, conv_clamp 256) and the half-precision synthesis blocks of the StyleGAN2
// stack (networks.py:2147-2194 with use_fp16; BASELINE config 5 runs them in bf16 at 1024^2) -- plus the
// `fma` / `bias_act` that follow it (networks.py:73-94, 170-179).
//
// Layout: activations are NHWC ("channels_last", the layout the reference's --nhwc option selects for fp16): the
// K = 8 consecutive input channels one MFMA lane consumes are one aligned 16-byte word in HBM and in LDS, for every
// tap shift.  GEMM view:
//     M = Cout   A = weights, packed [Cin/16][tap][k-half][Cout][8]: a lane's fragment is one 16-byte LDS read
//     N = pixels B = activations, halo tile [rows][cols][KC channels] in LDS, 16 bytes per (pixel, 8 channels);
//                the 16-byte slot index is XOR-swizzled with the pixel index so that the ds_read_b128 of 32
//                consecutive pixels is bank-conflict free; LDS-DMA writes lane-linear, so the swizzle sits in the
//                per-lane SOURCE address (cdna_hip_programming.md rule 21)
//     D          col = lane & 31 = pixel, rows = couts 8g + 4*(lane>>5) + j: after the conversion to 16 bit one
//                v_permlane32_swap per register pair leaves 8 consecutive couts of one pixel in a lane ->
//                16-byte stores into the NHWC output (guide T21).
// Workgroup = 512 threads = 8 waves (2 per SIMD, 256 VGPRs each), one workgroup per CU; output tile = BM couts x
// (TH rows x TW cols); all waves share the weight slab of a K chunk, so a weight byte is read from L2 once per 512
// pixels.  One continuous stream of K chunks runs through THREE LDS staging buffers with TWO chunks in flight
// (global -> LDS by 16-byte LDS-DMA issued from inline asm, counted s_waitcnt vmcnt, one s_barrier per chunk); the
// stream runs across tile boundaries, so a tile's epilogue stores and the next tile's first loads overlap MFMAs.
// Per-cout epilogue constants and the tile's noise samples travel by LDS-DMA too: the K loop and the epilogue
// contain no compiler-visible global load (one would make hipcc drain the whole DMA queue with vmcnt(0)).
//
// Roofline: near the ridge.  Algorithmic FLOPs 2*N*Cout*OH*OW*Cin*KH*KW against the 2.5 PFLOP/s dense 16-bit matrix
// peak; algorithmic bytes 2*(numel(x) + numel(y)) + weights against 8 TB/s HBM: 3x3 C=64 has 288 FLOP/B (ridge 312),
// C=32 is HBM-bound, C>=128 MFMA-bound.

#pragma once
#include <algorithm>
#include <cstdlib>
#include <type_traits>
#include "pg_common.h"

namespace pgconv16 {

using namespace pg;

typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef float f32x2 __attribute__((ext_vector_type(2)));
typedef int i32x4 __attribute__((ext_vector_type(4)));
typedef unsigned u32x2 __attribute__((ext_vector_type(2)));
typedef unsigned u32x4 __attribute__((ext_vector_type(4)));
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef __bf16 bf16x2 __attribute__((ext_vector_type(2)));
typedef _Float16 f16x8 __attribute__((ext_vector_type(8)));
typedef _Float16 f16x2 __attribute__((ext_vector_type(2)));

// dtype traits: the MFMA, fp32 -> packed pair, 16-bit -> fp32
template <typename T> struct Half16;
template <> struct Half16<bf16_t> {
    static __device__ __forceinline__ f32x16 mma(i32x4 a, i32x4 b, f32x16 c) {
        return __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf16x8, a), __builtin_bit_cast(bf16x8, b), c, 0, 0, 0);
    }
    static __device__ __forceinline__ unsigned pack(float lo, float hi) {
        const f32x2 v = {lo, hi};
        return __builtin_bit_cast(unsigned, __builtin_convertvector(v, bf16x2));
    }
    static __device__ __forceinline__ float widen(unsigned short u) { return __builtin_bit_cast(float, (unsigned)u << 16); }
};
template <> struct Half16<f16_t> {
    static __device__ __forceinline__ f32x16 mma(i32x4 a, i32x4 b, f32x16 c) {
        return __builtin_amdgcn_mfma_f32_32x32x16_f16(__builtin_bit_cast(f16x8, a), __builtin_bit_cast(f16x8, b), c, 0, 0, 0);
    }
    static __device__ __forceinline__ unsigned pack(float lo, float hi) {
        const f32x2 v = {lo, hi};
        return __builtin_bit_cast(unsigned, __builtin_convertvector(v, f16x2));
    }
    static __device__ __forceinline__ float widen(unsigned short u) { return (float)__builtin_bit_cast(f16_t, u); }
};

enum { OUT_VEC16 = 0, OUT_SCALAR16 = 1, OUT_SCALAR32 = 2, OUT_VEC32 = 3 };   // VEC32: float32 channels-last rows, 4 couts per 16-byte store (split-K workspace)

struct Conv16Params {
    const void* x;          // [N, H, W, Cin] 16-bit, dense
    const void* wp;         // packed weights, sample n at wp + n * w_nstride (elements); 0 = shared
    void* y;
    int64_t w_nstride;
    int64_t w_bytes;        // extent of the packed weight buffer (range check of the DMA descriptor)
    int64_t y_bytes;        // extent of y in bytes (range check of the vector epilogue's store descriptor; < 2^31 in that mode)
    int N, Cin, H, W, Cout, CoutP, OH, OW;
    int xC;                 // channels per pixel of x (>= Cin; the pixel stride)
    int pad_y, pad_x;
    int ksplit, kpart;      // split-K: `ksplit` workgroups share one output tile, each reducing `kpart` channels into
    int64_t ws_slice;       // its own slice (ws_slice elements apart) of the float32 workspace that y then points to
    int64_t ys[4];          // (n, c, y, x) strides of the output in ELEMENTS of its dtype
    int osy, osx, ooy, oox; // output pixel (oy, ox) is written at (oy*osy + ooy, ox*osx + oox)
    int out_mode;
    unsigned long long* stamps;   // dev: s_memtime stamps of workgroup 0 / wave 0 (PG_CONV16_DBG & 32; pg_conv2d16_debug_stamps)
    int dbg;                // dev ablations (PG_CONV16_DBG; results wrong by design): 1 no output stores, 2 no halo DMA, 4 no MFMA, 8 no epilogue maths
    int tilesX, tilesY, mblocks, total_tiles;
    unsigned m_tilesX, m_tilesY, m_mblocks, m_ksplit;      // ceil(2^32 / d) of the tile-decomposition divisors (0 for d == 1)
    pg_conv2d16_fusion f;
};

constexpr int THREADS = 512, WAVES = 8;

__device__ __forceinline__ unsigned lds_offset(const void* p) {
    return (unsigned)(uintptr_t)(const __attribute__((address_space(3))) unsigned char*)p;
}
// LDS-DMA (no staging registers); LDS destination = M0 + lane * size.  M0 is written in the statement that uses it and
// restored (compiler-reserved, cdna_hip_programming.md 5.7).  The range check of the buffer descriptor returns zeros for
// offsets beyond num_records: zero padding, channel padding and "this lane has nothing to load" are all the sentinel offset.
__device__ __forceinline__ void dma16(i32x4 rsrc, unsigned lds_byte, unsigned voff_bytes, unsigned soff_bytes) {
    unsigned keep;
    asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %1\n\ts_nop 0\n\tbuffer_load_dwordx4 %2, %3, %4 offen lds\n\ts_mov_b32 m0, %0"
                 : "=&s"(keep) : "s"(lds_byte), "v"(voff_bytes), "s"(rsrc), "s"(soff_bytes) : "memory");
}
__device__ __forceinline__ void dma4(i32x4 rsrc, unsigned lds_byte, unsigned voff_bytes, unsigned soff_bytes) {
    unsigned keep;
    asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %1\n\ts_nop 0\n\tbuffer_load_dword %2, %3, %4 offen lds\n\ts_mov_b32 m0, %0"
                 : "=&s"(keep) : "s"(lds_byte), "v"(voff_bytes), "s"(rsrc), "s"(soff_bytes) : "memory");
}
template <int N> __device__ __forceinline__ void vm_wait() { asm volatile("s_waitcnt vmcnt(%0)" :: "i"(N) : "memory"); }

#ifndef PG_CONV16_STAMPS
#define PG_CONV16_STAMPS 0       // 1 (tools/conv16_stamps.py / conv16_ablate.py build their own plugin with it): the PG_CONV16_DBG switches -- ablations, s_memtime
                                 // stamps of one wave; in the product build they are compiled out (the stamp checks alone were 2 % of config 5)
#endif
constexpr unsigned SENTINEL = 0x80000000u;

__device__ __forceinline__ i32x4 make_rsrc(const void* base, int64_t bytes) {
    const uint64_t b = (uint64_t)(uintptr_t)base;
    i32x4 r;
    r[0] = __builtin_amdgcn_readfirstlane((int)(unsigned)b);
    r[1] = __builtin_amdgcn_readfirstlane((int)(unsigned)(b >> 32) & 0xffff);
    r[2] = __builtin_amdgcn_readfirstlane((int)(bytes > 0x7fffffffLL ? 0x7fffffffLL : (bytes < 0 ? 0 : bytes)));
    r[3] = 0x00020000;
    return r;
}

template <int KH, int KW, int S, int TWL, int WM, int MT, int NT, int KC, int NB, bool SPLIT_G = false>
struct Geo16 {
    static constexpr int T = KH * KW;
    static constexpr int WP = WAVES / WM;                 // waves along pixels
    static constexpr int RP = 32 / TWL;                   // rows of one 32-pixel N tile
    static constexpr int TH = WP * NT * RP, TW = TWL;     // output tile of one workgroup
    static constexpr int BM = WM * MT * 32;
    static constexpr int IH_T = (TH - 1) * S + KH, IW_T = (TW - 1) * S + KW;
    static constexpr int SLOTS = KC / 8;                  // 16-byte slots per pixel per chunk
    static constexpr int PER = 16 / SLOTS;                // pixels per 256-byte LDS bank row
    static constexpr int KS = KC / 16;                    // MFMA k-steps per tap per chunk
    static constexpr int NXS = IH_T * IW_T * SLOTS;       // halo slots
    static constexpr int NXS_PAD = (NXS + 255) / 256 * 256;  // the halo / weight boundary is aligned to FOUR wave-instructions: in the two-role form
                                                          // instruction i2 of every loader wave is then on the same side of it (a compile-time fact per i2)
    static constexpr int NWS = KS * T * 2 * BM;           // weight slots
    static constexpr int DPC = (NXS_PAD + NWS + THREADS - 1) / THREADS;      // DMA instructions per thread per chunk
    // one staging buffer, in slots.  One-role form: a whole number of 512-thread request instructions.  Two-role form: when halo + weights
    // are a multiple of 256 slots (one request instruction of the FOUR loader waves) the buffer ends there and the request instructions past
    // it do not exist -- what lets two 32-channel buffers of the 16 x 32-pixel x 64-cout tile fit into 160 KB
    static constexpr bool TIGHT = SPLIT_G && (NXS_PAD + NWS) % 256 == 0;
    static constexpr int LDS_BUF = TIGHT ? NXS_PAD + NWS : DPC * THREADS;
    static constexpr int NBUF = NB;                       // staging buffers; NBUF - 1 chunks in flight
    static constexpr int EPS = (BM + 63) / 64 * 64;       // per-cout constants: [scale EPS][bias EPS] floats
    static constexpr int EP_FLOATS = 2 * EPS + THREADS;   // + the tile's noise samples; one extra 64-float pad for idle waves
    static constexpr size_t LDS_BYTES = (size_t)NBUF * LDS_BUF * 16 + (size_t)2 * EP_FLOATS * 4 + 256;
    static_assert(SLOTS == 2 || SLOTS == 4 || SLOTS == 8, "KC must be 16, 32 or 64");
    static_assert(TH * TW <= THREADS && 2 * EPS <= THREADS, "tile / cout block too large for the per-tile side loads");
    static_assert(LDS_BYTES <= 160 * 1024, "LDS budget");
};

__device__ __forceinline__ float act_slope(int act, float alpha) { return act == PG_ACT_LINEAR ? 1.f : (act == PG_ACT_RELU ? 0.f : alpha); }

// q = L / d for the tile decomposition, on the scalar unit: M = ceil(2^32 / d) from the host (0 for d == 1); exact while
// L * d < 2^32 (launch16 checks it).
__device__ __forceinline__ unsigned div_magic(unsigned L, unsigned M) { return M ? __umulhi(L, M) : L; }

// SPLIT (round 3): four more waves (8 .. 11) own everything that is not multiplying -- the chunk requests, the tile preparation, the
// per-tile side loads -- and run two chunks ahead of waves 0 .. 7, which only wait at the chunk barrier, multiply and write tiles out.
// In the one-role form every wave spends ~1.1-1.4 k cycles per chunk issuing its five requests (and ~3.5 k per tile preparing it)
// between the barrier and its first MFMA, all eight at the same time: the matrix pipe has nothing queued for a third of the K loop
// (tools/conv16_stamps.py).  Needs >= 2 chunks per tile (the side buffers of tile T + 2 are requested with its first chunk, which must
// not happen before tile T's epilogue has read its own).
constexpr int LOADERS = 4;
template <typename T, int KH, int KW, int S, int TWL, int WM, int MT, int NT, int KC, int NB, bool SPLIT = false>
__global__ __launch_bounds__(SPLIT ? THREADS + 64 * LOADERS : THREADS, SPLIT ? 3 : 2) void conv2d_mfma16(Conv16Params p) {
    const int dbg_ = PG_CONV16_STAMPS ? p.dbg : 0;      // the dev switches exist in the diagnostic build only (PG_CONV16_STAMPS)
    typedef Geo16<KH, KW, S, TWL, WM, MT, NT, KC, NB, SPLIT> G;
    typedef Half16<T> HT;
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    typedef const __attribute__((address_space(3))) i32x4* lds_v4;
    typedef const __attribute__((address_space(3))) f32x4* lds_f4;
    typedef const __attribute__((address_space(3))) float* lds_f;

    const int t = threadIdx.x, lane = t & 63;
    const int wave = __builtin_amdgcn_readfirstlane(t >> 6);
    const unsigned smem_b = __builtin_amdgcn_readfirstlane(lds_offset(smem));
    const unsigned side_b = smem_b + (unsigned)(G::NBUF * G::LDS_BUF) * 16u;       // [2][EP_FLOATS] floats, then a 256-byte dump
    const unsigned dump_b = side_b + 2u * G::EP_FLOATS * 4u;
    const int half = lane >> 5, l31 = lane & 31;
    const int wpx = wave % G::WP, wmx = wave / G::WP;
    const bool loader = SPLIT && wave >= WAVES;           // wave-uniform role
    const int lw = wave - WAVES;                          // loader index 0 .. 3
    // request instruction i2 of a chunk, as issued by this thread: the one-role form's instruction i of wave vw (same slots, same maps)
    constexpr int LD = SPLIT ? 2 * G::DPC : G::DPC;
    auto slot_of = [&](int i2) __attribute__((always_inline)) { return SPLIT ? ((i2 >> 1) * WAVES + (i2 & 1) * LOADERS + lw) * 64 : (i2 * WAVES + wave) * 64; };
    // is request i2 a weight-slab request?  wave-uniform; in the two-role form a constant per i2 (NXS_PAD is a multiple of 256 slots)
    // does request i2 exist?  (two-role form with a tight buffer: the instructions whose slots lie past the buffer do not)
    auto exists = [&](int i2) __attribute__((always_inline)) { return !SPLIT || ((i2 >> 1) * WAVES + (i2 & 1) * LOADERS) * 64 < G::LDS_BUF; };
    constexpr int LD_EFF = SPLIT ? (G::LDS_BUF + 255) / 256 : G::DPC;      // requests per issuing thread and chunk that exist
    auto is_weight = [&](int i2) __attribute__((always_inline)) { return SPLIT ? ((i2 >> 1) * WAVES + (i2 & 1) * LOADERS) * 64 >= G::NXS_PAD : slot_of(i2) >= G::NXS_PAD; };
    const int total = p.total_tiles;
    const int q8 = total >> 3, r8 = total & 7;
    const int cin_loop = p.ksplit > 1 ? p.kpart : p.Cin;  // channels one workgroup reduces per tile
    const int nchunks = (cin_loop + KC - 1) / KC;
    const int tail_ch = cin_loop % KC;                    // channels of a partial last chunk (0 = none)

    // ---- descriptors: all of them cover a whole tensor and never change; the image / sample offset travels in the
    // instruction's scalar offset.  (x larger than 2 GB: the host falls back to one launch per image.)
    const i32x4 xrsrc = make_rsrc(p.x, (int64_t)p.N * p.H * p.W * p.xC * 2);
    const i32x4 wrsrc = make_rsrc(p.wp, p.w_bytes);
    // the vector epilogues store through a range-checked descriptor: every wave issues exactly EP_STORES store instructions
    // per tile (masked-off lanes carry the sentinel offset), which the counted waits below rely on
    const __amdgpu_buffer_rsrc_t yrsrc = __builtin_amdgcn_make_buffer_rsrc(p.y, 0, (int)(p.y_bytes > 0x7fffffffLL ? 0x7fffffffLL : p.y_bytes), 0x00020000);

    // ---- per-thread DMA maps, fixed for the whole kernel.  Instruction i of a chunk moves 16-byte slot
    // s = (i * 8 + wave) * 64 + lane of the staging buffer: slots below NXS_PAD are the halo tile, the rest the weight slab.
    //   rel[i]  byte offset of the slot's source relative to the tile's first halo pixel (halo) / the slab's first row (weights);
    //           SENTINEL for padding slots
    //   hyx[i]  halo coordinates (row | col << 16) for the border test; rows >= 0x4000 never pass it
    unsigned rel[LD], hyx[LD];
    unsigned tailmask = 0;                                // bit i: this lane's halo slot i holds channels >= tail_ch of a chunk
    auto setup_maps = [&]() __attribute__((always_inline)) {
#pragma unroll
    for (int i = 0; i < LD; i++) {
        const int sb = slot_of(i);
        rel[i] = SENTINEL; hyx[i] = 0x4000u;
        if (!exists(i)) continue;
        if (is_weight(i)) {
            const int e = sb + lane - G::NXS_PAD;
            const int row = e / G::BM, col = e % G::BM;
            if (e < G::NWS) rel[i] = (unsigned)(row * p.CoutP + col) * 16u;
        } else {
            const int s = sb + lane;
            const int q = s / G::SLOTS;
            const int c = (s % G::SLOTS) ^ ((q / G::PER) & (G::SLOTS - 1));
            const int hy = q / G::IW_T, hx = q % G::IW_T;
            if (s < G::NXS) {
                rel[i] = (unsigned)((hy * p.W + hx) * p.xC + c * 8) * 2u;
                hyx[i] = (unsigned)hy | ((unsigned)hx << 16);
            }
            if (tail_ch && c * 8 >= tail_ch) tailmask |= 1u << i;
        }
    }

    };
    if constexpr (!SPLIT) setup_maps();               // (two-role form: by the loader waves, after the roles part -- the maps must not be live in the multiplying waves' code)

    // ---- tile descriptors: [parity] of the tile being multiplied / the next one
    int d_z0 = 0, d_z1 = 0, d_n0 = 0, d_n1 = 0, d_oy00 = 0, d_oy01 = 0, d_ox00 = 0, d_ox01 = 0, d_m00 = 0, d_m01 = 0;      // (explicit pairs: a runtime-indexed array would live in scratch)
    unsigned voff[LD];                                    // DMA offsets of the tile being requested (halo lanes; weight lanes = rel)
    unsigned w_soff = 0, x_soff0 = 0;

    // Tile -> (n, tile_y, tile_x, m-block[, K share]), XCD-aware: workgroups that share an XCD (id % 8) walk one contiguous
    // range of logical tiles, so adjacent halos and the weight slabs hit the same L2.  All of it runs on the scalar unit.
    auto prep_tile = [&](int tile, int par, const bool want_voff = true) __attribute__((always_inline)) {
        const int xcd = tile & 7;
        unsigned L = (unsigned)((xcd < r8 ? xcd * (q8 + 1) : r8 * (q8 + 1) + (xcd - r8) * q8) + (tile >> 3));
        int zsl = 0;
        if (p.ksplit > 1) { const unsigned q = div_magic(L, p.m_ksplit); zsl = (int)(L - q * p.ksplit); L = q; }      // the shares of one tile run side by side
        const int cbeg = zsl * p.kpart;
        unsigned q = div_magic(L, p.m_mblocks); const int mb = (int)(L - q * p.mblocks); L = q;
        q = div_magic(L, p.m_tilesX); const int tx = (int)(L - q * p.tilesX); L = q;
        q = div_magic(L, p.m_tilesY); const int ty = (int)(L - q * p.tilesY);
        const int n = (int)q;
        const int oy0 = ty * G::TH, ox0 = tx * G::TW, m0 = mb * G::BM;
        // value selects, not control flow: stores through a selected address would put the descriptors into scratch
        d_z1 = par ? zsl : d_z1;   d_z0 = par ? d_z0 : zsl;
        d_n1 = par ? n : d_n1;     d_n0 = par ? d_n0 : n;
        d_oy01 = par ? oy0 : d_oy01; d_oy00 = par ? d_oy00 : oy0;
        d_ox01 = par ? ox0 : d_ox01; d_ox00 = par ? d_ox00 : ox0;
        d_m01 = par ? m0 : d_m01;  d_m00 = par ? d_m00 : m0;
        if (!want_voff) return;                           // (SPLIT: the multiplying waves only need the coordinates, for their epilogue)
        // halo: source offset = tile origin + rel; lanes outside the image (border tiles only) get the sentinel
        const int ty0 = oy0 * S - p.pad_y, tx0 = ox0 * S - p.pad_x;
        const unsigned org = (unsigned)((ty0 * p.W + tx0) * p.xC * 2);
        const bool interior = ty0 >= 0 && tx0 >= 0 && ty0 + G::IH_T <= p.H && tx0 + G::IW_T <= p.W;      // wave-uniform
#pragma unroll
        for (int i = 0; i < LD; i++) {
            if (exists(i) && !is_weight(i)) {             // wave-uniform
                if (interior) {
                    voff[i] = org + rel[i];               // padding slots: org + 2^31 stays out of range (0 <= org < 2^31)
                } else {
                    const unsigned gy = (unsigned)(ty0 + (int)(hyx[i] & 0xffffu)), gx = (unsigned)(tx0 + (int)(hyx[i] >> 16));
                    voff[i] = (gy < (unsigned)p.H && gx < (unsigned)p.W) ? org + rel[i] : SENTINEL;
                }
            }
        }
        x_soff0 = (unsigned)(((int64_t)n * p.H * p.W * p.xC + cbeg) * 2);
        w_soff = (unsigned)(((int64_t)n * p.w_nstride + (int64_t)m0 * 8) * 2) + (unsigned)((cbeg / 16) * G::T * 2) * (unsigned)p.CoutP * 16u;
    };

    // Per-cout epilogue constants and the tile's noise samples: wave w < EPS/64 fetches 64 demodulation scales, the next
    // EPS/64 waves 64 biases (idle waves write zeros into the dump area); every thread one noise sample (tile pixel t).
    // Whole-tensor descriptors (constant), the row / plane offset in the scalar offset; the per-thread parts are fixed.
    // Four-phase mode (pg_conv2d16_fusion::phase_cout): m-block -> (phase, first channel inside the phase); the per-cout vectors and
    // the stores use the phase-local channel, the stores and the noise the phase's pixel offset.  A block never straddles phases
    // (the host checks phase_cout % BM == 0).
    const int pc = p.f.phase_cout, ce = pc ? pc : p.Cout;
    auto phase_of = [&](int m0) __attribute__((always_inline)) { return pc ? (int)(m0 >= pc) + (int)(m0 >= 2 * pc) + (int)(m0 >= 3 * pc) : 0; };
    static_assert(!SPLIT || 2 * (G::EPS / 64) <= LOADERS, "per-cout side vectors: one 64-float DMA per loader wave");
    const int sw = SPLIT ? lw : wave;                     // the wave index the side loads are dealt out by
    const bool side_scale = sw < G::EPS / 64, side_bias = !side_scale && sw < 2 * (G::EPS / 64);
    const i32x4 sbrsrc = side_scale ? make_rsrc(p.f.out_scale, p.f.out_scale ? (int64_t)p.N * ce * 4 : 0)
                                    : make_rsrc(p.f.bias, (p.f.bias && side_bias) ? (int64_t)ce * 4 : 0);
    const i32x4 nrsrc = make_rsrc(p.f.noise, p.f.noise ? ((int64_t)(p.N - 1) * p.f.noise_batch_stride + (pc ? 3 * p.f.noise_phase_stride : 0) + (int64_t)p.OH * p.OW) * 4 : 0);
    const unsigned side_rel = (unsigned)((sw * 64 + lane) % G::EPS) * 4u;                   // + m0 * 4 (+ n * Cout * 4 for the scales)
    auto issue_side = [&](int par) __attribute__((always_inline)) {
        const int n = par ? d_n1 : d_n0, oy0 = par ? d_oy01 : d_oy00, ox0 = par ? d_ox01 : d_ox00, m0 = par ? d_m01 : d_m00;
        const unsigned sb_ = side_b + (unsigned)(par * G::EP_FLOATS) * 4u;
        const int ph = phase_of(m0), mc0 = m0 - ph * pc;
        const bool live = (side_scale || side_bias) && mc0 + (int)(side_rel >> 2) < ce;
        dma4(sbrsrc, (side_scale || side_bias) ? sb_ + (unsigned)(sw * 64) * 4u : dump_b, live ? side_rel : SENTINEL,
             (unsigned)(mc0 + (side_scale ? n * ce : 0)) * 4u);
        // the tile's noise samples, one per pixel: every thread one (one-role form), every loader thread two
#pragma unroll
        for (int j = 0; j < (SPLIT ? WAVES / LOADERS : 1); j++) {
            const int vw = SPLIT ? j * LOADERS + lw : wave, tt = vw * 64 + lane;
            const int ny = oy0 + tt / G::TW, nx = ox0 + tt % G::TW;
            const unsigned noise_voff = (tt < G::TH * G::TW && ny < p.OH && nx < p.OW) ? (unsigned)(ny * p.OW + nx) * 4u : SENTINEL;
            dma4(nrsrc, sb_ + (unsigned)(2 * G::EPS + vw * 64) * 4u, noise_voff, (unsigned)(n * p.f.noise_batch_stride + ph * p.f.noise_phase_stride) * 4u);
        }
    };

    // ---- the chunk stream
    int c_tile = blockIdx.x, c_chunk = 0, c_ahead = 0;    // request cursor: tile, chunk, tiles ahead of the one being multiplied
    bool c_done = false;
    int ibuf = 0, cbuf = 0;                               // staging buffer of the next request / of the next chunk to multiply
    int inflight = 0;                                     // chunks requested and not yet multiplied
    int dpar = 0;                                         // descriptor parity of the tile being multiplied

    auto issue_next = [&]() __attribute__((always_inline)) {
        if (c_done || (!SPLIT && c_chunk == 0 && c_ahead > 1)) return;
        if (c_chunk == 0) {
            const int par = SPLIT ? (c_ahead & 1) : dpar ^ (c_ahead & 1);      // (SPLIT: c_ahead counts the tiles requested so far)
            prep_tile(c_tile, par);
            if (!(dbg_ & 16)) issue_side(par);
        }
        const int c0 = c_chunk * KC;
        const bool partial = tail_ch != 0 && c_chunk == nchunks - 1;
        const unsigned x_soff = x_soff0 + (unsigned)c0 * 2u;
        const unsigned wk_soff = w_soff + (unsigned)((c0 / 16) * G::T * 2) * (unsigned)p.CoutP * 16u;
        const unsigned buf_b = smem_b + (unsigned)(ibuf * G::LDS_BUF) * 16u;
#pragma unroll
        for (int i = 0; i < LD; i++) {
            if (!exists(i)) continue;
            const int sb = slot_of(i);
            const bool is_w = is_weight(i);               // wave-uniform
            unsigned vo = is_w ? rel[i] : voff[i];
            if (partial && ((tailmask >> i) & 1)) vo = SENTINEL;
            if (!is_w && (dbg_ & 2)) vo = SENTINEL;
            dma16(is_w ? wrsrc : xrsrc, buf_b + (unsigned)sb * 16u, vo, is_w ? wk_soff : x_soff);
        }
        ibuf = ibuf == G::NBUF - 1 ? 0 : ibuf + 1;
        inflight++;
        if (++c_chunk == nchunks) {
            c_chunk = 0;
            c_tile += gridDim.x;
            c_ahead++;
            if (c_tile >= total) c_done = true;
        }
    };

    // ---- operand addresses: constant for the whole kernel
    int qrow[NT];                                          // halo pixel index of this lane's output pixel, tap (0, 0)
#pragma unroll
    for (int nt = 0; nt < NT; nt++) {
        const int row_l = (wpx * NT + nt) * G::RP + l31 / TWL, col_l = l31 % TWL;
        qrow[nt] = row_l * S * G::IW_T + col_l * S;
    }
    const int a_lane = G::NXS_PAD + half * G::BM + (wmx * MT) * 32 + l31;

    f32x16 acc[MT][NT];

    // One chunk out of staging buffer `buf`.  With unit stride and one row per N tile the wave's NT output rows share
    // KH + NT - 1 halo rows: each B fragment is read once and feeds every (row, ky) pair it belongs to, and the KH weight
    // fragments of a kernel column are held across those rows -- (KH*MT + KH+NT-1) LDS reads per KH*MT*NT MFMAs.
    auto compute_chunk = [&](int buf) __attribute__((always_inline)) {
        const unsigned char* base = smem + (size_t)buf * G::LDS_BUF * 16;
        auto b_frag = [&](int q, int ks) __attribute__((always_inline)) {
            const int slot = q * G::SLOTS + ((ks * 2 + half) ^ ((q / G::PER) & (G::SLOTS - 1)));
            return *(lds_v4)(base + (size_t)slot * 16);
        };
        auto a_frag = [&](int ks, int tap, int mt) __attribute__((always_inline)) {
            return *(lds_v4)(base + (size_t)(a_lane + ((ks * G::T + tap) * 2) * G::BM + mt * 32) * 16);
        };
#pragma unroll
        for (int ks = 0; ks < G::KS; ks++) {
            if constexpr (S == 1 && G::RP == 1) {
#pragma unroll
                for (int kx = 0; kx < KW; kx++) {
                    i32x4 a[KH][MT];
#pragma unroll
                    for (int ky = 0; ky < KH; ky++)
#pragma unroll
                        for (int mt = 0; mt < MT; mt++) a[ky][mt] = a_frag(ks, ky * KW + kx, mt);
#pragma unroll
                    for (int hr = 0; hr < KH + NT - 1; hr++) {
                        const i32x4 b = b_frag(qrow[0] + hr * G::IW_T + kx, ks);
#pragma unroll
                        for (int nt = 0; nt < NT; nt++) {
                            const int ky = hr - nt;
                            if (ky < 0 || ky >= KH) continue;
#pragma unroll
                            for (int mt = 0; mt < MT; mt++) acc[mt][nt] = HT::mma(a[ky][mt], b, acc[mt][nt]);
                        }
                    }
                }
            } else {
#pragma unroll
                for (int ky = 0; ky < KH; ky++) {
#pragma unroll
                    for (int kx = 0; kx < KW; kx++) {
                        i32x4 a[MT], b[NT];
#pragma unroll
                        for (int mt = 0; mt < MT; mt++) a[mt] = a_frag(ks, ky * KW + kx, mt);
#pragma unroll
                        for (int nt = 0; nt < NT; nt++) b[nt] = b_frag(qrow[nt] + ky * G::IW_T + kx, ks);
#pragma unroll
                        for (int mt = 0; mt < MT; mt++)
#pragma unroll
                            for (int nt = 0; nt < NT; nt++) acc[mt][nt] = HT::mma(a[mt], b[nt], acc[mt][nt]);
                    }
                }
            }
        }
    };

    const float gain = p.f.gain;
    const float cl = p.f.clamp >= 0.f ? p.f.clamp : __builtin_inff();
    const float slope = act_slope(p.f.act, p.f.alpha);
    const bool has_scale = p.f.out_scale != nullptr;
    const float noise_gain = p.f.noise ? p.f.noise_gain : 0.f;
    constexpr int EP_STORES = MT * NT * 2;                // 16-byte stores of the vector epilogue per wave

    auto write_tile = [&]() __attribute__((always_inline)) {
        // ---- epilogue: D col = lane & 31 (pixel), row = (reg & 3) + 8 * (reg >> 2) + 4 * (lane >> 5) (cout)
        const int e_z = dpar ? d_z1 : d_z0;
        const int e_n = dpar ? d_n1 : d_n0, e_oy0 = dpar ? d_oy01 : d_oy00, e_ox0 = dpar ? d_ox01 : d_ox00, e_mt0 = dpar ? d_m01 : d_m00;
        const int e_ph = phase_of(e_mt0), e_m0 = e_mt0 - e_ph * pc;           // phase-local first channel of this block
        const int e_ooy = pc ? (e_ph >> 1) : p.ooy, e_oox = pc ? (e_ph & 1) : p.oox;
        const unsigned char* side = smem + (size_t)G::NBUF * G::LDS_BUF * 16 + (size_t)dpar * G::EP_FLOATS * 4;
        // per-cout constants of this lane's rows, gain folded in: v = clamp(act(acc * scale + noise + bias) * gain) with a
        // positively homogeneous activation (linear / relu / lrelu, gain > 0) is med3(max(u, u * slope), -cl, cl),
        // u = acc * (scale * gain) + (bias + noise) * gain  --  one fma, one multiply, one max, one median per value
        float nzv[NT];
#pragma unroll
        for (int nt = 0; nt < NT; nt++) {
            const int row_l = (wpx * NT + nt) * G::RP + l31 / TWL, col_l = l31 % TWL;
            nzv[nt] = *(lds_f)(side + (size_t)(2 * G::EPS + row_l * G::TW + col_l) * 4);
        }
#pragma unroll
        for (int nt = 0; nt < NT; nt++) {
            if (dbg_ & 8) break;
            const int row_l = (wpx * NT + nt) * G::RP + l31 / TWL, col_l = l31 % TWL;
            const int oy = e_oy0 + row_l, ox = e_ox0 + col_l;
            const bool pix_ok = oy < p.OH && ox < p.OW;
            const float nz = nzv[nt] * noise_gain * gain;
            const int64_t pix_off = (int64_t)e_z * p.ws_slice + (int64_t)e_n * p.ys[0] + (int64_t)(oy * p.osy + e_ooy) * p.ys[2] + (int64_t)(ox * p.osx + e_oox) * p.ys[3];
#pragma unroll
            for (int mt = 0; mt < MT; mt++) {
                const int mloc = (wmx * MT + mt) * 32;
                float v[16];
#pragma unroll
                for (int g = 0; g < 4; g++) {
                    // (read where they are used -- 2 x 16 bytes of LDS per 4 values: held for the whole tile they were 64 registers,
                    // which the three-waves-per-SIMD form of the kernel does not have)
                    const int r0 = mloc + 8 * g + 4 * half;
                    f32x4 sgv = *(lds_f4)(side + (size_t)r0 * 4), bgv = *(lds_f4)(side + (size_t)(G::EPS + r0) * 4);
                    sgv = has_scale ? sgv * gain : f32x4{gain, gain, gain, gain};
                    bgv = bgv * gain;
#pragma unroll
                    for (int j = 0; j < 4; j++) {
                        const float u = fmaf(acc[mt][nt][4 * g + j], sgv[j], bgv[j] + nz);
                        v[4 * g + j] = __builtin_amdgcn_fmed3f(fmaxf(u, u * slope), -cl, cl);
                    }
                }
                if (p.out_mode == OUT_VEC16) {
                    // ys[1] == 1, Cout % 8 == 0: 8 consecutive couts of one pixel per lane after the half-wave exchange
                    const unsigned short* rp = (const unsigned short*)p.f.residual;
                    u32x2 o[4];
#pragma unroll
                    for (int g = 0; g < 4; g++) {
                        const int co = e_m0 + mloc + 8 * g + 4 * half;
                        if (rp) {
                            const bool ok = pix_ok && co < ce;
                            const u32x2 rv = *(const u32x2*)(rp + (ok ? pix_off + co : 0));
                            v[4 * g + 0] += HT::widen((unsigned short)(rv[0] & 0xffff));
                            v[4 * g + 1] += HT::widen((unsigned short)(rv[0] >> 16));
                            v[4 * g + 2] += HT::widen((unsigned short)(rv[1] & 0xffff));
                            v[4 * g + 3] += HT::widen((unsigned short)(rv[1] >> 16));
                        }
                        o[g][0] = HT::pack(v[4 * g + 0], v[4 * g + 1]);
                        o[g][1] = HT::pack(v[4 * g + 2], v[4 * g + 3]);
                    }
#pragma unroll
                    for (int g0 = 0; g0 < 4; g0 += 2) {
                        // lanes 32-63 of group g0 <-> lanes 0-31 of group g0+1: lower half then holds couts 8*g0 .. +7,
                        // upper half couts 8*(g0+1) .. +7 of its pixel
                        u32x2 a = o[g0], b = o[g0 + 1];
#pragma unroll
                        for (int d = 0; d < 2; d++) {
                            const auto r = __builtin_amdgcn_permlane32_swap(a[d], b[d], false, false);
                            a[d] = r[0]; b[d] = r[1];
                        }
                        const int co = e_m0 + mloc + 8 * (g0 + half);
                        const unsigned so = (pix_ok && co < ce && !(dbg_ & 1)) ? (unsigned)(pix_off + co) * 2u : SENTINEL;
                        __builtin_amdgcn_raw_buffer_store_b128(u32x4{a[0], a[1], b[0], b[1]}, yrsrc, (int)so, 0, 0);
                    }
                } else if (p.out_mode == OUT_VEC32) {
                    // float32 channels-last (the split-K workspace): a lane's four consecutive couts are one 16-byte store
#pragma unroll
                    for (int g = 0; g < 4; g++) {
                        const int co = e_m0 + mloc + 8 * g + 4 * half;
                        const unsigned so = (pix_ok && co < ce) ? (unsigned)(pix_off + co) * 4u : SENTINEL;
                        __builtin_amdgcn_raw_buffer_store_b128(u32x4{__builtin_bit_cast(unsigned, v[4 * g]), __builtin_bit_cast(unsigned, v[4 * g + 1]),
                                                                     __builtin_bit_cast(unsigned, v[4 * g + 2]), __builtin_bit_cast(unsigned, v[4 * g + 3])}, yrsrc, (int)so, 0, 0);
                    }
                } else {
#pragma unroll
                    for (int i = 0; i < 16; i++) {
                        const int co = e_m0 + mloc + (i & 3) + 8 * (i >> 2) + 4 * half;
                        if (pix_ok && co < ce) {
                            const int64_t off = pix_off + (int64_t)co * p.ys[1];
                            if (p.out_mode == OUT_SCALAR32) {
                                float u = v[i];
                                if (p.f.residual) u += ((const float*)p.f.residual)[off];
                                ((float*)p.y)[off] = u;
                            } else {
                                float u = v[i];
                                if (p.f.residual) u += HT::widen(((const unsigned short*)p.f.residual)[off]);
                                ((unsigned short*)p.y)[off] = (unsigned short)(HT::pack(u, 0.f) & 0xffff);
                            }
                        }
                    }
                }
            }
        }
    };

    // One loop, one issue site: the first NBUF - 1 passes only fill the pipeline (`it` < 0), every later pass waits for the oldest
    // chunk in flight, requests one more and multiplies; a tile's epilogue runs in the pass of its last chunk.
    int tile = blockIdx.x;
    int k_cur = 0;
    bool after_ep = false;
    if (!loader) {                                        // (two-role form: the loader waves must not carry 64 accumulator registers around)
#pragma unroll
        for (int mt = 0; mt < MT; mt++)
#pragma unroll
            for (int nt = 0; nt < NT; nt++)
#pragma unroll
                for (int k = 0; k < 16; k++) acc[mt][nt][k] = 0.f;
    }
    int n_stamp = 0;
    auto stamp = [&](int tag) __attribute__((always_inline)) {
        if (PG_CONV16_STAMPS && (dbg_ & 32) && blockIdx.x == 0 && wave == (dbg_ >> 8) && n_stamp < 4000) {
            const unsigned long long tm = __builtin_amdgcn_s_memtime();
            if (lane == 0) p.stamps[n_stamp] = (tm << 8) | (unsigned)tag;
            n_stamp++;
        }
    };
    if constexpr (SPLIT) {
        // chunks this workgroup multiplies: its tiles blockIdx.x, blockIdx.x + gridDim.x, ... x nchunks; chunk c lives in staging buffer c % NBUF
        // (NBUF = 3: requests two chunks ahead; NBUF = 2, the 32-channel chunks: one).
        // Barrier c (one per chunk, all twelve waves) says: chunk c has landed, everybody is done with chunk c - 1.
        const int my_chunks = ((total - (int)blockIdx.x + (int)gridDim.x - 1) / (int)gridDim.x) * nchunks;
        if (loader) {
            setup_maps();
            int req = 0;
#pragma unroll
            for (int j = 0; j < G::NBUF - 1; j++)                       // NBUF - 1 chunks ahead (the host only launches this form with nchunks >= 2)
                if (req < my_chunks) { issue_next(); req++; }
            for (int c = 0; c < my_chunks; c++) {
                // chunk c must have landed; what is younger in this wave's queue -- chunk c + 1's requests (and, in front of them, the
                // side loads of its tile when it is a tile's first chunk) -- may stay in flight
                stamp(1);
                if (G::NBUF > 2 && req > c + 1) vm_wait<LD_EFF * (G::NBUF - 2)>(); else vm_wait<0>();
                stamp(2);
                __builtin_amdgcn_s_barrier();
                stamp(3);
                if (req < my_chunks) { issue_next(); req++; }           // into the buffer of chunk c - 1: every multiplying wave is past it
                stamp(4);
            }
            return;
        }
        for (;;) {
            for (int k = 0; k < nchunks; k++) {
                stamp(1);
                __builtin_amdgcn_s_barrier();
                stamp(3);
                if (!(dbg_ & 4)) compute_chunk(cbuf);
                stamp(5);
                cbuf = cbuf == G::NBUF - 1 ? 0 : cbuf + 1;
            }
            prep_tile(tile, dpar, false);                               // coordinates of this tile for the epilogue (scalar unit)
            write_tile();
            stamp(6);
            if (tile + (int)gridDim.x >= total) break;
            tile += gridDim.x;
            dpar ^= 1;
#pragma unroll
            for (int mt = 0; mt < MT; mt++)
#pragma unroll
                for (int nt = 0; nt < NT; nt++)
#pragma unroll
                    for (int k = 0; k < 16; k++) acc[mt][nt][k] = 0.f;
        }
        return;
    }
#pragma unroll 1
    for (int it = -(G::NBUF - 1);; it++) {
        stamp(1);
        if (it >= 0) {
            // chunk `cbuf` must have landed: everything younger in this wave's queue may stay in flight -- the next chunk's
            // DMAs (inflight == 2) and, right after a tile's vector epilogue, its stores
            if (inflight > 1) { if (after_ep && p.out_mode == OUT_VEC16) vm_wait<G::DPC + EP_STORES>(); else vm_wait<G::DPC>(); }
            else              { if (after_ep && p.out_mode == OUT_VEC16) vm_wait<EP_STORES>(); else vm_wait<0>(); }
            after_ep = false;
            stamp(2);
            __builtin_amdgcn_s_barrier();                  // every wave's share has landed; every wave is done with the buffer requested next
            stamp(3);
        }
        issue_next();
        stamp(4);
        if (it < 0) continue;
        if (!(dbg_ & 4)) compute_chunk(cbuf);
        stamp(5);
        cbuf = cbuf == G::NBUF - 1 ? 0 : cbuf + 1;
        inflight--;
        if (++k_cur < nchunks) continue;
        k_cur = 0;

        write_tile();
        after_ep = true;
        stamp(6);
        if (tile + (int)gridDim.x >= total) break;
        tile += gridDim.x;
        dpar ^= 1;
        c_ahead--;
#pragma unroll
        for (int mt = 0; mt < MT; mt++)
#pragma unroll
            for (int nt = 0; nt < NT; nt++)
#pragma unroll
                for (int k = 0; k < 16; k++) acc[mt][nt][k] = 0.f;
    }
}

template <typename T, int KH, int KW, int S, int TWL, int WM, int MT, int NT, int KC, int NB, bool SPLIT = false>
int launch16(const Conv16Params& p0, hipStream_t s) {
    // 32-channel chunks exist in the two-role form only (their one-role staging buffers would not fit 160 KB): the caller checked the conditions
    if constexpr (KC == 32 && KH == 3 && KW == 3 && S == 1 && !SPLIT) return launch16<T, KH, KW, S, TWL, WM, MT, NT, KC, NB, true>(p0, s);
    else {
    typedef Geo16<KH, KW, S, TWL, WM, MT, NT, KC, NB, SPLIT> G;
    Conv16Params p = p0;
    p.tilesX = (p.OW + G::TW - 1) / G::TW;
    p.tilesY = (p.OH + G::TH - 1) / G::TH;
    p.mblocks = (p.Cout + G::BM - 1) / G::BM;              // (not CoutP: the packing pads to 64, a 32-cout block would be all padding)
    if (p.f.phase_cout && p.f.phase_cout % G::BM != 0) return PG_ERR_UNSUPPORTED;        // a cout block must not straddle two phases
    const int64_t tiles = (int64_t)p.N * p.tilesX * p.tilesY * p.mblocks * (p.ksplit > 1 ? p.ksplit : 1);
    if (tiles > 0x7fffffffLL) return PG_ERR_TOO_LARGE;
    p.total_tiles = (int)tiles;
    if (p.ksplit > 1 && p.kpart % KC != 0) return PG_ERR_INVALID_ARG;
    auto magic = [&](int d) -> unsigned { return d <= 1 ? 0u : (unsigned)((0x100000000ULL + (unsigned)d - 1) / (unsigned)d); };
    const int dmax = std::max(std::max(p.tilesX, p.tilesY), std::max(p.mblocks, p.ksplit > 1 ? p.ksplit : 1));
    if (tiles * dmax >= 0x100000000LL) return PG_ERR_TOO_LARGE;          // exactness bound of div_magic
    p.m_tilesX = magic(p.tilesX); p.m_tilesY = magic(p.tilesY); p.m_mblocks = magic(p.mblocks); p.m_ksplit = magic(p.ksplit);
    if ((int64_t)p.N * p.H * p.W * p.xC * 2 > 0x7fffffffLL) return PG_ERR_TOO_LARGE;      // whole-tensor descriptor of x
    const int64_t blocks = tiles < (int64_t)num_cu() ? tiles : (int64_t)num_cu();       // persistent: one workgroup per CU
    if constexpr (((KH == 3 && KW == 3) || (KH == 2 && KW == 2)) && S == 1 && NB == 3 && !SPLIT && TWL > 8) {      // the two-role form (round 4: also the 2x2 kernels of the merged transposed phases) (see the kernel): >= 2 chunks per tile; (the 8 x 8-pixel tile of the 8^2 layers measured slower with it: 44.5 vs 36.5 us)
        static const bool split_on = [] { const char* e = getenv("PG_CONV16_SPLIT"); return e ? atoi(e) != 0 : true; }();       // A/B switch
        const int cin_loop = p.ksplit > 1 ? p.kpart : p.Cin;
        if (split_on && (cin_loop + KC - 1) / KC >= 2) return launch16<T, KH, KW, S, TWL, WM, MT, NT, KC, NB, true>(p0, s);
    }
    auto kern = conv2d_mfma16<T, KH, KW, S, TWL, WM, MT, NT, KC, NB, SPLIT>;
    static PerDeviceOnce lds_attr;
    const hipError_t e = lds_attr.run([&] { return hipFuncSetAttribute((const void*)kern, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024); });
    if (e != hipSuccess) return (int)e;
    hipLaunchKernelGGL(kern, dim3((unsigned)blocks), dim3(SPLIT ? THREADS + 64 * LOADERS : THREADS), G::LDS_BYTES, s, p);
    return launch_status();
    }
}

// Low-resolution 3x3 layers (8^2 / 16^2 images of the wide blocks): the 16 x 32 pixel tile would be 7/8 or 1/2 padding -- matrix work
// and weight traffic spent on nothing.  1 = one 8 x 8 image x 128 couts per workgroup, 2 = 16 x 16 pixels x 64 couts (images up to 64 x 64: four times the tiles of the
// regular shape, so a quarter of the split-K shares and workspace traffic), 0 = the regular tile.  Shared by the launcher and the split-K planner.  PG_CONV16_SMALL=0 switches it off (A/B).
inline int small_tile16(int KH, int KW, int S, int OH, int OW, int Cout, int phase_cout = 0) {
    static const bool on = [] { const char* e = getenv("PG_CONV16_SMALL"); return e ? atoi(e) != 0 : true; }();
    if (!on || !((KH == 3 && KW == 3) || (KH == 2 && KW == 2)) || S != 1 || Cout < 128 || phase_cout % 64 != 0) return 0;      // (four-phase mode: whole cout blocks per phase)
    if (KH == 3 && OH <= 8 && OW <= 8 && phase_cout % 128 == 0) return 1;
    static const int lim2 = [] { const char* e = getenv("PG_CONV16_SMALL2_MAX"); return e ? atoi(e) : 32; }();      // (round 4, config 5: 16 -> 2.74, 32 -> 2.72, 64 -> 2.68 ms/step; round 5, with split-K plans at one workgroup per CU: 32 -> 1.590 / 1.595, 64 -> 1.599 / 1.598, 16 -> 1.611 / 1.598)
    if (OH <= lim2 && OW <= lim2) return 2;
    return 0;
}

// M-tile count by the width of the layer (32-cout blocks for narrow layers: no MFMA spent on padding rows)
template <typename T, int KH, int KW, int S, int NT, int KC, int NB>
int launch16_mt(const Conv16Params& p, hipStream_t s) {
    if constexpr (((KH == 3 && KW == 3) || (KH == 2 && KW == 2)) && S == 1) {
        const int sm = small_tile16(KH, KW, S, p.OH, p.OW, p.Cout, p.f.phase_cout);
        if (sm == 1) {                                                               // TH x TW = 8 x 8, BM = 128
            if constexpr (KC == 32) return PG_ERR_UNSUPPORTED;                       // (no 32-channel-chunk form of this tile: launch16_k3s1 does not send it here)
            else return launch16<T, KH, KW, S, 8, 4, 1, 1, KC, NB>(p, s);
        }
        if (sm == 2) return launch16<T, KH, KW, S, 16, 1, 2, 1, KC, NB>(p, s);       // TH x TW = 16 x 16, BM = 64
    }
    if ((p.f.phase_cout ? p.f.phase_cout : p.Cout) <= 32) return launch16<T, KH, KW, S, 32, 1, 1, NT, KC, NB>(p, s);
    return launch16<T, KH, KW, S, 32, 1, 2, NT, KC, NB>(p, s);
}

template <int KH, int KW, int S, int NT, int KC, int NB = 3>
int launch16_dt(const Conv16Params& p, int dtype, hipStream_t s) {
    if (dtype == PG_BF16) return launch16_mt<bf16_t, KH, KW, S, NT, KC, NB>(p, s);
    if (dtype == PG_F16) return launch16_mt<f16_t, KH, KW, S, NT, KC, NB>(p, s);
    return PG_ERR_INVALID_ARG;
}

// One entry per geometry family, each compiled in its own translation unit (conv2d16_inst_*.hip).
int launch16_k3s1(const Conv16Params& p, int dtype, hipStream_t s);
int launch16_k1s1(const Conv16Params& p, int dtype, hipStream_t s);
int launch16_k2x2(const Conv16Params& p, int dtype, hipStream_t s);
int launch16_k2x1(const Conv16Params& p, int dtype, hipStream_t s);
int launch16_k1x2(const Conv16Params& p, int dtype, hipStream_t s);
int launch16_k3s2(const Conv16Params& p, int dtype, hipStream_t s);

}  // namespace pgconv16
