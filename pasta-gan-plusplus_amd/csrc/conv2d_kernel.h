// fp32 NCHW convolution for gfx950 as an implicit GEMM on v_mfma_f32_32x32x2_f32
// (exact f32: a k-ordered fmaf chain, MI355X_MICROARCH.md "Matrix cores"), with the
// StyleGAN2 / SPADE elementwise work folded into its prologue and epilogue.
//
// What it replaces in the reference: the cuDNN call behind conv2d_gradfix.conv2d
// (torch_utils/ops/conv2d_gradfix.py:35-43, conv2d_resample.py:29-54), plus, when the
// fusion struct is used, the surrounding `x * styles`, `fma(x, dcoefs, noise)`
// (training/networks.py:73-82), `bias_act` (networks.py:170-179, 1623-1635) and the residual
// adds (networks.py:315, 1903).
//
// GEMM view:  M = Cout (rows of D, MFMA A operand = weights)
//             N = output pixels (columns of D = lanes, so stores are contiguous in x)
//             K = Cin * KH * KW, walked as (channel pair) x (tap): one MFMA consumes the two
//                 channels 2j, 2j+1 of one tap (lanes 0-31 hold channel 2j, lanes 32-63
//                 channel 2j+1 -- the k index of the 32x32x2 operand layout).
// Workgroup = 256 threads = 4 waves; output tile = BM couts x (8 rows x 32 cols) of one
// image; wave w owns rows 2w, 2w+1 (two 32-pixel N tiles) x BM/32 M tiles.
// Per K chunk of KC input channels: the input halo tile [KC][IH_T][IW_T] and the packed
// weight slab [KC][taps][BM] are prefetched global -> registers while the previous chunk
// is being multiplied, then written to LDS (one buffer, two barriers per chunk).  All LDS
// operand reads are ds_read_b32 of 32 consecutive dwords per half-wave: conflict-free.
//
// Roofline: MFMA-bound.  Algorithmic FLOPs = 2*N*Cout*OH*OW*Cin*KH*KW against the 157.3
// TFLOP/s f32 matrix peak.

#pragma once
#include <type_traits>
#ifndef DIRECT_EXP
#define DIRECT_EXP 0     // dev ablations (tools/direct_variants.py; results wrong by design): 1 no output stores, 2 no halo DMA
#endif
#include "pg_common.h"

namespace pgconv {

using namespace pg;

typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef float f32x4 __attribute__((ext_vector_type(4)));

struct ConvParams {
    const float* x; const float* wp; float* y;
    int N, Cin, H, W, Cout, CoutP, OH, OW;
    int pad_y, pad_x;
    int64_t ys[4];
    int osy, osx, ooy, oox;
    int tilesX, tilesY, mblocks, total_tiles;
    int in_xform;          // prologue bias/act/gain/clamp stage on
    int ksplit, kpart;     // split-K: `ksplit` workgroups share one output tile, each reducing `kpart` input channels into its own
    int64_t ws_slice;      // slice (ws_slice floats apart) of the partial-sum workspace that y then points to; 1 = off
    int rowpair_pack;      // the packed weights are in the ROWPAIR form (pg_conv2d_pack_weight makes it for every 7x7 kernel with Cin = 3)
    int rowpair;           // 7x7, Cin = 3 (see conv2d_mfma: ROWPAIR): the fourth channel slot carries channel 2 one row down
    int wino_gmap;         // Winograd: interior tiles take the tile-independent gather map from LDS (0 = off: A/B switch PG_WINO_GMAP=0)
    pg_conv2d_fusion f;
};

constexpr int TH = 8, TW = 32;   // output tile of one workgroup (rows x cols)

typedef int i32x4 __attribute__((ext_vector_type(4)));

// ---- LDS-DMA (global -> LDS without staging registers), issued from inline asm.
// hipcc (ROCm 7.2) tracks the builtin forms so conservatively that it puts `s_waitcnt vmcnt(0)` in front of every
// global_load_lds, serialising the transfers; from asm the compiler counts nothing, so the kernel waits itself
// (`dma_wait_all()` before the barrier that precedes the first ds_read of the landed buffer).
// LDS destination = M0 (wave-uniform byte offset) + lane * size; M0 is written in the same statement that uses it
// and restored, as it is compiler-reserved (cdna_hip_programming.md section 5.7).
__device__ __forceinline__ unsigned lds_offset(const float* p) {
    return (unsigned)(uintptr_t)(const __attribute__((address_space(3))) float*)p;
}
__device__ __forceinline__ void dma_dword(i32x4 rsrc, unsigned lds_byte, unsigned voff_bytes, int soff_bytes) {
    unsigned keep;
    asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %1\n\ts_nop 0\n\tbuffer_load_dword %2, %3, %4 offen lds\n\ts_mov_b32 m0, %0"
                 : "=&s"(keep) : "s"(lds_byte), "v"(voff_bytes), "s"(rsrc), "s"(soff_bytes) : "memory");
}
__device__ __forceinline__ void dma_dwordx4(const float* gsrc, unsigned lds_byte) {
    unsigned keep;
    asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %1\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %2, off\n\ts_mov_b32 m0, %0"
                 : "=&s"(keep) : "s"(lds_byte), "v"(gsrc) : "memory");
}
// 16-byte form through a buffer descriptor (range check = zero fill): used by the Winograd kernels and the up-conv's aligned halo rows
__device__ __forceinline__ void dma_dwordx4_buf(i32x4 rsrc, unsigned lds_byte, unsigned voff_bytes, int soff_bytes) {
    unsigned keep;
    asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %1\n\ts_nop 0\n\tbuffer_load_dwordx4 %2, %3, %4 offen lds\n\ts_mov_b32 m0, %0"
                 : "=&s"(keep) : "s"(lds_byte), "v"(voff_bytes), "s"(rsrc), "s"(soff_bytes) : "memory");
}
__device__ __forceinline__ void dma_wait_all() { asm volatile("s_waitcnt vmcnt(0)" ::: "memory"); }

// Once-per-tile scalar fetches (scales, bias) of the persistent loop.  They are inline asm with their own wait on purpose: a
// load the compiler can see leaves a pending-VMEM mark on its destination register, and when that register is later reused
// by the chunk loop's LDS reads the compiler protects the reuse with `s_waitcnt vmcnt(0)` at the top of EVERY chunk -- which
// also waits for the halo DMA requested a moment earlier and for the whole U ring (found in the round-1 build's assembly).
__device__ __forceinline__ float ld_opaque(const float* ptr) {
    float v;
    asm volatile("global_load_dword %0, %1, off\n\ts_waitcnt vmcnt(0)" : "=v"(v) : "v"(ptr) : "memory");
    return v;
}


template <int KH, int KW, int S, int BM, int KC>
struct Geo {
    static constexpr int T = KH * KW;
    static constexpr int IH_T = (TH - 1) * S + KH;
    static constexpr int IW_T = (TW - 1) * S + KW;
    static constexpr int PLANE = IH_T * IW_T;
    static constexpr int NX = KC * PLANE;                 // staged input floats per chunk
    static constexpr int XPT = (NX + 255) / 256;          // dword DMA instructions per thread per chunk
    static constexpr int NW4 = KC * T * BM / 4;           // staged weight float4s per chunk
    static constexpr int WPT = (NW4 + 255) / 256;         // 16-byte DMA instructions per thread per chunk
    static constexpr int MT = BM / 32;
    static constexpr int NT = 2;
    static constexpr int LDS_X = XPT * 256;               // floats; padded to whole wave-instructions
    static constexpr int LDS_W = WPT * 256 * 4;
    static constexpr int LDS_BUF = LDS_X + LDS_W;         // one staging buffer (floats)
    static constexpr size_t LDS_BYTES = (size_t)2 * LDS_BUF * 4;   // double buffered
    // workgroups per CU the staging buffers leave room for (the launcher computes the same from the full LDS size): geometries that
    // cannot have four anyway -- the 7x7 stem: 68 KB of weights + halo per workgroup -- get the registers of the waves that cannot exist
    static constexpr int MIN_BLOCKS = LDS_BYTES * 4 <= 156 * 1024 ? 4 : LDS_BYTES * 3 <= 156 * 1024 ? 3 : LDS_BYTES * 2 <= 156 * 1024 ? 2 : 1;
};

// The fused activations are linear / relu / lrelu (everything the synthesis path uses): one select,
// v > 0 ? v : v * slope.  Other activations are rejected by pg_conv2d_forward (PG_ERR_UNSUPPORTED); callers
// run bias_act separately for those.
__device__ __forceinline__ float act_slope(int act, float alpha) { return act == PG_ACT_LINEAR ? 1.f : (act == PG_ACT_RELU ? 0.f : alpha); }

// MODE: 0 = plain input (the B operand goes from LDS to the MFMA untouched: VALU instructions cost matrix-pipe time),
//       1 = per-(n, channel) input scale (modulated convolution), 2 = scale + pre-activation (XF).
template <int KH, int KW, int S, int BM, int KC, int MODE>
__global__ __launch_bounds__(256, (Geo<KH, KW, S, BM, KC>::MIN_BLOCKS)) void conv2d_mfma(ConvParams p) {
    constexpr bool XF = MODE == 2;
    typedef Geo<KH, KW, S, BM, KC> G;
    extern __shared__ __attribute__((aligned(16))) float smem[];
    // LDS (one array): two staging buffers { xs[KC][IH_T][IW_T] (+pad), ws[KC][T][BM] (+pad) }, the prologue
    // scale of two consecutive tiles cs[2][cin_loop], the epilogue constants ep_scale/ep_bias[BM].
    const int cin_loop = p.ksplit > 1 ? p.kpart : ((p.Cin + KC - 1) / KC) * KC;      // channels this workgroup reduces per tile
    const int nchunks = cin_loop / KC;
    float* cs0 = smem + 2 * G::LDS_BUF;
    float* ep0 = cs0 + 2 * cin_loop;           // epilogue constants of two consecutive tiles [2][BM + BM]: the next tile's are written
                                               // (single-chunk tiles: immediately) while slow waves still read this tile's

    const int t = threadIdx.x, lane = t & 63;
    const int wave = __builtin_amdgcn_readfirstlane(t >> 6);
    const unsigned smem_b = __builtin_amdgcn_readfirstlane(lds_offset(smem));
    const int half = lane >> 5, l31 = lane & 31;
    const int HW = p.H * p.W;
    const int total = p.total_tiles;
    const int q8 = total >> 3, r8 = total & 7;

    // ---- state of the tile whose chunks are being requested
    int n = 0, oy0 = 0, ox0 = 0, m0 = 0, cbeg = 0, zsl = 0;      // cbeg / zsl: first channel and workspace slice of a split-K share
    unsigned xoff[G::XPT];
    // ROWPAIR (the network's 7x7 stem, Cin = 3, K chunk = 2 channels): chunk 1 would multiply channel 2 beside an all-zero channel 3.
    // Instead its second slot carries channel 2 ONE ROW DOWN and the pack kernel puts w[.., 2, ky + 1, kx] into "channel 3" of tap
    // (ky, kx) for even ky: the MFMA's two k-slices then are the taps (ky, kx) and (ky + 1, kx) of channel 2, and the chunk walks the
    // kernel rows 0, 2, 4, 6 only -- 49 + 28 instead of 98 MFMA steps per tile and register tile.  Operand reads are unchanged (the
    // shift sits in the staging offsets); without the mode the same pack is still correct (channel 3 of x reads as zero).
    constexpr bool ROWPAIR = KH == 7 && KW == 7 && S == 1 && KC == 2 && MODE == 0;
    unsigned xoff2[ROWPAIR ? G::XPT : 1];
    i32x4 xrsrc, xrsrc2;             // image n of x (channels [0, split)) and of the optional second source x2
    const int split = p.f.x2 ? p.f.cin_split : p.Cin;

    // Tile -> (n, tile_y, tile_x, m-block), XCD-aware: workgroups sharing an XCD (id % 8) walk one contiguous
    // range of logical tiles, so neighbouring tiles and the m-blocks of one tile hit the same L2.
    // Staging map of the halo tile: BYTE offsets into image n, or a sentinel >= 2^31 for halo elements outside
    // the image.  The tile is fetched by buffer loads whose hardware range check returns 0 for the sentinel AND
    // for channels >= Cin, so zero padding costs neither a branch nor a select.
    auto prep_tile = [&](int tile, float* cs) {
        const int xcd = tile & 7;
        int L = (xcd < r8 ? xcd * (q8 + 1) : r8 * (q8 + 1) + (xcd - r8) * q8) + (tile >> 3);
        if (p.ksplit > 1) { zsl = L % p.ksplit; L /= p.ksplit; cbeg = zsl * p.kpart; }     // the shares of one tile run side by side
        const int mb = L % p.mblocks; L /= p.mblocks;
        const int tx = L % p.tilesX; L /= p.tilesX;
        const int ty = L % p.tilesY;
        n = L / p.tilesY;
        oy0 = ty * TH; ox0 = tx * TW; m0 = mb * BM;
        const float* in_scale = p.f.in_scale ? p.f.in_scale + (int64_t)n * p.Cin + cbeg : nullptr;
        if (MODE != 0)
            for (int c = t; c < cin_loop; c += 256) cs[c] = (in_scale && cbeg + c < p.Cin) ? ld_opaque(in_scale + c) : 1.f;
        // Opaque copy of the thread id: without it the compiler hoists the tile-independent index maths of every
        // element out of the persistent loop and keeps ~20 values live in VGPRs (spilling at 4 waves/SIMD).
        int tt = t;
        asm volatile("" : "+v"(tt));
#pragma unroll
        for (int i = 0; i < G::XPT; i++) {
            const int e = tt + 256 * i;
            const int c = e / G::PLANE, rem = e % G::PLANE;
            const int rr = rem / G::IW_T, cc = rem % G::IW_T;
            const int gy = oy0 * S - p.pad_y + rr, gx = ox0 * S - p.pad_x + cc;
            const bool ok = e < G::NX && gy >= 0 && gy < p.H && gx >= 0 && gx < p.W;
            xoff[i] = ok ? (unsigned)(c * HW + gy * p.W + gx) * 4u : 0x80000000u;
            if constexpr (ROWPAIR) {
                const bool ok2 = e < G::NX && gy + c >= 0 && gy + c < p.H && gx >= 0 && gx < p.W;      // slot 1 = the same channel, one row down
                xoff2[i] = ok2 ? (unsigned)((gy + c) * p.W + gx) * 4u : 0x80000000u;
            }
        }
        // raw buffer descriptor of image n: base, stride 0, num_records = bytes, flags as make_buffer_rsrc's 0x00020000
        const uint64_t base = (uint64_t)(uintptr_t)(p.x + ((int64_t)n * split + cbeg) * HW);
        xrsrc[0] = __builtin_amdgcn_readfirstlane((int)(unsigned)base);
        xrsrc[1] = __builtin_amdgcn_readfirstlane((int)(unsigned)(base >> 32) & 0xffff);
        xrsrc[2] = (split - cbeg > 0 ? split - cbeg : 0) * HW * 4;
        xrsrc[3] = 0x00020000;
        xrsrc2 = xrsrc;
        if (p.f.x2) {                      // channels [split, Cin) come from x2: conv(cat([x, x2], 1)) without the copy
            const uint64_t base2 = (uint64_t)(uintptr_t)(p.f.x2 + (int64_t)n * (p.Cin - split) * HW);
            xrsrc2[0] = __builtin_amdgcn_readfirstlane((int)(unsigned)base2);
            xrsrc2[1] = __builtin_amdgcn_readfirstlane((int)(unsigned)(base2 >> 32) & 0xffff);
            xrsrc2[2] = (p.Cin - split) * HW * 4;
        }
    };

    // Request one K chunk: global -> LDS directly (no staging registers, no ds_write).  Each wave-instruction
    // writes 64 lanes x 4 B (input) or 64 x 16 B (weights) contiguously at a wave-uniform LDS base; the input
    // gather address is per lane.
    auto issue_chunk = [&](int c0, int buf) {
        const unsigned xs_b = smem_b + (unsigned)(buf * G::LDS_BUF + 64 * wave) * 4u;            // bytes, wave-uniform
        const unsigned ws_b = smem_b + (unsigned)(buf * G::LDS_BUF + G::LDS_X + 256 * wave) * 4u;
        if (c0 < split) {                                   // wave-uniform: split is a multiple of the chunk size
            const int soff = c0 * HW * 4;
#pragma unroll
            for (int i = 0; i < G::XPT; i++) {
                unsigned xo = xoff[i];
                if constexpr (ROWPAIR) { if (p.rowpair && c0 == 2) xo = xoff2[i]; }
                if (!(DIRECT_EXP & 2)) dma_dword(xrsrc, xs_b + 1024u * i, xo, soff);
            }
        } else {
            const int soff = (c0 - split) * HW * 4;
#pragma unroll
            for (int i = 0; i < G::XPT; i++) dma_dword(xrsrc2, xs_b + 1024u * i, xoff[i], soff);
        }
        const float* wb = p.wp + (int64_t)(cbeg + c0) * G::T * p.CoutP + m0;
#pragma unroll
        for (int i = 0; i < G::WPT; i++) {
            int e4 = t + 256 * i;
            if (G::NW4 % 256 != 0 && e4 >= G::NW4) e4 = G::NW4 - 1;     // clamp: the pad lanes copy a duplicate
            const int row = (e4 * 4) / BM, col = (e4 * 4) % BM;
            dma_dwordx4(wb + (int64_t)row * p.CoutP + col, ws_b + 4096u * i);
        }
    };

    // prologue activation folded to 3-4 VALU ops per operand: t = x * (scale * gain); lrelu/relu/linear(t) = max(t, t * slope)
    // for gain > 0 and 0 <= slope <= 1 (the host guarantees both); clamp by one v_med3.
    const float in_slope = act_slope(p.f.in_act, p.f.in_alpha);
    const float in_cl = p.f.in_clamp >= 0.f ? p.f.in_clamp : __builtin_inff();
    const float in_gain = XF ? p.f.in_gain : 1.f;

    f32x16 acc[G::MT][G::NT];

    // Multiply one chunk out of LDS buffer `buf`.  The prologue (modulation scale, SPADE pre-activation without
    // bias: act(0) = 0 keeps the zero padding) is applied to the B operand as it is read: a few VALU ops per
    // 64-cycle MFMA pair, on the otherwise idle vector pipe.
    auto compute_chunk = [&](int buf, const float* cs, int c0) {
        const float* a_base = smem + buf * G::LDS_BUF + G::LDS_X + half * (G::T * BM) + l31;
        const float* b_base = smem + buf * G::LDS_BUF + half * G::PLANE + (wave * 2 * S) * G::IW_T + l31 * S;
#pragma unroll
        for (int cp = 0; cp < KC / 2; cp++) {
            const float sc = MODE != 0 ? cs[c0 + 2 * cp + half] * in_gain : 1.f;
#pragma unroll
            for (int ky = 0; ky < KH; ky++) {
                if constexpr (ROWPAIR) { if (p.rowpair && c0 == 2 && (ky & 1)) continue; }      // (wave-uniform) the odd rows ride in the second k-slice
#pragma unroll
                for (int kx = 0; kx < KW; kx++) {
                    float a[G::MT], b[G::NT];
#pragma unroll
                    for (int mt = 0; mt < G::MT; mt++) a[mt] = a_base[((2 * cp) * G::T + ky * KW + kx) * BM + mt * 32];
#pragma unroll
                    for (int nt = 0; nt < G::NT; nt++) {
                        float v = b_base[(2 * cp) * G::PLANE + (nt * S + ky) * G::IW_T + kx];
                        if (MODE != 0) v *= sc;
                        if (XF) v = __builtin_amdgcn_fmed3f(fmaxf(v, v * in_slope), -in_cl, in_cl);
                        b[nt] = v;
                    }
#pragma unroll
                    for (int mt = 0; mt < G::MT; mt++)
#pragma unroll
                        for (int nt = 0; nt < G::NT; nt++)
                            acc[mt][nt] = __builtin_amdgcn_mfma_f32_32x32x2f32(a[mt], b[nt], acc[mt][nt], 0, 0, 0);
                }
                // keep the scheduler from hoisting every LDS read of the chunk to the top (that costs > 100 VGPRs
                // and forces spills at 4 waves/SIMD); one kernel row of operands in flight is plenty
                __builtin_amdgcn_sched_barrier(0);
            }
        }
    };

    const float gain = p.f.gain;
    const float cl = p.f.clamp >= 0.f ? p.f.clamp : __builtin_inff();
    const float slope = act_slope(p.f.act, p.f.alpha);
    const bool plain_tail = !p.f.noise && slope == 1.f && gain == 1.f && p.f.clamp < 0.f;      // wave-uniform

    // ---- persistent loop over this workgroup's tiles: one continuous stream of K chunks through the two LDS
    // buffers, one barrier per chunk.  While chunk g is multiplied, chunk g+1 is in flight -- the next chunk of
    // the same tile or, on a tile's last chunk, the FIRST chunk of the next tile, so the next tile's load latency
    // and this tile's output stores both hide behind MFMAs.
    int tile = blockIdx.x;
    int par = 0;                       // which cs[] half the current tile uses
    int g = 0;                         // running chunk counter -> LDS buffer g & 1
    prep_tile(tile, cs0);
    issue_chunk(0, 0);
    dma_wait_all();
    __syncthreads();                   // DMA landed, cs published
    while (true) {
#pragma unroll
        for (int mt = 0; mt < G::MT; mt++)
#pragma unroll
            for (int nt = 0; nt < G::NT; nt++)
#pragma unroll
                for (int k = 0; k < 16; k++) acc[mt][nt][k] = 0.f;

        int e_n = n, e_oy0 = oy0, e_ox0 = ox0, e_m0 = m0, e_z = zsl;
        bool has_next = false;
        int next = tile;
        const float* cs_cur = cs0 + par * cin_loop;
        float* ep_scale = ep0 + par * 2 * BM;
        float* ep_bias = ep_scale + BM;
        __builtin_amdgcn_s_setprio(1);                // K loop above other workgroups' epilogues on this SIMD
        for (int k = 0; k < nchunks; k++, g++) {
            const int buf = g & 1;
            if (k + 1 < nchunks) {
                issue_chunk((k + 1) * KC, buf ^ 1);        // that buffer was last read in the previous iteration (barrier below)
            } else {
                // last chunk of this tile: publish its epilogue constants, then stage the next tile
                e_n = n; e_oy0 = oy0; e_ox0 = ox0; e_m0 = m0; e_z = zsl;
                if (p.f.spade_x) {                          // SPADE mode: per-(n, channel) mean / rstd of the normalised tensor
                    if (t < 32) {
                        const int ch = (e_m0 >> 1) + t;     // this tile's 32 output channels
                        ep_scale[t] = ld_opaque(p.f.spade_mean + e_n * (p.Cout >> 1) + ch);
                        ep_bias[t] = ld_opaque(p.f.spade_rstd + e_n * (p.Cout >> 1) + ch);
                    }
                } else if (t < BM) {
                    const int co = e_m0 + t;
                    const bool ok = co < p.Cout;
                    const int cc = ok ? co : 0;
                    const float sc = p.f.out_scale ? ld_opaque(p.f.out_scale + (int64_t)e_n * p.Cout + cc) : 1.f;
                    const float bi = p.f.bias ? ld_opaque(p.f.bias + cc) : 0.f;
                    ep_scale[t] = ok ? sc : 0.f;
                    ep_bias[t] = ok ? bi : 0.f;
                }
                next = tile + gridDim.x;
                has_next = next < total;
                if (has_next) {
                    prep_tile(next, cs0 + (par ^ 1) * cin_loop);
                    issue_chunk(0, buf ^ 1);
                }
            }
            compute_chunk(buf, cs_cur, k * KC);
            dma_wait_all();            // the chunk requested above has landed (also drains this wave's older stores)
            __syncthreads();
        }

        __builtin_amdgcn_s_setprio(0);
        // ---- epilogue of this tile: D layout col = lane&31 (pixel), row = (reg&3) + 8*(reg>>2) + 4*(lane>>5) (cout).
        // Per-cout constants come from LDS; every load is unconditional from a clamped, always-valid address.  The
        // outputs are walked in groups of 4 consecutive couts; the group's extra operand (residual, or the tensor being
        // SPADE-normalised) is fetched ONE GROUP AHEAD, so a group never waits a full memory round trip (nor for the
        // previous group's stores: the loads are older than those stores in the in-order vmcnt queue).
        const int ox = e_ox0 + l31;
        const int oxc = ox < p.OW ? ox : p.OW - 1;
        const int cstride = (int)p.ys[1];
        const bool spade = G::MT == 2 && p.f.spade_x != nullptr;
        const float* extra = spade ? p.f.spade_x : p.f.residual;
        int pix_off[G::NT];
        bool pix_ok[G::NT];
        float nz[G::NT];
#pragma unroll
        for (int nt = 0; nt < G::NT; nt++) {
            const int oy = e_oy0 + wave * 2 + nt;
            pix_ok[nt] = oy < p.OH && ox < p.OW;
            const int oyc = oy < p.OH ? oy : p.OH - 1;
            nz[nt] = p.f.noise ? p.f.noise[(int)(e_n * p.f.noise_batch_stride) + oyc * p.OW + oxc] * p.f.noise_gain : 0.f;
            pix_off[nt] = (int)((int64_t)e_z * p.ws_slice + (int64_t)e_n * p.ys[0] + (int64_t)(oyc * p.osy + p.ooy) * p.ys[2] + (int64_t)(oxc * p.osx + p.oox) * p.ys[3]);
        }
        constexpr int NG = G::NT * G::MT * 4;
        // fetch the extra operand of group g (SPADE mode: this tile's 32 output channels live at m0/2); channel clamped per
        // element so every address is valid
        const int ex_lim = (spade ? (p.Cout >> 1) : p.Cout) - 1;
        auto fetch_extra = [&](int g, float (&dst)[4]) {
            const int nt = g / (G::MT * 4), mt = (g / 4) % G::MT, kq = g % 4;
            const int co0 = spade ? (e_m0 >> 1) + 8 * kq + 4 * half : e_m0 + mt * 32 + 8 * kq + 4 * half;
#pragma unroll
            for (int j = 0; j < 4; j++) dst[j] = extra[pix_off[nt] + (co0 + j < ex_lim ? co0 + j : ex_lim) * cstride];
        };
        float ex_next[4] = {0.f, 0.f, 0.f, 0.f};
        if (extra) fetch_extra(0, ex_next);
#pragma unroll
        for (int g = 0; g < NG; g++) {
            const int nt = g / (G::MT * 4), mt = (g / 4) % G::MT, kq = g % 4;
            float ex[4];
#pragma unroll
            for (int j = 0; j < 4; j++) ex[j] = ex_next[j];
            if (extra && g + 1 < NG) fetch_extra(g + 1, ex_next);   // prefetch the next group's extra operand
            const int row0 = mt * 32 + 8 * kq + 4 * half;
            if (spade) {
                // SPADE combine: the packed weights interleave 32 gamma rows (M-tile 0) with the 32 beta rows of the same
                // channels (M-tile 1), so one lane holds gamma and beta of one (channel, pixel):
                //   y = (x - mean) * rstd * (1 + gamma) + beta          (networks.py:1715-1722)
                if (mt == 0) {
                    const int r0 = 8 * kq + 4 * half;
                    const f32x4 mu4 = *(const f32x4*)(ep_scale + r0);
                    const f32x4 rs4 = *(const f32x4*)(ep_bias + r0);
                    const int o = pix_off[nt] + ((e_m0 >> 1) + r0) * cstride;
#pragma unroll
                    for (int j = 0; j < 4; j++) {
                        float v = (ex[j] - mu4[j]) * rs4[j] * (1.f + acc[0][nt][4 * kq + j]) + acc[G::MT - 1][nt][4 * kq + j];
                        v = v > 0.f ? v : v * slope;                 // the consumer's pre-activation (Spade_Conv2dLayer), when folded in
                        v = fminf(fmaxf(v * gain, -cl), cl);
                        if (pix_ok[nt]) p.y[o + j * cstride] = v;
                    }
                }
            } else {
                const f32x4 sc4 = *(const f32x4*)(ep_scale + row0);
                const f32x4 bi4 = *(const f32x4*)(ep_bias + row0);
                const int o = pix_off[nt] + (e_m0 + row0) * cstride;
                if (plain_tail) {                  // no noise, identity activation chain: one fma (+ the residual add) per value --
                                                   // VALU instructions are matrix-pipe time, and a 1x1 layer has one output per 64 MACs
#pragma unroll
                    for (int j = 0; j < 4; j++) {
                        const float v = fmaf(acc[mt][nt][4 * kq + j], sc4[j], bi4[j]) + ex[j];
                        if ((DIRECT_EXP & 1) ? (v == 12345.678f) : (pix_ok[nt] && e_m0 + row0 + j < p.Cout)) p.y[o + j * cstride] = v;
                    }
                } else {
#pragma unroll
                    for (int j = 0; j < 4; j++) {
                        float v = acc[mt][nt][4 * kq + j] * sc4[j] + nz[nt] + bi4[j];
                        v = v > 0.f ? v : v * slope;
                        v = fminf(fmaxf(v * gain, -cl), cl) + ex[j];
                        if ((DIRECT_EXP & 1) ? (v == 12345.678f) : (pix_ok[nt] && e_m0 + row0 + j < p.Cout)) p.y[o + j * cstride] = v;
                    }
                }
            }
            __builtin_amdgcn_sched_barrier(0);               // do not hoist later groups' reads (register pressure)
        }
        if (!has_next) break;
        tile = next;
        par ^= 1;
    }
}

template <int KH, int KW, int S, int BM, int KC, int MODE>
int launch_conv_xf(const ConvParams& p0, hipStream_t s) {
    typedef Geo<KH, KW, S, BM, KC> G;
    ConvParams p = p0;
    p.tilesX = (p.OW + TW - 1) / TW;
    p.tilesY = (p.OH + TH - 1) / TH;
    p.mblocks = p.CoutP / BM;
    p.rowpair = KH == 7 && KW == 7 && S == 1 && KC == 2 && MODE == 0 && p.Cin == 3 && !p.f.x2 && p.ksplit <= 1 && p.rowpair_pack;
    const int64_t tiles = (int64_t)p.N * p.tilesX * p.tilesY * p.mblocks * (p.ksplit > 1 ? p.ksplit : 1);
    if (tiles > 0x7fffffffLL) return PG_ERR_TOO_LARGE;
    p.total_tiles = (int)tiles;
    const int cin_loop = p.ksplit > 1 ? p.kpart : ((p.Cin + KC - 1) / KC) * KC;
    if (p.ksplit > 1 && (p.kpart % KC != 0 || p.f.x2)) return PG_ERR_INVALID_ARG;
    const size_t lds = G::LDS_BYTES + ((size_t)2 * cin_loop + 4 * BM) * sizeof(float);
    if (lds > 160 * 1024) return PG_ERR_UNSUPPORTED;
    // persistent grid: as many workgroups as stay resident (4 per CU by registers; fewer if LDS-limited);
    // every workgroup walks tiles id, id + grid, ...
    int per_cu = (int)((160 * 1024) / lds);
    if (per_cu > G::MIN_BLOCKS) per_cu = G::MIN_BLOCKS;       // the kernel's launch bounds (registers)
    if (per_cu < 1) per_cu = 1;
    const int64_t blocks = tiles < (int64_t)num_cu() * per_cu ? tiles : (int64_t)num_cu() * per_cu;
    static PerDeviceOnce lds_attr;
    const hipError_t e = lds_attr.run([] { return hipFuncSetAttribute((const void*)conv2d_mfma<KH, KW, S, BM, KC, MODE>, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024); });
    if (e != hipSuccess) return (int)e;
    hipLaunchKernelGGL((conv2d_mfma<KH, KW, S, BM, KC, MODE>), dim3((unsigned)blocks), dim3(256), lds, s, p);
    return launch_status();
}

// XFORM: also instantiate the variant with the prologue activation (only the geometries SPADE layers use).
template <int KH, int KW, int S, int BM, int KC, bool XFORM>
int launch_conv(const ConvParams& p, hipStream_t s) {
    if (p.in_xform) {
        if constexpr (XFORM) return launch_conv_xf<KH, KW, S, BM, KC, 2>(p, s);
        else return PG_ERR_UNSUPPORTED;
    }
    if (p.f.in_scale) return launch_conv_xf<KH, KW, S, BM, KC, 1>(p, s);
    return launch_conv_xf<KH, KW, S, BM, KC, 0>(p, s);
}

template <int KH, int KW, int S, int KC, bool XFORM = false>
int launch_bm(const ConvParams& p, hipStream_t s) {
    // 64-cout tiles unless that leaves most of the chip idle (low-resolution layers: a handful of pixel tiles, latency-bound
    // K loops): 32-cout tiles double the number of workgroups
    const int64_t tiles64 = (int64_t)p.N * ((p.OW + TW - 1) / TW) * ((p.OH + TH - 1) / TH) * (p.CoutP / 64) * (p.ksplit > 1 ? p.ksplit : 1);
    if (p.CoutP % 64 == 0 && (tiles64 >= 2 * num_cu() || p.f.spade_x)) return launch_conv<KH, KW, S, 64, KC, XFORM>(p, s);
    return launch_conv<KH, KW, S, 32, KC, XFORM>(p, s);
}


// Input channels per LDS chunk of each geometry family (the instantiations below and the split-K planner share it).
constexpr int kc_for(int kh, int kw, int stride) {
    return stride == 1 ? (kh == 3 && kw == 3 ? 4 : kh == 1 && kw == 1 ? 16 : kh == 7 ? 2 : 8)      // 2x2, 2x1, 1x2: 8
                       : (kh == 3 ? 2 : 8);
}

// One entry per geometry family, each compiled in its own translation unit (conv2d_inst_*.hip).
int launch_k3s1(const ConvParams& p, hipStream_t s);
int launch_k1s1(const ConvParams& p, hipStream_t s);
int launch_k2x2(const ConvParams& p, hipStream_t s);
int launch_k2x1(const ConvParams& p, hipStream_t s);
int launch_k1x2(const ConvParams& p, hipStream_t s);
int launch_k7s1(const ConvParams& p, hipStream_t s);
int launch_k3s2(const ConvParams& p, hipStream_t s);
int launch_k1s2(const ConvParams& p, hipStream_t s);
int launch_wino(const ConvParams& p, hipStream_t s);     // conv2d_wino.h
int launch_wino4b(const ConvParams& p, hipStream_t s);   // conv2d_wino4b.h: F(4x4,3x3), two workgroups per CU (16x16x4 MFMA); PG_ERR_UNSUPPORTED when the launch is not its kind
int launch_wino4(const ConvParams& p, hipStream_t s);    // conv2d_wino4.h: F(4x4,3x3); PG_ERR_UNSUPPORTED when the launch is not its kind
int launch_wino4x3(const ConvParams& p, hipStream_t s);  // conv2d_wino4.h, X3 form: the transform-domain GEMM on the bf16 pipe (three-term operand splits); same acceptance as launch_wino4
int launch_s1x1(const ConvParams& p, hipStream_t s);     // conv2d_s1x1.h: streaming 1x1 (returns PG_ERR_UNSUPPORTED when the launch is not its kind)

}  // namespace pgconv
