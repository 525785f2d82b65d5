// The last output column (2W) and row (2H) of the fp32 `up = 2` layer, for the launches whose main tiles run on the bf16 pipe (conv2d_up2x3.h; round 6).
// conv2d_up2.h's own edge pass is a sliver of work behind a long serial chain: 256 positions x 32 couts per workgroup, all K chunks one after the other
// (8 ... 32 round trips to L2 / HBM at ~2.5 us each), the column tiles gathering x[.., W - 1] with a stride of one image row -- 97 ... 103 us per launch
// beside a 350 ... 480 us main kernel, 60 us of it the gather (tools/up2x3_variants.py, UX_EXP 64 / 128 / 256 / 512).  This kernel cuts the chain instead of
// the work:
//   * tile = 32 positions x 32 couts (one MFMA block) -> 272 ... 384 workgroups instead of 48 ... 256;
//   * the four waves of a workgroup SPLIT K: wave w multiplies the chunks w, w + 4, ... into its own accumulators from its own staging buffers (LDS-DMA,
//     two chunks deep, counted waits, no barrier inside the loop); the partial sums meet in LDS and are added in a fixed order (deterministic);
//   * the column tiles read input column W - 1 from a dense [N][Cin][H] copy the main kernel's producer waves leave behind (they hold the column in LDS anyway;
//     `xcol`, NULL: gathered from x at a stride of one image row = one 128-byte line per sample, +10 ... 34 us per launch);
//   * only the three taps an edge touches are staged (column: 2, 5, 8; row: 6, 7, 8): 6 KB of weights per chunk instead of 18.
// Arithmetic: v_mfma_f32_32x32x2_f32 as in conv2d_up2.h (the operands are float32; the edge is ~1 % of the layer's multiplies).
// Semantics (conv2d_up2.h's edge pass): column 2W, rows 2q | 2q + 1: w2 x[q][W-1] + w8 x[q-1][W-1] | w5 x[q][W-1]; row 2H, columns 2r | 2r + 1:
// w6 x[H-1][r] + w8 x[H-1][r-1] | w7 x[H-1][r]; the corner (2H, 2W) belongs to the column tiles.
#pragma once
#include "conv2d_up2.h"

namespace pgconv {

constexpr int UE_KC = 16, UE_POS = 32;
constexpr int UE_XS = UE_KC * (UE_POS + 1);                  // 528 samples per chunk: [channel][positions q0-1 .. q0+31]
constexpr int UE_XPT = (UE_XS + 63) / 64;                    // 9 four-byte requests per lane
constexpr int UE_XS_F = UE_XPT * 64;                         // floats reserved
constexpr int UE_WS4 = UE_KC * 3 * (U_BM / 4);               // 384 sixteen-byte words: [channel][tap 3][32 couts]
constexpr int UE_WPT = UE_WS4 / 64;                          // 6 per lane
constexpr int UE_BUF_F = UE_XS_F + UE_WS4 * 4;               // 2112 floats per staged chunk
#ifndef UE_DEPTH_DEF
#define UE_DEPTH_DEF 2
#endif
constexpr int UE_DEPTH = UE_DEPTH_DEF;
constexpr int UE_REQ = UE_XPT + UE_WPT;                      // vector-memory requests per lane and chunk
static_assert(UE_WS4 % 64 == 0 && 4 * UE_DEPTH * UE_BUF_F >= 4 * 2 * 16 * 64, "staging geometry");

template <bool MOD>
__global__ __launch_bounds__(256, 2) void conv2d_up2_edges(Up2Params p, int col_blocks, int row_blocks, const float* __restrict__ xcol) {
    extern __shared__ __attribute__((aligned(16))) float smem[];
    const int t = threadIdx.x, lane = t & 63;
    const int wave = __builtin_amdgcn_readfirstlane(t >> 6);
    const int half = lane >> 5, l31 = lane & 31;
    const int HW = p.H * p.W;
    const int nchunks = p.Cin / UE_KC;                       // (the launcher checks Cin % 16 == 0)
    float* stage = smem + wave * (UE_DEPTH * UE_BUF_F);      // this wave's own buffers
    float* cs = smem + 4 * UE_DEPTH * UE_BUF_F;              // input scales of this image [Cin]
    const unsigned stage_b = __builtin_amdgcn_readfirstlane(lds_offset(stage));

    int L = blockIdx.x;
    const int per_n = p.mblocks * (col_blocks + row_blocks);
    const int n = L / per_n;
    L -= n * per_n;
    const int mb = L % p.mblocks;
    L /= p.mblocks;
    const bool row_tile = L >= col_blocks;
    const int q0 = (row_tile ? L - col_blocks : L) * UE_POS; // (row tiles: the first COLUMN position)
    const int m0 = mb * U_BM;
    const int lim = row_tile ? p.W : p.H;

    if (MOD)
        for (int c = t; c < p.Cin; c += 256) cs[c] = p.in_scale[(int64_t)n * p.Cin + c];

    unsigned xoff[UE_XPT];
#pragma unroll
    for (int i = 0; i < UE_XPT; i++) {
        const int e = i * 64 + lane;
        const int c = e / (UE_POS + 1), g = q0 - 1 + e % (UE_POS + 1);
        const bool ok = e < UE_XS && g >= 0 && g < lim;
        // column tiles: the dense column [N][Cin][H] the main kernel left behind (conv2d_up2x3.h), or x[.., W - 1] itself at a stride of one image row
        xoff[i] = !ok ? 0x80000000u : (unsigned)(row_tile ? c * HW + (p.H - 1) * p.W + g : (xcol ? c * p.H + g : c * HW + g * p.W + p.W - 1)) * 4u;
    }
    int woff[UE_WPT];
#pragma unroll
    for (int i = 0; i < UE_WPT; i++) {
        const int e4 = i * 64 + lane;
        const int row = e4 >> 3, col4 = e4 & 7;
        const int c = row / 3, ts = row - 3 * c;
        const int tap = ts == 2 ? 8 : (row_tile ? 6 + ts : 2 + 3 * ts);
        woff[i] = (c * 9 + tap) * p.CoutP + m0 + 4 * col4;
    }
    const bool dense = !row_tile && xcol != nullptr;
    const int cstride = dense ? p.H : HW;                    // floats between two channels of the staged source
    i32x4 xrsrc;
    const uint64_t base = (uint64_t)(uintptr_t)(dense ? xcol + (int64_t)n * p.Cin * p.H : p.x + (int64_t)n * p.Cin * HW);
    xrsrc[0] = __builtin_amdgcn_readfirstlane((int)(unsigned)base);
    xrsrc[1] = __builtin_amdgcn_readfirstlane((int)(unsigned)(base >> 32) & 0xffff);
    xrsrc[2] = p.Cin * cstride * 4;
    xrsrc[3] = 0x00020000;
    __syncthreads();                                         // the scales are in LDS

    auto issue = [&](int chunk, int buf) __attribute__((always_inline)) {
        const unsigned xb = stage_b + (unsigned)(buf * UE_BUF_F) * 4u;
        const int soff = chunk * UE_KC * cstride * 4;
#pragma unroll
        for (int i = 0; i < UE_XPT; i++) dma_dword(xrsrc, xb + 256u * i, xoff[i], soff);
        const float* wb = p.wp + (int64_t)chunk * UE_KC * 9 * p.CoutP;
#pragma unroll
        for (int i = 0; i < UE_WPT; i++) dma_dwordx4(wb + woff[i], xb + (unsigned)UE_XS_F * 4u + 1024u * i);
    };

    f32x16 ea[2];                                            // [parity of the output row (row tiles: column)]
#pragma unroll
    for (int a = 0; a < 2; a++)
#pragma unroll
        for (int k = 0; k < 16; k++) ea[a][k] = 0.f;
    const int my = (nchunks - wave + 3) >> 2;                // chunks wave, wave + 4, ...
#pragma unroll
    for (int d = 0; d < UE_DEPTH - 1; d++)
        if (d < my) issue(wave + 4 * d, d);
    for (int j = 0; j < my; j++) {
        const int chunk = wave + 4 * j, buf = j % UE_DEPTH;
        const int ahead = my - 1 - j;                        // chunks requested after this one once the next request is out
        if (j + UE_DEPTH - 1 < my) {
            issue(chunk + 4 * (UE_DEPTH - 1), (j + UE_DEPTH - 1) % UE_DEPTH);
            asm volatile("s_waitcnt vmcnt(%0)" :: "n"(UE_REQ * (UE_DEPTH - 1)) : "memory");     // in-order return: this chunk is home, the younger ones stay in flight
        } else if (UE_DEPTH > 2 && ahead == 1) {
            asm volatile("s_waitcnt vmcnt(%0)" :: "n"(UE_REQ) : "memory");
        } else if (UE_DEPTH > 3 && ahead == 2) {
            asm volatile("s_waitcnt vmcnt(%0)" :: "n"(UE_REQ * 2) : "memory");
        } else {
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        }
        const float* xs = stage + buf * UE_BUF_F;
        const float* ws = xs + UE_XS_F;
#pragma unroll
        for (int cp = 0; cp < UE_KC / 2; cp++) {
            const int ch = 2 * cp + half;
            const float a_lo = ws[(ch * 3 + 0) * U_BM + l31], a_mid = ws[(ch * 3 + 1) * U_BM + l31], a8 = ws[(ch * 3 + 2) * U_BM + l31];
            const float sc = MOD ? cs[chunk * UE_KC + ch] : 1.f;
            const float xm = MOD ? xs[ch * (UE_POS + 1) + l31] * sc : xs[ch * (UE_POS + 1) + l31];
            const float x0 = MOD ? xs[ch * (UE_POS + 1) + l31 + 1] * sc : xs[ch * (UE_POS + 1) + l31 + 1];
            ea[0] = __builtin_amdgcn_mfma_f32_32x32x2f32(a_lo, x0, ea[0], 0, 0, 0);
            ea[0] = __builtin_amdgcn_mfma_f32_32x32x2f32(a8, xm, ea[0], 0, 0, 0);
            ea[1] = __builtin_amdgcn_mfma_f32_32x32x2f32(a_mid, x0, ea[1], 0, 0, 0);
        }
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");   // this buffer has been read: the request after next may overwrite it
    }

    // ---- the four partial sums meet in LDS (the staging buffers are done), fixed order ((w0 + w1) + w2) + w3
    __syncthreads();
    float* red = smem;                                       // [wave][a][k][lane]
#pragma unroll
    for (int a = 0; a < 2; a++)
#pragma unroll
        for (int k = 0; k < 16; k++) red[((wave * 2 + a) * 16 + k) * 64 + lane] = ea[a][k];
    __syncthreads();
    const int q = q0 + l31;
#pragma unroll
    for (int a = 0; a < 2; a++) {
        const bool pos_ok = row_tile ? q < p.W : (a == 0 ? q <= p.H : q < p.H);
#pragma unroll
        for (int kk = 0; kk < 4; kk++) {
            const int k = 4 * wave + kk;                     // this wave finishes registers 4 wave .. 4 wave + 3 of both parities
            const float v = ((red[((0 * 2 + a) * 16 + k) * 64 + lane] + red[((1 * 2 + a) * 16 + k) * 64 + lane]) + red[((2 * 2 + a) * 16 + k) * 64 + lane]) +
                            red[((3 * 2 + a) * 16 + k) * 64 + lane];
            const int rowc = (k & 3) + 8 * (k >> 2) + 4 * half;      // D row = cout, D column = lane & 31 = position
            const int co = m0 + rowc;
            if (pos_ok && co < p.Cout) {
                const float sc = p.out_scale ? p.out_scale[(int64_t)n * p.Cout + co] : 1.f;
                p.y[(int64_t)n * p.ys[0] + (int64_t)co * p.ys[1] + (int64_t)(row_tile ? 2 * p.H : 2 * q + a) * p.ys[2] + (int64_t)(row_tile ? 2 * q + a : 2 * p.W) * p.ys[3]] = v * sc;
            }
        }
    }
}

inline size_t up2_edges_lds_bytes(int Cin) { return ((size_t)4 * UE_DEPTH * UE_BUF_F + Cin) * sizeof(float); }

// `row_edge`: the main tiles stop at position row H - 1 (conv2d_up2.h: H a multiple of the main tile's rows) and output row 2H is made here
inline int launch_up2_edges(const Up2Params& p0, bool row_edge, const float* xcol, hipStream_t s) {
    if (p0.Cin % UE_KC != 0 || p0.ksplit != 1) return PG_ERR_UNSUPPORTED;
    Up2Params p = p0;
    p.mblocks = p.CoutP / U_BM;
    const int col_blocks = (p.H + 1 + UE_POS - 1) / UE_POS, row_blocks = row_edge ? (p.W + UE_POS - 1) / UE_POS : 0;
    const int64_t tiles = (int64_t)p.N * p.mblocks * (col_blocks + row_blocks);
    if (tiles > 0x7fffffffLL) return PG_ERR_TOO_LARGE;
    const size_t lds = up2_edges_lds_bytes(p.Cin);
    if (lds > 160 * 1024) return PG_ERR_UNSUPPORTED;
    if (p.in_scale) {
        static PerDeviceOnce a1;
        const hipError_t e = a1.run([] { return hipFuncSetAttribute((const void*)conv2d_up2_edges<true>, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024); });
        if (e != hipSuccess) return (int)e;
        hipLaunchKernelGGL((conv2d_up2_edges<true>), dim3((unsigned)tiles), dim3(256), lds, s, p, col_blocks, row_blocks, xcol);
    } else {
        static PerDeviceOnce a0;
        const hipError_t e = a0.run([] { return hipFuncSetAttribute((const void*)conv2d_up2_edges<false>, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024); });
        if (e != hipSuccess) return (int)e;
        hipLaunchKernelGGL((conv2d_up2_edges<false>), dim3((unsigned)tiles), dim3(256), lds, s, p, col_blocks, row_blocks, xcol);
    }
    return launch_status();
}

}  // namespace pgconv
