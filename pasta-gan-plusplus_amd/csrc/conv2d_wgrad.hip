// Weight gradient of a stride-1 (3x3, 1x1) or stride-2 (3x3) fp32 NCHW convolution for gfx950 as a GEMM over pixels on v_mfma_f32_32x32x2_f32.
//
// What it replaces in the reference: `aten::cudnn_convolution_backward_weight` behind conv2d_gradfix
// (torch_utils/ops/conv2d_gradfix.py:137-150) for the geometries that carry the training step's FLOPs (3x3 and 1x1,
// stride 1: the SPADE blocks, the style-branch conv1 layers, ToRGB, the discriminator's conv0):
//     dw[co, ci, ky, kx] = sum_{n, oy, ox} dy[n, co, oy, ox] * x[n, ci, S oy + ky - pad_y, S ox + kx - pad_x]
//
// GEMM view:  M = Cout (A = dy: for a fixed cout the pixels are contiguous -> K-contiguous),
//             N = Cin, once per tap (B = x shifted by the tap: K-contiguous as well),
//             K = N * OH * OW pixels, two adjacent pixels per MFMA (lanes 0-31 hold pixel 2j, lanes 32-63 pixel 2j+1).
// Workgroup = 4 waves = 64 couts x 64 cins x all taps; a wave owns one 32 x 32 block for every tap (KH*KW accumulators).
// The K axis is what gives parallelism: workgroup s of `splits` walks pixel chunks s, s + splits, ... (2 rows x 32 columns of one
// image each), stages dy[64][64] and the x halo [64][2 + KH - 1][32 + KW - 1] in LDS by LDS-DMA (odd pitches: both operand reads
// are conflict-free; two buffers, the next chunk in flight behind the MFMAs) and accumulates; its partial block goes to
// workspace[s][tap][co][ci] and a second pass adds the partials in fixed order (deterministic, no atomics).
//
// Roofline: MFMA.  Algorithmic FLOPs = 2 * N * OH * OW * Cout * Cin * KH * KW (the forward pass's count) against 157.3 TFLOP/s.
#include "conv2d_kernel.h"        // LDS-DMA helpers (dma_dword, lds_offset)

namespace {

using namespace pg;
typedef float f32x16 __attribute__((ext_vector_type(16)));

constexpr int WG_TW = 32;                      // pixel chunk: R rows x WG_TW columns of dy (R = 2 at stride 1, 1 at stride 2: the x footprint must fit LDS twice)
constexpr int WG_BM = 64, WG_BN = 64;

template <int KH, int KW, int S>
struct WGeo {
    static constexpr int T = KH * KW;
    static constexpr int R = S == 1 ? 2 : 1, PIX = WG_TW * R;
    static constexpr int IH = (R - 1) * S + KH, IW = (WG_TW - 1) * S + KW;
    static constexpr int PA = PIX + 1;                                  // dy row pitch (odd)
    static constexpr int PB = (IH * IW) | 1;                            // x plane pitch (odd)
    static constexpr int LDS_FLOATS = 2 * ((WG_BM * PA + 255) / 256 + (WG_BN * PB + 255) / 256) * 256;   // two staging buffers of whole 256-lane DMA rows
};

struct WgradParams {
    const float* x; const float* dy; float* ws;
    int N, Cin, H, W, Cout, OH, OW, pad_y, pad_x;
    int tilesX, tilesY, chunks, splits, coB, ciB;
};

constexpr unsigned WG_SENTINEL = 0x80000000u;     // byte offset beyond any descriptor range: the DMA writes 0.0f

// Staging: both tiles travel global -> LDS by 4-byte LDS-DMA (`buffer_load_dword ... lds`, conv2d_kernel.h), lane-linear, so the
// LDS image of a tile is its flat element order and the gather sits in the per-lane source offset.  The offsets relative to the
// chunk origin are kernel-lifetime registers; a chunk whose footprint lies inside the image just adds its origin (in the
// descriptor base) -- no per-element arithmetic; border chunks mask their out-of-image elements to a sentinel offset.  Two
// staging buffers: chunk g + 1 is in flight while chunk g is multiplied, one barrier per chunk.  One workgroup of EIGHT waves per
// CU: waves 0-3 multiply (9 x 16 accumulators each), waves 4-7 only request the next chunk (52 requests + M0 traffic per lane and
// chunk) -- issued by the multiplying waves themselves those requests cost 20 % of the kernel (measured: 88 -> 110 TFLOP/s with
// the requests removed), from a wave of their own they overlap the other wave's MFMAs on the same SIMD.
// TS (round 4): multiplying wave groups.  TS = 2: waves 0-3 own the first ceil(T / 2) taps, waves 4-7 the rest (the dy operand is read by both), the loader
// waves follow -- TWO multiplying waves per SIMD that cover each other's LDS round trips and barrier waits (a single one has nothing but its own MFMAs).
template <int KH, int KW, int S, int TS = 1>
__global__ __launch_bounds__(256 * TS + 256, 1) void conv2d_wgrad(WgradParams p) {
    typedef WGeo<KH, KW, S> G;
    constexpr int TG = (G::T + TS - 1) / TS;           // taps per multiplying group (the last group may own fewer)
    constexpr int WG_R = G::R, WG_PIX = G::PIX;
    constexpr int NDY = (WG_BM * G::PA + 255) / 256, NX = (WG_BN * G::PB + 255) / 256;
    extern __shared__ float smem[];
    constexpr int BUF = (NDY + NX) * 256;       // floats per staging buffer: dy [64 co][PA] then x [64 ci][PB], whole DMA rows
    const int lane = threadIdx.x & 63;
    const int wave8 = __builtin_amdgcn_readfirstlane((int)threadIdx.x >> 6);
    const bool loader = wave8 >= 4 * TS;             // wave-uniform role
    const int wave = wave8 & 3, t = threadIdx.x & 255;
    const int tg = TS == 1 ? 0 : (wave8 >> 2);       // tap group of a multiplying wave
    const int tap0 = tg * TG, ntap = (tap0 + TG <= G::T ? TG : G::T - tap0);
    const int half = lane >> 5, l31 = lane & 31;
    const int mt = wave & 1, nt = wave >> 1;
    int b = blockIdx.x;
    const int s = b % p.splits; b /= p.splits;
    const int cib = b % p.ciB, cob = b / p.ciB;
    const int co0 = cob * WG_BM, ci0 = cib * WG_BN;
    const int OHW = p.OH * p.OW, HW = p.H * p.W;
    const unsigned smem_b = __builtin_amdgcn_readfirstlane(pgconv::lds_offset(smem));

    if (loader) {
        // ---- loader waves: gather maps, then request chunk after chunk, one barrier per chunk with the multiplying waves
        // per-thread gather maps (bytes from the chunk origin); statically invalid elements (pitch padding, channels beyond the tensor) = sentinel
        unsigned rel_dy[NDY], rel_x[NX];
    #pragma unroll
        for (int i = 0; i < NDY; i++) {
            const int f = t + 256 * i, co = f / G::PA, px = f % G::PA;
            const bool ok = co < WG_BM && px < WG_PIX && co0 + co < p.Cout;
            rel_dy[i] = ok ? (unsigned)(co * OHW + (px / WG_TW) * p.OW + px % WG_TW) * 4u : WG_SENTINEL;
        }
    #pragma unroll
        for (int i = 0; i < NX; i++) {
            const int f = t + 256 * i, ci = f / G::PB, rr = f % G::PB;
            const bool ok = ci < WG_BN && rr < G::IH * G::IW && ci0 + ci < p.Cin;
            rel_x[i] = ok ? (unsigned)(ci * HW + (rr / G::IW) * p.W + rr % G::IW) * 4u : WG_SENTINEL;
        }
        auto issue = [&](int ch, int buf) __attribute__((always_inline)) {
            int c = ch;
            const int tx = c % p.tilesX; c /= p.tilesX;
            const int ty = c % p.tilesY;
            const int n = c / p.tilesY;
            const int oy0 = ty * WG_R, ox0 = tx * WG_TW;
            const int iy0 = oy0 * S - p.pad_y, ix0 = ox0 * S - p.pad_x;
            // descriptors based at the chunk origin (the x origin may lie before the tensor: such elements are masked below)
            const uint64_t dyb = (uint64_t)(uintptr_t)(p.dy + ((int64_t)n * p.Cout + co0) * OHW + (int64_t)oy0 * p.OW + ox0);
            const uint64_t xb = (uint64_t)(uintptr_t)(p.x + ((int64_t)n * p.Cin + ci0) * HW) + ((int64_t)iy0 * p.W + ix0) * 4;
            pgconv::i32x4 rdy, rx;
            rdy[0] = __builtin_amdgcn_readfirstlane((int)(unsigned)dyb); rdy[1] = __builtin_amdgcn_readfirstlane((int)(unsigned)(dyb >> 32) & 0xffff);
            rdy[2] = 0x7ffffffe; rdy[3] = 0x00020000;
            rx[0] = __builtin_amdgcn_readfirstlane((int)(unsigned)xb); rx[1] = __builtin_amdgcn_readfirstlane((int)(unsigned)(xb >> 32) & 0xffff);
            rx[2] = 0x7ffffffe; rx[3] = 0x00020000;
            const bool inner_dy = oy0 + WG_R <= p.OH && ox0 + WG_TW <= p.OW;                                       // wave-uniform
            const bool inner_x = iy0 >= 0 && iy0 + G::IH <= p.H && ix0 >= 0 && ix0 + G::IW <= p.W;
            const unsigned base_b = smem_b + (unsigned)(buf * BUF + 64 * wave) * 4u;
    #pragma unroll
            for (int i = 0; i < NDY; i++) {
                unsigned v = rel_dy[i];
                if (!inner_dy) {
                    const int px = (t + 256 * i) % G::PA;
                    if (oy0 + px / WG_TW >= p.OH || ox0 + px % WG_TW >= p.OW) v = WG_SENTINEL;
                }
                pgconv::dma_dword(rdy, base_b + (unsigned)(256 * i) * 4u, v, 0);
            }
    #pragma unroll
            for (int i = 0; i < NX; i++) {
                unsigned v = rel_x[i];
                if (!inner_x) {
                    const int rr = (t + 256 * i) % G::PB;
                    const int iy = iy0 + rr / G::IW, ix = ix0 + rr % G::IW;
                    if (iy < 0 || iy >= p.H || ix < 0 || ix >= p.W) v = WG_SENTINEL;
                }
                pgconv::dma_dword(rx, base_b + (unsigned)(NDY * 256 + 256 * i) * 4u, v, 0);
            }
        };
        int ch = s, g = 0;
        if (ch < p.chunks) issue(ch, 0);
        pgconv::dma_wait_all();
        __syncthreads();
        for (; ch < p.chunks; ch += p.splits, g++) {
            if (ch + p.splits < p.chunks) issue(ch + p.splits, (g & 1) ^ 1);      // that buffer was last read one iteration ago (barrier below)
            pgconv::dma_wait_all();                                              // the chunk requested above has landed
            __syncthreads();
        }
        return;
    }

    f32x16 acc[TG];
#pragma unroll
    for (int tp = 0; tp < TG; tp++)
#pragma unroll
        for (int k = 0; k < 16; k++) acc[tp][k] = 0.f;

    int ch = s, g = 0;
    __syncthreads();                             // chunk 0 has landed
    for (; ch < p.chunks; ch += p.splits, g++) {
        const int buf = g & 1;
        const float* dyt = smem + buf * BUF;
        const float* xt = dyt + NDY * 256;
        const float* a_base = dyt + (mt * 32 + l31) * G::PA + half;
        const float* b_base = xt + (nt * 32 + l31) * G::PB + half * S;
        // One wave per SIMD: nothing hides an LDS round trip but the wave's own MFMAs, so the operands of step kk + 1 are
        // requested before the MFMAs of step kk are issued (register double buffering, order pinned by sched_barrier).
        // (TS = 2: the tap offsets of this wave's group, wave-uniform, folded into the base; a group with one tap less repeats its last one into a spare slot)
        auto tap_off = [&](int j) { const int tp = tap0 + (j < ntap ? j : ntap - 1); return (tp / KW) * G::IW + tp % KW; };
        auto fetch = [&](int kk, float& a, float (&bv)[TG]) __attribute__((always_inline)) {
            const int r = (2 * kk) / WG_TW, cc = (2 * kk) % WG_TW;
            a = a_base[2 * kk];
            if (TS == 1) {
#pragma unroll
                for (int ky = 0; ky < KH; ky++)
#pragma unroll
                    for (int kx = 0; kx < KW; kx++) bv[ky * KW + kx] = b_base[(r * S + ky) * G::IW + cc * S + kx];
            } else {
#pragma unroll
                for (int j = 0; j < TG; j++) bv[j] = b_base[(r * S) * G::IW + cc * S + tap_off(j)];
            }
        };
        // two operand register sets used alternately (steps kk, kk + 1 per trip): no register moves between steps (round 4: the one-set form spent
        // 10 v_mov per 9 MFMAs, and on this part vector instructions and the f32 MFMA share the SIMD's issue time)
        float a0, b0[TG], a1, b1[TG];
        static_assert((WG_PIX / 2) % 2 == 0, "an even number of steps per chunk");
        fetch(0, a0, b0);
        for (int kk = 0; kk < WG_PIX / 2; kk += 2) {                    // two adjacent pixels per step
            fetch(kk + 1, a1, b1);
            __builtin_amdgcn_sched_barrier(0);
#pragma unroll
            for (int tp = 0; tp < TG; tp++)
                if (TS == 1 || tp < ntap) acc[tp] = __builtin_amdgcn_mfma_f32_32x32x2f32(a0, b0[tp], acc[tp], 0, 0, 0);
            __builtin_amdgcn_sched_barrier(0);
            if (kk + 2 < WG_PIX / 2) fetch(kk + 2, a0, b0);
            __builtin_amdgcn_sched_barrier(0);
#pragma unroll
            for (int tp = 0; tp < TG; tp++)
                if (TS == 1 || tp < ntap) acc[tp] = __builtin_amdgcn_mfma_f32_32x32x2f32(a1, b1[tp], acc[tp], 0, 0, 0);
            __builtin_amdgcn_sched_barrier(0);
        }
        __syncthreads();
    }
    // partial block -> workspace[s][tap][co][ci]: D col = lane & 31 = ci (contiguous), row = (reg & 3) + 8 * (reg >> 2) + 4 * half = co
    float* wsp = p.ws + (int64_t)s * G::T * p.Cout * p.Cin;
    const int ci = ci0 + nt * 32 + l31;
#pragma unroll
    for (int tp = 0; tp < TG; tp++)
#pragma unroll
        for (int k = 0; k < 16; k++) {
            const int co = co0 + mt * 32 + (k & 3) + 8 * (k >> 2) + 4 * half;
            if ((TS == 1 || tp < ntap) && co < p.Cout && ci < p.Cin) wsp[((int64_t)(tap0 + tp) * p.Cout + co) * p.Cin + ci] = acc[tp][k];
        }
}

// dw[co][ci][tap] = sum_s ws[s][tap][co][ci].  256 threads = 64 consecutive elements x 4 split groups: group g adds the partials s = g, g + 4, ... in
// ascending order (four independent load streams per element instead of one), the four group sums are folded in group order through LDS --
// a fixed association, the same on every run.
__global__ __launch_bounds__(256) void wgrad_reduce(const float* __restrict__ ws, float* __restrict__ dw, int splits, int T, int Cout, int Cin) {
    __shared__ float part[4][64];
    const int64_t total = (int64_t)T * Cout * Cin;
    const int e = threadIdx.x & 63, g = threadIdx.x >> 6;
    for (int64_t i0 = (int64_t)blockIdx.x * 64; i0 < total; i0 += (int64_t)gridDim.x * 64) {
        const int64_t i = i0 + e;
        float v = 0.f;
        if (i < total) {
            int z = g;
            for (; z + 12 < splits; z += 16) {          // four loads in flight per thread
                const float a = ws[(int64_t)z * total + i], b = ws[(int64_t)(z + 4) * total + i];
                const float c = ws[(int64_t)(z + 8) * total + i], d = ws[(int64_t)(z + 12) * total + i];
                v += a; v += b; v += c; v += d;
            }
            for (; z < splits; z += 4) v += ws[(int64_t)z * total + i];
        }
        part[g][e] = v;
        __syncthreads();
        if (g == 0 && i < total) {
            const float r = ((part[0][e] + part[1][e]) + part[2][e]) + part[3][e];
            const int ci = (int)(i % Cin), co = (int)((i / Cin) % Cout), tp = (int)(i / ((int64_t)Cin * Cout));
            dw[((int64_t)co * Cin + ci) * T + tp] = r;
        }
        __syncthreads();
    }
}

inline int cdiv(int a, int b) { return (a + b - 1) / b; }

// The kernel's gather maps are 32-bit BYTE offsets of a 64-channel block, (ci * plane + y * W + x) * 4, with 0x80000000 as the
// "write zero" sentinel: a plane must keep every valid offset (plus a halo row's slack) below 2^31.
inline bool offsets_fit(int64_t plane) { return (64 * plane + 4096) * 4 < 0x7fffffffLL; }


// =====================================================================================================================
// fp32 weight gradient of a FEW-CHANNEL large-kernel layer (the encoders' 7x7 stem on a 3-channel image, networks.py:2242): the (ci, ky, kx)
// triples are the GEMM's N axis -- Cin * KH * KW <= 160 = five 32-wide blocks -- instead of one N block of 64 input channels per tap (which would
// spend 61 of 64 columns on padding).  M = 64 couts, K = pixels as above.  Ten multiplying waves (2 cout halves x 5 N blocks, one accumulator each)
// + two loader waves; the B operand of lane n is x[ci(n)][row + ky(n)][col + kx(n)]: a per-lane constant offset into the staged halo.
// Partials go to workspace[s][co][n], n = ci * KH*KW + tap -- dw's own [Cout][Cin][KH][KW] order -- and through wgrad_reduce with T = 1.
template <int KH, int KW>
struct WSmallGeo {
    static constexpr int T = KH * KW, NB = 5;
    static constexpr int R = 2, PIX = WG_TW * R;
    static constexpr int IH = R - 1 + KH, IW = WG_TW - 1 + KW;
    static constexpr int CMAX = (32 * NB) / T;                          // input channels the five N blocks hold (3 for 7x7)
    static constexpr int PA = PIX + 1, PB = (IH * IW) | 1;
    static constexpr int LOADERS = 128;
    static constexpr int NDY = (WG_BM * PA + LOADERS - 1) / LOADERS, NX = (CMAX * PB + LOADERS - 1) / LOADERS;
    static constexpr int BUF = (NDY + NX) * LOADERS;                    // floats per staging buffer
};

template <int KH, int KW>
__global__ __launch_bounds__(768, 1) void conv2d_wgrad_fewcin(WgradParams p) {
    typedef WSmallGeo<KH, KW> G;
    extern __shared__ float smem[];
    const int lane = threadIdx.x & 63;
    const int wave12 = __builtin_amdgcn_readfirstlane((int)threadIdx.x >> 6);
    const bool loader = wave12 >= 10;
    const int half = lane >> 5, l31 = lane & 31;
    int b = blockIdx.x;
    const int s = b % p.splits;
    const int co0 = (b / p.splits) * WG_BM;
    const int OHW = p.OH * p.OW, HW = p.H * p.W;
    const unsigned smem_b = __builtin_amdgcn_readfirstlane(pgconv::lds_offset(smem));

    if (loader) {
        const int t = threadIdx.x - 640, lw = wave12 - 10;
        unsigned rel_dy[G::NDY], rel_x[G::NX];
#pragma unroll
        for (int i = 0; i < G::NDY; i++) {
            const int f = t + G::LOADERS * i, co = f / G::PA, px = f % G::PA;
            const bool ok = co < WG_BM && px < G::PIX && co0 + co < p.Cout;
            rel_dy[i] = ok ? (unsigned)(co * OHW + (px / WG_TW) * p.OW + px % WG_TW) * 4u : WG_SENTINEL;
        }
#pragma unroll
        for (int i = 0; i < G::NX; i++) {
            const int f = t + G::LOADERS * i, ci = f / G::PB, rr = f % G::PB;
            const bool ok = ci < G::CMAX && rr < G::IH * G::IW && ci < p.Cin;
            rel_x[i] = ok ? (unsigned)(ci * HW + (rr / G::IW) * p.W + rr % G::IW) * 4u : WG_SENTINEL;
        }
        auto issue = [&](int chk, int buf) __attribute__((always_inline)) {
            int c = chk;
            const int tx = c % p.tilesX; c /= p.tilesX;
            const int ty = c % p.tilesY;
            const int n = c / p.tilesY;
            const int oy0 = ty * G::R, ox0 = tx * WG_TW;
            const int iy0 = oy0 - p.pad_y, ix0 = ox0 - p.pad_x;
            const uint64_t dyb = (uint64_t)(uintptr_t)(p.dy + ((int64_t)n * p.Cout + co0) * OHW + (int64_t)oy0 * p.OW + ox0);
            const uint64_t xb = (uint64_t)(uintptr_t)(p.x + (int64_t)n * p.Cin * HW) + ((int64_t)iy0 * p.W + ix0) * 4;
            pgconv::i32x4 rdy, rx;
            rdy[0] = __builtin_amdgcn_readfirstlane((int)(unsigned)dyb); rdy[1] = __builtin_amdgcn_readfirstlane((int)(unsigned)(dyb >> 32) & 0xffff);
            rdy[2] = 0x7ffffffe; rdy[3] = 0x00020000;
            rx[0] = __builtin_amdgcn_readfirstlane((int)(unsigned)xb); rx[1] = __builtin_amdgcn_readfirstlane((int)(unsigned)(xb >> 32) & 0xffff);
            rx[2] = 0x7ffffffe; rx[3] = 0x00020000;
            const bool inner_dy = oy0 + G::R <= p.OH && ox0 + WG_TW <= p.OW;
            const bool inner_x = iy0 >= 0 && iy0 + G::IH <= p.H && ix0 >= 0 && ix0 + G::IW <= p.W;
            const unsigned base_b = smem_b + (unsigned)(buf * G::BUF + 64 * lw) * 4u;
#pragma unroll
            for (int i = 0; i < G::NDY; i++) {
                unsigned v = rel_dy[i];
                if (!inner_dy) {
                    const int px = (t + G::LOADERS * i) % G::PA;
                    if (oy0 + px / WG_TW >= p.OH || ox0 + px % WG_TW >= p.OW) v = WG_SENTINEL;
                }
                pgconv::dma_dword(rdy, base_b + (unsigned)(G::LOADERS * i) * 4u, v, 0);
            }
#pragma unroll
            for (int i = 0; i < G::NX; i++) {
                unsigned v = rel_x[i];
                if (!inner_x) {
                    const int rr = (t + G::LOADERS * i) % G::PB;
                    const int iy = iy0 + rr / G::IW, ix = ix0 + rr % G::IW;
                    if (iy < 0 || iy >= p.H || ix < 0 || ix >= p.W) v = WG_SENTINEL;
                }
                pgconv::dma_dword(rx, base_b + (unsigned)((G::NDY + i) * G::LOADERS) * 4u, v, 0);
            }
        };
        int ch = s, g = 0;
        if (ch < p.chunks) issue(ch, 0);
        pgconv::dma_wait_all();
        __syncthreads();
        for (; ch < p.chunks; ch += p.splits, g++) {
            if (ch + p.splits < p.chunks) issue(ch + p.splits, (g & 1) ^ 1);
            pgconv::dma_wait_all();
            __syncthreads();
        }
        return;
    }

    const int mt = wave12 & 1, nt = wave12 >> 1;
    const int nidx = nt * 32 + l31;                                    // (ci, ky, kx) of this lane's column
    const bool ncol_ok = nidx < p.Cin * G::T;
    const int nn = ncol_ok ? nidx : 0;
    const int b_off = (nn / G::T) * G::PB + ((nn % G::T) / KW) * G::IW + (nn % G::T) % KW + half;
    f32x16 acc;
#pragma unroll
    for (int k = 0; k < 16; k++) acc[k] = 0.f;
    int ch = s, g = 0;
    __syncthreads();
    for (; ch < p.chunks; ch += p.splits, g++) {
        const float* dyt = smem + (g & 1) * G::BUF;
        const float* xt = dyt + G::NDY * G::LOADERS;
        const float* a_base = dyt + (mt * 32 + l31) * G::PA + half;
        const float* b_base = xt + b_off;
#pragma unroll 8
        for (int kk = 0; kk < G::PIX / 2; kk++) {
            const int r = (2 * kk) / WG_TW, cc = (2 * kk) % WG_TW;
            acc = __builtin_amdgcn_mfma_f32_32x32x2f32(a_base[2 * kk], b_base[r * G::IW + cc], acc, 0, 0, 0);
        }
        __syncthreads();
    }
    float* wsp = p.ws + (int64_t)s * p.Cout * p.Cin * G::T;
    if (ncol_ok) {
#pragma unroll
        for (int k = 0; k < 16; k++) {
            const int co = co0 + mt * 32 + (k & 3) + 8 * (k >> 2) + 4 * half;
            if (co < p.Cout) wsp[(int64_t)co * p.Cin * G::T + nidx] = acc[k];
        }
    }
}

inline bool fewcin_covers(int Cin, int KH, int KW, int stride) { return KH == 7 && KW == 7 && stride == 1 && Cin * 49 <= 160; }

// =====================================================================================================================
// 16-bit (fp16 / bf16) weight gradient, channels-last operands: x [N, H, W, Cin], dy [N, OH, OW, Cout] -> dw fp32 [Cout, Cin, KH, KW].
// What it replaces: aten::convolution_backward (MIOpen's igemm_wrw_*_fp16) behind the discriminator's half-precision blocks
// (reference conv2d_gradfix.py:137-150 -> cudnn_convolution_backward_weight).
//
// Same GEMM as above (M = Cout, N = Cin per tap, K = pixels) on v_mfma_f32_32x32x16_{f16,bf16}: an operand fragment is 32 channels x 16
// pixels with EIGHT CONSECUTIVE PIXELS per lane -- but channels-last memory has the channels contiguous.  The tiles are staged as they lie
// in memory ([pixel][64 channels], 16-byte LDS-DMA, coalesced) and read with gfx950's transposing LDS read `ds_read_b64_tr_b16`: per
// 16-lane group a block of 4 pixel rows x 16 channels arrives channel-major, four consecutive pixels of one channel per lane; two such
// reads make one operand.  A k-step = 16 consecutive output pixels of one output row; a chunk = R rows x 32 columns.
//   stride 1: 128-byte pixel rows, 16-byte slot s of pixel p holds channel slot s ^ (4 * ((p >> 1) & 1)): the four rows of a read sit on
//             banks {0,32,16,48} + [0,16) for ANY four consecutive pixels -> conflict-free; the x tile's rows are 36 pixels wide so that
//             the swizzle bit of a lane's pixel depends on the tap's kx only (three address registers, everything else immediates)
//   stride 2: a read's four pixels are two apart: 160-byte pixel rows (10 slots, 2 of padding), no swizzle -> conflict-free as well
// Workgroup = 4 multiplying waves (2 x 2 blocks of 32 couts x 32 cins, all taps: 9 x 16 accumulators) + 4 loader waves (gather maps in
// registers, one barrier per chunk, two staging buffers), split over K like the fp32 kernel, partials reduced in fixed order by wgrad_reduce.
typedef short s16x4 __attribute__((ext_vector_type(4)));
typedef short s16x8 __attribute__((ext_vector_type(8)));
typedef __bf16 wbf16x8 __attribute__((ext_vector_type(8)));
typedef _Float16 wf16x8 __attribute__((ext_vector_type(8)));

template <int KH, int KW, int S>
struct W16Geo {
    static constexpr int T = KH * KW;
    static constexpr int R = S == 1 ? 4 : 2, TW = 32, PIX = R * TW, KSTEPS = PIX / 16;
    static constexpr int IH = (R - 1) * S + KH;
    static constexpr int IW = S == 1 ? ((TW - 1 + KW) + 3) / 4 * 4 : (TW - 1) * S + KW;      // stride 1: a multiple of 4 pixels (see above)
    static constexpr int XSLOTS = S == 1 ? 8 : 10;                                           // 16-byte slots per x pixel row
    static constexpr int XROWB = XSLOTS * 16;
    static constexpr int NDY = (PIX * 8 + 255) / 256, NX = (IH * IW * XSLOTS + 255) / 256;   // 16-byte DMA requests per loader thread
    static constexpr int BUF = (NDY + NX) * 256 * 16;                                        // bytes per staging buffer
};

struct Wgrad16Params {
    const void* x; const void* dy; float* ws;
    int N, Cin, H, W, Cout, OH, OW, pad_y, pad_x;
    int tilesX, tilesY, chunks, splits, coB, ciB;
    // operand-split form (pg_conv2d16_wgrad_x3): N = 6 * N0 "images"; image n = product g = n / N0 of sample r = n % N0, read from plane xpl[g] of x and plane
    // dpl[g] of dy (planes of N0 images each).  N0 = 0: the plain form.
    int N0;
    unsigned char xpl[8], dpl[8];
};

template <bool BF16>
__device__ __forceinline__ f32x16 mfma16(s16x8 a, s16x8 b, f32x16 c) {
    if (BF16) return __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(wbf16x8, a), __builtin_bit_cast(wbf16x8, b), c, 0, 0, 0);
    return __builtin_amdgcn_mfma_f32_32x32x16_f16(__builtin_bit_cast(wf16x8, a), __builtin_bit_cast(wf16x8, b), c, 0, 0, 0);
}

#ifndef W16_SHARED_WINDOWS
#define W16_SHARED_WINDOWS 1      // dev A/B: 0 = two transposing reads per tap (rounds 4)
#endif
template <int KH, int KW, int S, bool BF16>
__global__ __launch_bounds__(512, 1) void conv2d16_wgrad(Wgrad16Params p) {
    typedef W16Geo<KH, KW, S> G;
    typedef __attribute__((address_space(3))) s16x4* lp;
    extern __shared__ float smem[];
    const int lane = threadIdx.x & 63;
    const int wave8 = __builtin_amdgcn_readfirstlane((int)threadIdx.x >> 6);
    const bool loader = wave8 >= 4;
    const int wave = wave8 & 3, t = threadIdx.x & 255;
    int b = blockIdx.x;
    const int s = b % p.splits; b /= p.splits;
    const int cib = b % p.ciB, cob = b / p.ciB;
    const int co0 = cob * 64, ci0 = cib * 64;
    const unsigned smem_b = __builtin_amdgcn_readfirstlane(pgconv::lds_offset(smem));

    if (loader) {
        // gather maps: byte offsets from the chunk origin for every 16-byte slot this thread requests (sentinel = the DMA writes zeros)
        unsigned rel_dy[G::NDY], rel_x[G::NX];
#pragma unroll
        for (int i = 0; i < G::NDY; i++) {
            const int f = t + 256 * i, px = f >> 3, sl = f & 7;
            const int ch = sl ^ (((px >> 1) & 1) << 2);                        // the dy tile is always read at four consecutive pixels: swizzled at either stride
            const bool ok = px < G::PIX && co0 + ch * 8 < p.Cout;
            rel_dy[i] = ok ? (unsigned)(((px / G::TW) * p.OW + px % G::TW) * p.Cout + ch * 8) * 2u : WG_SENTINEL;
        }
#pragma unroll
        for (int i = 0; i < G::NX; i++) {
            const int f = t + 256 * i, pix = f / G::XSLOTS, sl = f % G::XSLOTS;
            const int ch = S == 1 ? (sl ^ (((pix >> 1) & 1) << 2)) : sl;
            const bool ok = pix < G::IH * G::IW && sl < 8 && ci0 + ch * 8 < p.Cin;
            rel_x[i] = ok ? (unsigned)(((pix / G::IW) * p.W + pix % G::IW) * p.Cin + ch * 8) * 2u : WG_SENTINEL;
        }
        auto issue = [&](int chk, int buf) __attribute__((always_inline)) {
            int c = chk;
            const int tx = c % p.tilesX; c /= p.tilesX;
            const int ty = c % p.tilesY;
            const int n = c / p.tilesY;
            int n_x = n, n_dy = n;
            if (p.N0) {
                const int g = n / p.N0, r = n - g * p.N0;
                n_x = p.xpl[g] * p.N0 + r; n_dy = p.dpl[g] * p.N0 + r;
            }
            const int oy0 = ty * G::R, ox0 = tx * G::TW;
            const int iy0 = oy0 * S - p.pad_y, ix0 = ox0 * S - p.pad_x;
            const uint64_t dyb = (uint64_t)(uintptr_t)p.dy + ((((int64_t)n_dy * p.OH + oy0) * p.OW + ox0) * p.Cout + co0) * 2;
            const uint64_t xb = (uint64_t)(uintptr_t)p.x + ((((int64_t)n_x * p.H + iy0) * p.W + ix0) * p.Cin + ci0) * 2;     // may lie before the tensor: masked below
            pgconv::i32x4 rdy, rx;
            rdy[0] = __builtin_amdgcn_readfirstlane((int)(unsigned)dyb); rdy[1] = __builtin_amdgcn_readfirstlane((int)(unsigned)(dyb >> 32) & 0xffff);
            rdy[2] = 0x7ffffffe; rdy[3] = 0x00020000;
            rx[0] = __builtin_amdgcn_readfirstlane((int)(unsigned)xb); rx[1] = __builtin_amdgcn_readfirstlane((int)(unsigned)(xb >> 32) & 0xffff);
            rx[2] = 0x7ffffffe; rx[3] = 0x00020000;
            const bool inner_dy = oy0 + G::R <= p.OH && ox0 + G::TW <= p.OW;
            const bool inner_x = iy0 >= 0 && iy0 + G::IH <= p.H && ix0 >= 0 && ix0 + G::IW <= p.W;
            const unsigned base_b = smem_b + (unsigned)(buf * G::BUF + 64 * 16 * wave);
#pragma unroll
            for (int i = 0; i < G::NDY; i++) {
                unsigned v = rel_dy[i];
                if (!inner_dy) {
                    const int px = (t + 256 * i) >> 3;
                    if (oy0 + px / G::TW >= p.OH || ox0 + px % G::TW >= p.OW) v = WG_SENTINEL;
                }
                pgconv::dma_dwordx4_buf(rdy, base_b + (unsigned)(256 * 16 * i), v, 0);
            }
#pragma unroll
            for (int i = 0; i < G::NX; i++) {
                unsigned v = rel_x[i];
                if (!inner_x) {
                    const int pix = (t + 256 * i) / G::XSLOTS;
                    const int iy = iy0 + pix / G::IW, ix = ix0 + pix % G::IW;
                    if (iy < 0 || iy >= p.H || ix < 0 || ix >= p.W) v = WG_SENTINEL;
                }
                pgconv::dma_dwordx4_buf(rx, base_b + (unsigned)((G::NDY + i) * 256 * 16), v, 0);
            }
        };
        int ch = s, g = 0;
        if (ch < p.chunks) issue(ch, 0);
        pgconv::dma_wait_all();
        __syncthreads();
        for (; ch < p.chunks; ch += p.splits, g++) {
            if (ch + p.splits < p.chunks) issue(ch + p.splits, (g & 1) ^ 1);
            pgconv::dma_wait_all();
            __syncthreads();
        }
        return;
    }

    // ---- multiplying waves
    const int mt = wave & 1, nt = wave >> 1;
    const int grp = lane >> 4, q = (lane >> 2) & 3, pp = lane & 3;
    const int kq = 8 * (grp >> 1) + q;                       // this lane's pixel row inside a 16-pixel k-step (first of the two reads; the second is 4 further)
    const int cbyte_a = (mt * 32 + 16 * (grp & 1) + 4 * pp) * 2, cbyte_b = (nt * 32 + 16 * (grp & 1) + 4 * pp) * 2;
    // dy tile: pixel kq + 16 kk + 4 j -- bit 1 of the pixel index is bit 1 of q: one address register
    const unsigned a_lane = (unsigned)(kq * 128 + (cbyte_a ^ (((q >> 1) & 1) << 6)));
    // x tile: pixel (r S + ky) IW + (c0 + kq + 4 j) S + kx.  Stride 1: IW % 4 == 0 and c0 % 16 == 0, so bit 1 of the pixel index is bit 1 of (q + kx): one register per kx
    unsigned b_lane[S == 1 ? KW : 1];
    if (S == 1) {
#pragma unroll
        for (int kx = 0; kx < KW; kx++) b_lane[kx] = (unsigned)((kq + kx) * G::XROWB + (cbyte_b ^ ((((q + kx) >> 1) & 1) << 6)));
    } else {
        b_lane[0] = (unsigned)(kq * S * G::XROWB + cbyte_b);
    }

    f32x16 acc[G::T];
#pragma unroll
    for (int tp = 0; tp < G::T; tp++)
#pragma unroll
        for (int k = 0; k < 16; k++) acc[tp][k] = 0.f;

    int ch = s, g = 0;
    __syncthreads();                             // chunk 0 has landed
    for (; ch < p.chunks; ch += p.splits, g++) {
        const unsigned dy_b = smem_b + (unsigned)((g & 1) * G::BUF);
        const unsigned x_b = dy_b + (unsigned)(G::NDY * 256 * 16);
#pragma unroll
        for (int kk = 0; kk < G::KSTEPS; kk++) {
            const int r = (kk * 16) / G::TW, c0 = (kk * 16) % G::TW;
            const unsigned aa = dy_b + a_lane + (unsigned)(kk * 16 * 128);
            const s16x4 a0 = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lp)(uintptr_t)aa);
            const s16x4 a1 = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lp)(uintptr_t)(aa + 4 * 128));
            const s16x8 av = __builtin_shufflevector(a0, a1, 0, 1, 2, 3, 4, 5, 6, 7);
            if constexpr (S == 1 && KW == 3 && W16_SHARED_WINDOWS) {
                // round 5: the three kx taps of a row read overlapping pixel windows (p .. p+7, p+1 .. p+8, p+2 .. p+9 of this lane's channel).  Three reads -- pixels
                // p .. p+3, p+4 .. p+7, p+8 .. p+11, all at the kx = 0 swizzle -- and four v_alignbit for the odd shift replace six: 11 instead of 20 transposing reads
                // per k-step.  (At 20 the four multiplying waves asked the LDS for 40 KB per k-step = 320 cycles at 128 B/clk against 288 cycles of MFMA.)
#pragma unroll
                for (int ky = 0; ky < KH; ky++) {
                    const unsigned ba = x_b + b_lane[0] + (unsigned)(((r + ky) * G::IW + c0) * G::XROWB);
                    const s16x4 r0 = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lp)(uintptr_t)ba);
                    const s16x4 r1 = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lp)(uintptr_t)(ba + 4 * G::XROWB));
                    const s16x4 r2 = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lp)(uintptr_t)(ba + 8 * G::XROWB));
                    typedef unsigned u32x2w __attribute__((ext_vector_type(2)));
                    typedef unsigned u32x4w __attribute__((ext_vector_type(4)));
                    const u32x2w d0 = __builtin_bit_cast(u32x2w, r0), d1 = __builtin_bit_cast(u32x2w, r1), d2 = __builtin_bit_cast(u32x2w, r2);
                    const u32x4w w0 = {d0[0], d0[1], d1[0], d1[1]};
                    const u32x4w w1 = {__builtin_amdgcn_alignbit(d0[1], d0[0], 16), __builtin_amdgcn_alignbit(d1[0], d0[1], 16),
                                       __builtin_amdgcn_alignbit(d1[1], d1[0], 16), __builtin_amdgcn_alignbit(d2[0], d1[1], 16)};
                    const u32x4w w2 = {d0[1], d1[0], d1[1], d2[0]};
                    acc[ky * KW + 0] = mfma16<BF16>(av, __builtin_bit_cast(s16x8, w0), acc[ky * KW + 0]);
                    acc[ky * KW + 1] = mfma16<BF16>(av, __builtin_bit_cast(s16x8, w1), acc[ky * KW + 1]);
                    acc[ky * KW + 2] = mfma16<BF16>(av, __builtin_bit_cast(s16x8, w2), acc[ky * KW + 2]);
                }
            } else {
#pragma unroll
            for (int ky = 0; ky < KH; ky++)
#pragma unroll
                for (int kx = 0; kx < KW; kx++) {
                    unsigned ba;
                    if (S == 1) ba = x_b + b_lane[kx] + (unsigned)(((r + ky) * G::IW + c0) * G::XROWB);
                    else ba = x_b + b_lane[0] + (unsigned)(((r * S + ky) * G::IW + c0 * S + kx) * G::XROWB);
                    const s16x4 b0 = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lp)(uintptr_t)ba);
                    const s16x4 b1 = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lp)(uintptr_t)(ba + 4 * S * G::XROWB));
                    const s16x8 bv = __builtin_shufflevector(b0, b1, 0, 1, 2, 3, 4, 5, 6, 7);
                    acc[ky * KW + kx] = mfma16<BF16>(av, bv, acc[ky * KW + kx]);
                }
            }
        }
        __syncthreads();
    }
    // partial block -> workspace[s][tap][co][ci] (D col = lane & 31 = ci, row = (reg & 3) + 8 (reg >> 2) + 4 (lane >> 5) = co)
    float* wsp = p.ws + (int64_t)s * G::T * p.Cout * p.Cin;
    const int ci = ci0 + nt * 32 + (lane & 31);
#pragma unroll
    for (int tp = 0; tp < G::T; tp++)
#pragma unroll
        for (int k = 0; k < 16; k++) {
            const int co = co0 + mt * 32 + (k & 3) + 8 * (k >> 2) + 4 * (lane >> 5);
            if (co < p.Cout && ci < p.Cin) wsp[((int64_t)tp * p.Cout + co) * p.Cin + ci] = acc[tp][k];
        }
}

// ---- the six operand-split products from ONE staging of the planes (round 5; stride-1 3x3, bf16) -----------------------------------------------------------------
// conv2d16_wgrad over 6 N plane-mapped images stages every plane three times (d1 and x1 are in three products each) and is bound by that traffic: a 64 x 64 block
// has 220 FLOP per staged byte, 11 TB/s of operand traffic at the bf16 peak.  Here a chunk is 2 output rows x 32 columns of ONE sample with all three planes of dy
// and of x side by side in LDS -- 3 x (8 KB + 18 KB), two buffers = 156 KB -- and a k-step issues the six products' 54 MFMAs from them: (d1|d2|d3, x1), (d1|d2, x2),
// (d1, x3), into the same nine accumulators.  Half the staged bytes per multiply, 33 transposing reads per 54 MFMAs instead of 66.
struct W16X3Geo {
    static constexpr int R = 2, TW = 32, PIX = R * TW, KSTEPS = PIX / 16, IH = R + 2, IW = 36, XROWB = 128;
    static constexpr int DY_SLOTS = PIX * 8, X_SLOTS = IH * IW * 8;                    // 512, 1152 sixteen-byte slots per plane
    static constexpr int NDY = DY_SLOTS / 256, NX = (X_SLOTS + 255) / 256;             // 2, 5 (the fifth pass: loader waves 0 and 1 only)
    static constexpr int DYB = DY_SLOTS * 16, XB = X_SLOTS * 16;                       // 8192, 18432 bytes
    static constexpr int BUF = 3 * (DYB + XB);                                         // 79872
    static_assert(X_SLOTS - 4 * 256 == 128, "the last pass covers exactly two loader waves");
    static_assert(2 * BUF <= 160 * 1024, "LDS budget");
};

__global__ __launch_bounds__(512, 1) void conv2d16_wgrad_x3k(Wgrad16Params p) {
    typedef W16X3Geo G;
    typedef __attribute__((address_space(3))) s16x4* lp;
    extern __shared__ float smem[];
    const int lane = threadIdx.x & 63;
    const int wave8 = __builtin_amdgcn_readfirstlane((int)threadIdx.x >> 6);
    const bool loader = wave8 >= 4;
    const int wave = wave8 & 3, t = threadIdx.x & 255;
    int b = blockIdx.x;
    const int s = b % p.splits; b /= p.splits;
    const int cib = b % p.ciB, cob = b / p.ciB;
    const int co0 = cob * 64, ci0 = cib * 64;
    const unsigned smem_b = __builtin_amdgcn_readfirstlane(pgconv::lds_offset(smem));
    const int N0 = p.N0;

    if (loader) {
        unsigned rel_dy[G::NDY], rel_x[G::NX];
#pragma unroll
        for (int i = 0; i < G::NDY; i++) {
            const int f = t + 256 * i, px = f >> 3, sl = f & 7;
            const int ch = sl ^ (((px >> 1) & 1) << 2);
            rel_dy[i] = co0 + ch * 8 < p.Cout ? (unsigned)(((px / G::TW) * p.OW + px % G::TW) * p.Cout + ch * 8) * 2u : WG_SENTINEL;
        }
#pragma unroll
        for (int i = 0; i < G::NX; i++) {
            const int f = t + 256 * i, pix = f >> 3, sl = f & 7;
            const int ch = sl ^ (((pix >> 1) & 1) << 2);
            const bool ok = f < G::X_SLOTS && ci0 + ch * 8 < p.Cin;
            rel_x[i] = ok ? (unsigned)(((pix / G::IW) * p.W + pix % G::IW) * p.Cin + ch * 8) * 2u : WG_SENTINEL;
        }
        auto issue = [&](int chk, int buf) __attribute__((always_inline)) {
            int c = chk;
            const int tx = c % p.tilesX; c /= p.tilesX;
            const int ty = c % p.tilesY;
            const int n = c / p.tilesY;
            const int oy0 = ty * G::R, ox0 = tx * G::TW;
            const int iy0 = oy0 - p.pad_y, ix0 = ox0 - p.pad_x;
            const bool inner_dy = oy0 + G::R <= p.OH && ox0 + G::TW <= p.OW;
            const bool inner_x = iy0 >= 0 && iy0 + G::IH <= p.H && ix0 >= 0 && ix0 + G::IW <= p.W;
            unsigned vdy[G::NDY], vx[G::NX];
#pragma unroll
            for (int i = 0; i < G::NDY; i++) {
                vdy[i] = rel_dy[i];
                if (!inner_dy) {
                    const int px = (t + 256 * i) >> 3;
                    if (oy0 + px / G::TW >= p.OH || ox0 + px % G::TW >= p.OW) vdy[i] = WG_SENTINEL;
                }
            }
#pragma unroll
            for (int i = 0; i < G::NX; i++) {
                vx[i] = rel_x[i];
                if (!inner_x) {
                    const int pix = (t + 256 * i) >> 3;
                    const int iy = iy0 + pix / G::IW, ix = ix0 + pix % G::IW;
                    if (iy < 0 || iy >= p.H || ix < 0 || ix >= p.W) vx[i] = WG_SENTINEL;
                }
            }
            const unsigned base_b = smem_b + (unsigned)(buf * G::BUF + 64 * 16 * wave);
#pragma unroll
            for (int pl = 0; pl < 3; pl++) {
                const int np = pl * N0 + n;
                const uint64_t dyb = (uint64_t)(uintptr_t)p.dy + ((((int64_t)np * p.OH + oy0) * p.OW + ox0) * p.Cout + co0) * 2;
                const uint64_t xb = (uint64_t)(uintptr_t)p.x + ((((int64_t)np * p.H + iy0) * p.W + ix0) * p.Cin + ci0) * 2;       // may lie before the tensor: masked above
                pgconv::i32x4 rdy, rx;
                rdy[0] = __builtin_amdgcn_readfirstlane((int)(unsigned)dyb); rdy[1] = __builtin_amdgcn_readfirstlane((int)(unsigned)(dyb >> 32) & 0xffff);
                rdy[2] = 0x7ffffffe; rdy[3] = 0x00020000;
                rx[0] = __builtin_amdgcn_readfirstlane((int)(unsigned)xb); rx[1] = __builtin_amdgcn_readfirstlane((int)(unsigned)(xb >> 32) & 0xffff);
                rx[2] = 0x7ffffffe; rx[3] = 0x00020000;
#pragma unroll
                for (int i = 0; i < G::NDY; i++) pgconv::dma_dwordx4_buf(rdy, base_b + (unsigned)(pl * G::DYB + 256 * 16 * i), vdy[i], 0);
#pragma unroll
                for (int i = 0; i < G::NX; i++)
                    if (i < G::NX - 1 || wave < 2) pgconv::dma_dwordx4_buf(rx, base_b + (unsigned)(3 * G::DYB + pl * G::XB + 256 * 16 * i), vx[i], 0);
            }
        };
        int ch = s, g = 0;
        if (ch < p.chunks) issue(ch, 0);
        pgconv::dma_wait_all();
        __syncthreads();
        for (; ch < p.chunks; ch += p.splits, g++) {
            if (ch + p.splits < p.chunks) issue(ch + p.splits, (g & 1) ^ 1);
            pgconv::dma_wait_all();
            __syncthreads();
        }
        return;
    }

    const int mt = wave & 1, nt = wave >> 1;
    const int grp = lane >> 4, q = (lane >> 2) & 3, pp = lane & 3;
    const int kq = 8 * (grp >> 1) + q;
    const int cbyte_a = (mt * 32 + 16 * (grp & 1) + 4 * pp) * 2, cbyte_b = (nt * 32 + 16 * (grp & 1) + 4 * pp) * 2;
    const unsigned a_lane = (unsigned)(kq * 128 + (cbyte_a ^ (((q >> 1) & 1) << 6)));
    const unsigned b_lane = (unsigned)(kq * G::XROWB + (cbyte_b ^ (((q >> 1) & 1) << 6)));
    typedef unsigned u32x2w __attribute__((ext_vector_type(2)));
    typedef unsigned u32x4w __attribute__((ext_vector_type(4)));

    f32x16 acc[9];
#pragma unroll
    for (int tp = 0; tp < 9; tp++)
#pragma unroll
        for (int k = 0; k < 16; k++) acc[tp][k] = 0.f;

    int ch = s, g = 0;
    __syncthreads();
    for (; ch < p.chunks; ch += p.splits, g++) {
        const unsigned dy_b = smem_b + (unsigned)((g & 1) * G::BUF);
        const unsigned x_b = dy_b + (unsigned)(3 * G::DYB);
        // software pipeline, written out: the three window reads of group (ky, x plane) + 1 are requested before the MFMAs of group (ky, x plane) are issued, and the
        // groups are fenced -- left alone the scheduler hoists every read of a k-step above its first MFMA and spills (256 registers + scratch; 252 vs 243 ms per iteration)
        struct Win { s16x4 r0, r1, r2; };
        struct Row { Win w[3]; };                                      // the window reads of one kernel row ky: x planes 0, 1, 2
        auto rd = [&](int kk, int ky) __attribute__((always_inline)) {
            const int r = (kk * 16) / G::TW, c0 = (kk * 16) % G::TW;
            Row o;
#pragma unroll
            for (int pl = 0; pl < 3; pl++) {
                const unsigned ba = x_b + (unsigned)(pl * G::XB) + b_lane + (unsigned)(((r + ky) * G::IW + c0) * G::XROWB);
                o.w[pl].r0 = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lp)(uintptr_t)ba);
                o.w[pl].r1 = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lp)(uintptr_t)(ba + 4 * G::XROWB));
                o.w[pl].r2 = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lp)(uintptr_t)(ba + 8 * G::XROWB));
            }
            return o;
        };
        auto rda = [&](int kk, s16x8 (&av)[3]) __attribute__((always_inline)) {
#pragma unroll
            for (int pl = 0; pl < 3; pl++) {
                const unsigned aa = dy_b + (unsigned)(pl * G::DYB) + a_lane + (unsigned)(kk * 16 * 128);
                const s16x4 a0 = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lp)(uintptr_t)aa);
                const s16x4 a1 = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lp)(uintptr_t)(aa + 4 * 128));
                av[pl] = __builtin_shufflevector(a0, a1, 0, 1, 2, 3, 4, 5, 6, 7);
            }
        };
        // software pipeline, written out: the nine window reads of kernel row ky + 1 (and the dy operands of the next k-step) are requested before the 18 MFMAs of row ky
        // are issued, and the rows are fenced -- left alone the scheduler hoists every read of a k-step above its first MFMA and spills (256 registers + scratch:
        // 252 instead of 243 ms per iteration); fenced per (row, plane) group the three-MFMA groups did not cover the next group's read latency
        Row cur = rd(0, 0), nxt = cur;
        s16x8 av[3], avn[3];
        rda(0, av);
#pragma unroll
        for (int kk = 0; kk < G::KSTEPS; kk++) {
#pragma unroll
            for (int ky = 0; ky < 3; ky++) {
                if (ky + 1 < 3) nxt = rd(kk, ky + 1);
                else if (kk + 1 < G::KSTEPS) { nxt = rd(kk + 1, 0); rda(kk + 1, avn); }
                __builtin_amdgcn_sched_barrier(0);
#pragma unroll
                for (int pl = 2; pl >= 0; pl--) {                      // x3 first: the smallest products enter the accumulators first
                    const u32x2w d0 = __builtin_bit_cast(u32x2w, cur.w[pl].r0), d1 = __builtin_bit_cast(u32x2w, cur.w[pl].r1), d2 = __builtin_bit_cast(u32x2w, cur.w[pl].r2);
                    const u32x4w w0 = {d0[0], d0[1], d1[0], d1[1]};
                    const u32x4w w1 = {__builtin_amdgcn_alignbit(d0[1], d0[0], 16), __builtin_amdgcn_alignbit(d1[0], d0[1], 16),
                                       __builtin_amdgcn_alignbit(d1[1], d1[0], 16), __builtin_amdgcn_alignbit(d2[0], d1[1], 16)};
                    const u32x4w w2 = {d0[1], d1[0], d1[1], d2[0]};
                    const s16x8 bw[3] = {__builtin_bit_cast(s16x8, w0), __builtin_bit_cast(s16x8, w1), __builtin_bit_cast(s16x8, w2)};
#pragma unroll
                    for (int i = 2 - pl; i >= 0; i--)                  // dy planes whose product with x plane pl is kept: i + pl <= 2
#pragma unroll
                        for (int kx = 0; kx < 3; kx++) acc[ky * 3 + kx] = mfma16<true>(av[i], bw[kx], acc[ky * 3 + kx]);
                }
                __builtin_amdgcn_sched_barrier(0);
                cur = nxt;
            }
#pragma unroll
            for (int pl = 0; pl < 3; pl++) av[pl] = avn[pl];
        }
        __syncthreads();
    }
    float* wsp = p.ws + (int64_t)s * 9 * p.Cout * p.Cin;
    const int ci = ci0 + nt * 32 + (lane & 31);
#pragma unroll
    for (int tp = 0; tp < 9; tp++)
#pragma unroll
        for (int k = 0; k < 16; k++) {
            const int co = co0 + mt * 32 + (k & 3) + 8 * (k >> 2) + 4 * (lane >> 5);
            if (co < p.Cout && ci < p.Cin) wsp[((int64_t)tp * p.Cout + co) * p.Cin + ci] = acc[tp][k];
        }
}

inline bool wgrad16_covers(int Cin, int Cout, int KH, int KW, int stride) {
    return Cin > 0 && Cout > 0 && Cin % 8 == 0 && Cout % 8 == 0 && ((KH == 3 && KW == 3 && (stride == 1 || stride == 2)) || (KH == 1 && KW == 1 && stride == 1));
}

}  // namespace

/* Number of K splits pg_conv2d_wgrad wants (its workspace is splits * KH*KW * Cout * Cin floats); 0 = geometry not covered. */
PG_EXPORT int pg_conv2d_wgrad_plan(int N, int Cin, int OH, int OW, int Cout, int KH, int KW, int stride) {
    if (N <= 0 || Cin <= 0 || OH <= 0 || OW <= 0 || Cout <= 0) return 0;
    if (!offsets_fit((int64_t)OH * OW) || !offsets_fit(((int64_t)OH * stride + KH) * ((int64_t)OW * stride + KW))) return 0;    // x plane bounded through the output extent
    if (fewcin_covers(Cin, KH, KW, stride)) {          // the few-channel form: one workgroup per 64 couts and K split
        const int64_t chunks = (int64_t)N * cdiv(OH, 2) * cdiv(OW, WG_TW);
        int64_t s = ((int64_t)pg::num_cu() + cdiv(Cout, WG_BM) - 1) / cdiv(Cout, WG_BM);
        if (s > chunks) s = chunks;
        return (int)(s < 1 ? 1 : s);
    }
    if (!((KH == 3 && KW == 3 && (stride == 1 || stride == 2)) || (KH == 1 && KW == 1 && stride == 1))) return 0;
    const int64_t chunks = (int64_t)N * cdiv(OH, stride == 1 ? 2 : 1) * cdiv(OW, WG_TW);
    const int blocks = cdiv(Cout, WG_BM) * cdiv(Cin, WG_BN);
    int64_t s = ((int64_t)pg::num_cu() + blocks - 1) / blocks;            // one (persistent-for-its-share) workgroup per CU
    if (s > chunks) s = chunks;
    if (s < 1) s = 1;
    if (s > 4096) s = 4096;
    return (int)s;
}

PG_EXPORT int pg_conv2d_wgrad(const float* x, const float* dy, float* dw, float* workspace,
                              int N, int Cin, int H, int W, int Cout, int KH, int KW, int stride, int pad_y, int pad_x, int OH, int OW,
                              int splits, void* stream) {
    if (!x || !dy || !dw || !workspace || N <= 0 || Cin <= 0 || H <= 0 || W <= 0 || Cout <= 0 || OH <= 0 || OW <= 0 || splits <= 0) return PG_ERR_INVALID_ARG;
    const bool fewcin = fewcin_covers(Cin, KH, KW, stride);
    if (!fewcin && !((KH == 3 && KW == 3 && (stride == 1 || stride == 2)) || (KH == 1 && KW == 1 && stride == 1))) return PG_ERR_UNSUPPORTED;
    if (OH != (H + 2 * pad_y - KH) / stride + 1 || OW != (W + 2 * pad_x - KW) / stride + 1 || pad_y < 0 || pad_x < 0) return PG_ERR_INVALID_ARG;
    if (!offsets_fit((int64_t)H * W) || !offsets_fit((int64_t)OH * OW)) return PG_ERR_TOO_LARGE;
    if (fewcin) {
        WgradParams q;
        q.x = x; q.dy = dy; q.ws = workspace;
        q.N = N; q.Cin = Cin; q.H = H; q.W = W; q.Cout = Cout; q.OH = OH; q.OW = OW; q.pad_y = pad_y; q.pad_x = pad_x;
        q.tilesX = cdiv(OW, WG_TW); q.tilesY = cdiv(OH, 2);
        const int64_t nchunks = (int64_t)N * q.tilesX * q.tilesY;
        if (nchunks > 0x7fffffffLL) return PG_ERR_TOO_LARGE;
        q.chunks = (int)nchunks; q.splits = splits; q.coB = cdiv(Cout, WG_BM); q.ciB = 1;
        const size_t lds = 2 * (size_t)WSmallGeo<7, 7>::BUF * sizeof(float);
        hipLaunchKernelGGL((conv2d_wgrad_fewcin<7, 7>), dim3((unsigned)(q.coB * splits)), dim3(768), lds, (hipStream_t)stream, q);
        int st0 = pg::launch_status();
        if (st0 != PG_OK) return st0;
        const int64_t tot = (int64_t)KH * KW * Cout * Cin;
        int64_t rb0 = (tot + 63) / 64;
        if (rb0 > pg::max_stream_blocks()) rb0 = pg::max_stream_blocks();
        hipLaunchKernelGGL(wgrad_reduce, dim3((unsigned)rb0), dim3(256), 0, (hipStream_t)stream, workspace, dw, splits, 1, Cout, Cin * KH * KW);
        return pg::launch_status();
    }
    WgradParams p;
    p.x = x; p.dy = dy; p.ws = workspace;
    p.N = N; p.Cin = Cin; p.H = H; p.W = W; p.Cout = Cout; p.OH = OH; p.OW = OW; p.pad_y = pad_y; p.pad_x = pad_x;
    p.tilesX = cdiv(OW, WG_TW); p.tilesY = cdiv(OH, stride == 1 ? 2 : 1);
    const int64_t chunks = (int64_t)N * p.tilesX * p.tilesY;
    if (chunks > 0x7fffffffLL) return PG_ERR_TOO_LARGE;
    p.chunks = (int)chunks; p.splits = splits;
    p.coB = cdiv(Cout, WG_BM); p.ciB = cdiv(Cin, WG_BN);
    const int64_t blocks = (int64_t)p.coB * p.ciB * splits;
    if (blocks > 0x7fffffffLL) return PG_ERR_TOO_LARGE;
    hipStream_t s = (hipStream_t)stream;
#define PG_WGRAD(KK, SS, TT) { \
        const size_t lds = WGeo<KK, KK, SS>::LDS_FLOATS * sizeof(float); \
        static pg::PerDeviceOnce attr; \
        const hipError_t e = attr.run([] { return hipFuncSetAttribute((const void*)conv2d_wgrad<KK, KK, SS, TT>, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024); }); \
        if (e != hipSuccess) return (int)e; \
        hipLaunchKernelGGL((conv2d_wgrad<KK, KK, SS, TT>), dim3((unsigned)blocks), dim3(256 * TT + 256), lds, s, p); }
    // A/B switch, off: measured in round 4 (tools/wgrad_probe.py, same box) -- two multiplying waves per SIMD by tap split are 13-17 % SLOWER than one
    // (128->128 at 256^2: 97.6 -> 83.9 TFLOP/s; config 4 291 -> 299 ms): the dy operand is read twice and, as in conv2d_wino4b.h, a second f32-MFMA wave on a
    // SIMD does not add matrix throughput, it shares it.
    static const bool tap_split = [] { const char* e = getenv("PG_WGRAD_TAPSPLIT"); return e ? atoi(e) != 0 : false; }();
    if (KH == 3 && stride == 1) { if (tap_split) PG_WGRAD(3, 1, 2) else PG_WGRAD(3, 1, 1) }
    else if (KH == 3) { if (tap_split) PG_WGRAD(3, 2, 2) else PG_WGRAD(3, 2, 1) }
    else PG_WGRAD(1, 1, 1)
#undef PG_WGRAD
    int st = pg::launch_status();
    if (st != PG_OK) return st;
    const int64_t total = (int64_t)KH * KW * Cout * Cin;
    int64_t rb = (total + 63) / 64;
    if (rb > pg::max_stream_blocks()) rb = pg::max_stream_blocks();
    hipLaunchKernelGGL(wgrad_reduce, dim3((unsigned)rb), dim3(256), 0, s, workspace, dw, splits, KH * KW, Cout, Cin);
    return pg::launch_status();
}

/* 16-bit weight gradient (channels-last x [N,H,W,Cin], dy [N,OH,OW,Cout] of `dtype` PG_F16 | PG_BF16; dw float32 [Cout,Cin,KH,KW]).
 * pg_conv2d16_wgrad_plan: the K splits to use (workspace = splits * KH*KW * Cout * Cin floats); 0 = geometry not covered (3x3 stride 1 | 2, 1x1;
 * channel counts multiples of 8). */
PG_EXPORT int pg_conv2d16_wgrad_plan(int N, int Cin, int OH, int OW, int Cout, int KH, int KW, int stride) {
    if (N <= 0 || OH <= 0 || OW <= 0 || !wgrad16_covers(Cin, Cout, KH, KW, stride)) return 0;
    const int64_t xpix = ((int64_t)OH * stride + KH) * ((int64_t)OW * stride + KW);
    if (xpix * Cin * 2 >= 0x7fff0000LL || (int64_t)OH * OW * Cout * 2 >= 0x7fff0000LL) return 0;     // 32-bit byte offsets inside one image
    const int64_t chunks = (int64_t)N * cdiv(OH, stride == 1 ? 4 : 2) * cdiv(OW, 32);
    const int blocks = cdiv(Cout, 64) * cdiv(Cin, 64);
    int64_t s = ((int64_t)pg::num_cu() + blocks - 1) / blocks;
    if (s > chunks) s = chunks;
    if (s < 1) s = 1;
    if (s > 4096) s = 4096;
    return (int)s;
}

static int wgrad16_run(const void* x, const void* dy, float* dw, float* workspace, int dtype,
                       int N, int Cin, int H, int W, int Cout, int KH, int KW, int stride, int pad_y, int pad_x, int OH, int OW,
                       int splits, void* stream, int N0) {
    if (!x || !dy || !dw || !workspace || N <= 0 || H <= 0 || W <= 0 || OH <= 0 || OW <= 0 || splits <= 0) return PG_ERR_INVALID_ARG;
    if (dtype != PG_F16 && dtype != PG_BF16) return PG_ERR_INVALID_ARG;
    if (!wgrad16_covers(Cin, Cout, KH, KW, stride)) return PG_ERR_UNSUPPORTED;
    if (OH != (H + 2 * pad_y - KH) / stride + 1 || OW != (W + 2 * pad_x - KW) / stride + 1 || pad_y < 0 || pad_x < 0) return PG_ERR_INVALID_ARG;
    if ((int64_t)H * W * Cin * 2 >= 0x7fff0000LL || (int64_t)OH * OW * Cout * 2 >= 0x7fff0000LL) return PG_ERR_TOO_LARGE;
    if (!pg::aligned16(x) || !pg::aligned16(dy)) return PG_ERR_UNSUPPORTED;
    Wgrad16Params p;
    p.x = x; p.dy = dy; p.ws = workspace;
    p.N = N; p.Cin = Cin; p.H = H; p.W = W; p.Cout = Cout; p.OH = OH; p.OW = OW; p.pad_y = pad_y; p.pad_x = pad_x;
    p.tilesX = cdiv(OW, 32); p.tilesY = cdiv(OH, stride == 1 ? 4 : 2);
    const int64_t chunks = (int64_t)N * p.tilesX * p.tilesY;
    if (chunks > 0x7fffffffLL) return PG_ERR_TOO_LARGE;
    p.chunks = (int)chunks; p.splits = splits;
    p.coB = cdiv(Cout, 64); p.ciB = cdiv(Cin, 64);
    p.N0 = N0;
    {   // products in ascending magnitude: (d3,x1) (d1,x3) (d2,x2) (d2,x1) (d1,x2) (d1,x1) -- a split's chunks run in image order, so the small terms are summed first
        static const unsigned char xt[8] = {0, 2, 1, 0, 1, 0, 0, 0}, dt[8] = {2, 0, 1, 1, 0, 0, 0, 0};
        for (int i = 0; i < 8; i++) { p.xpl[i] = xt[i]; p.dpl[i] = dt[i]; }
    }
    const int64_t blocks = (int64_t)p.coB * p.ciB * splits;
    if (blocks > 0x7fffffffLL) return PG_ERR_TOO_LARGE;
    hipStream_t s = (hipStream_t)stream;
    // the six operand-split products of a stride-1 3x3 layer from one staging of the planes (conv2d16_wgrad_x3k); PG_WGRAD_X3_FUSED=0: the 6N-image launch (A/B)
    static const bool x3_fused = [] { const char* e = getenv("PG_WGRAD_X3_FUSED"); return !e || atoi(e) != 0; }();
    if (N0 > 0 && x3_fused && KH == 3 && KW == 3 && stride == 1 && dtype == PG_BF16) {
        p.N = N0;
        p.tilesY = cdiv(OH, W16X3Geo::R);
        const int64_t ch3 = (int64_t)N0 * p.tilesX * p.tilesY;
        if (ch3 > 0x7fffffffLL) return PG_ERR_TOO_LARGE;
        p.chunks = (int)ch3;
        if (splits > p.chunks) splits = p.chunks;                 // (the plan counted the 6N launch's chunks: three times as many)
        p.splits = splits;
        static pg::PerDeviceOnce attr3;
        const hipError_t e3 = attr3.run([] { return hipFuncSetAttribute((const void*)conv2d16_wgrad_x3k, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024); });
        if (e3 != hipSuccess) return (int)e3;
        hipLaunchKernelGGL(conv2d16_wgrad_x3k, dim3((unsigned)((int64_t)p.coB * p.ciB * splits)), dim3(512), 2 * (size_t)W16X3Geo::BUF, s, p);
        int st3 = pg::launch_status();
        if (st3 != PG_OK) return st3;
        const int64_t total3 = (int64_t)KH * KW * Cout * Cin;
        int64_t rb3 = (total3 + 63) / 64;
        if (rb3 > pg::max_stream_blocks()) rb3 = pg::max_stream_blocks();
        hipLaunchKernelGGL(wgrad_reduce, dim3((unsigned)rb3), dim3(256), 0, s, workspace, dw, splits, KH * KW, Cout, Cin);
        return pg::launch_status();
    }
#define PG_WGRAD16(KK, SS, BF) { \
        const size_t lds = 2 * (size_t)W16Geo<KK, KK, SS>::BUF; \
        static pg::PerDeviceOnce attr; \
        const hipError_t e = attr.run([] { return hipFuncSetAttribute((const void*)conv2d16_wgrad<KK, KK, SS, BF>, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024); }); \
        if (e != hipSuccess) return (int)e; \
        hipLaunchKernelGGL((conv2d16_wgrad<KK, KK, SS, BF>), dim3((unsigned)blocks), dim3(512), lds, s, p); }
    const bool bf = dtype == PG_BF16;
    if (KH == 3 && stride == 1) { if (bf) PG_WGRAD16(3, 1, true) else PG_WGRAD16(3, 1, false) }
    else if (KH == 3) { if (bf) PG_WGRAD16(3, 2, true) else PG_WGRAD16(3, 2, false) }
    else { if (bf) PG_WGRAD16(1, 1, true) else PG_WGRAD16(1, 1, false) }
#undef PG_WGRAD16
    int st = pg::launch_status();
    if (st != PG_OK) return st;
    const int64_t total = (int64_t)KH * KW * Cout * Cin;
    int64_t rb = (total + 63) / 64;
    if (rb > pg::max_stream_blocks()) rb = pg::max_stream_blocks();
    hipLaunchKernelGGL(wgrad_reduce, dim3((unsigned)rb), dim3(256), 0, s, workspace, dw, splits, KH * KW, Cout, Cin);
    return pg::launch_status();
}

PG_EXPORT int pg_conv2d16_wgrad(const void* x, const void* dy, float* dw, float* workspace, int dtype,
                                int N, int Cin, int H, int W, int Cout, int KH, int KW, int stride, int pad_y, int pad_x, int OH, int OW,
                                int splits, void* stream) {
    return wgrad16_run(x, dy, dw, workspace, dtype, N, Cin, H, W, Cout, KH, KW, stride, pad_y, pad_x, OH, OW, splits, stream, 0);
}

// ---------------------------------------------------------------------------------------------------------------------------------------------------------
// float32 weight gradient on the bf16 matrix pipe by three-term operand splitting (round 5; the default for the 3x3 layers with >= 64 channels,
// PG_WGRAD_BF16X3=auto|0|1 in torch_utils/ops/conv2d_mfma.py).
//
// A float32 value is the exact sum of three bf16 values (8 + 8 + 8 significand bits, by truncation): x = x1 + x2 + x3, dy = d1 + d2 + d3.  Of the nine products
// the three smallest (d2 x3, d3 x2, d3 x3: <= 2^-24 of |dy x|) are dropped, the other six are bf16 x bf16 products -- exact in float32 -- accumulated in float32 by
// v_mfma_f32_32x32x16_bf16: float32-class arithmetic (measured against float64: 1.3e-6 ... 2.8e-6 of max|dw|, the fp32 kernel 1.0e-6 ... 1.7e-6) at 6 / 16 of the
// fp32 MFMA's cost per multiply.  The weight gradient's K axis is (image, pixel), so the six products are ONE launch of conv2d16_wgrad over 6 N "images" whose
// operands come from plane tables (Wgrad16Params::xpl / dpl): nothing is summed outside the kernel's own accumulation and split-K reduction.
//   pg_split3_bf16_cl: float32 NCHW -> three bf16 channels-last planes [3][N][H][W][C] (one transposing pass through LDS: 4 bytes read, 6 written per element).
// Integer-valued data below 2^24 splits and multiplies exactly, like the fp32 kernel.
namespace {
__global__ __launch_bounds__(256) void split3_bf16_cl_kernel(const float* __restrict__ x, unsigned short* __restrict__ out, int C, int HW, int64_t plane) {
    __shared__ float tile[64][65];                                    // [channel][pixel], odd pitch: the transposed reads below are conflict-free
    const int n = blockIdx.z, c0 = blockIdx.y * 64, p0 = blockIdx.x * 64;
    const float* xn = x + ((int64_t)n * C + c0) * HW + p0;
    {
        const int pq = (threadIdx.x & 15) * 4, cr = threadIdx.x >> 4;
#pragma unroll
        for (int i = 0; i < 4; i++) {
            const int c = cr + 16 * i;
            const bool cok = c0 + c < C;
            if (cok && p0 + pq + 3 < HW && (HW & 3) == 0) {
                const pgconv::f32x4 v = *(const pgconv::f32x4*)(xn + (int64_t)c * HW + pq);
#pragma unroll
                for (int e = 0; e < 4; e++) tile[c][pq + e] = v[e];
            } else {
#pragma unroll
                for (int e = 0; e < 4; e++) tile[c][pq + e] = (cok && p0 + pq + e < HW) ? xn[(int64_t)c * HW + pq + e] : 0.f;
            }
        }
    }
    __syncthreads();
    const int cg = (threadIdx.x & 7) * 8;                              // eight consecutive channels = one 16-byte word per plane
#pragma unroll
    for (int i = 0; i < 2; i++) {
        const int px = (threadIdx.x >> 3) + 32 * i;
        if (p0 + px >= HW || c0 + cg >= C) continue;
        unsigned hi[4], mid[4], lo[4];
#pragma unroll
        for (int d = 0; d < 4; d++) {
            unsigned h2[2], m2[2], l2[2];
#pragma unroll
            for (int e = 0; e < 2; e++) {
                const float v = tile[cg + 2 * d + e][px];
                const unsigned hb = __builtin_bit_cast(unsigned, v) & 0xffff0000u;
                // a non-finite value keeps its leading plane and gets zero remainders (inf - inf would turn an overflow into NaN where the fp32 kernel keeps +-inf: ADVICE r5)
                const bool fin = (hb & 0x7f800000u) != 0x7f800000u;
                const float r = fin ? v - __builtin_bit_cast(float, hb) : 0.f;
                const unsigned mb = __builtin_bit_cast(unsigned, r) & 0xffff0000u;
                const float r2 = r - __builtin_bit_cast(float, mb);
                h2[e] = hb >> 16; m2[e] = mb >> 16; l2[e] = __builtin_bit_cast(unsigned, r2) >> 16;
            }
            hi[d] = h2[0] | (h2[1] << 16); mid[d] = m2[0] | (m2[1] << 16); lo[d] = l2[0] | (l2[1] << 16);
        }
        unsigned short* o = out + (((int64_t)n * HW + p0 + px) * C + c0 + cg);
        *(pgconv::i32x4*)(o) = (pgconv::i32x4){(int)hi[0], (int)hi[1], (int)hi[2], (int)hi[3]};
        *(pgconv::i32x4*)(o + plane) = (pgconv::i32x4){(int)mid[0], (int)mid[1], (int)mid[2], (int)mid[3]};
        *(pgconv::i32x4*)(o + 2 * plane) = (pgconv::i32x4){(int)lo[0], (int)lo[1], (int)lo[2], (int)lo[3]};
    }
}
}  // namespace

/* float32 NCHW x [N, C, HW] -> out: three bf16 channels-last planes [3][N][HW][C] with x == plane 0 + plane 1 + plane 2 exactly (truncation split).  C % 8 == 0. */
PG_EXPORT int pg_split3_bf16_cl(const float* x, void* out, int N, int C, int64_t HW, void* stream) {
    if (!x || !out || N <= 0 || C <= 0 || HW <= 0) return PG_ERR_INVALID_ARG;
    if (C % 8 != 0 || !pg::aligned16(x) || !pg::aligned16(out)) return PG_ERR_UNSUPPORTED;
    if (HW > 0x7fffffffLL || N > 65535 || (C + 63) / 64 > 65535) return PG_ERR_TOO_LARGE;
    const dim3 grid((unsigned)((HW + 63) / 64), (unsigned)((C + 63) / 64), (unsigned)N);
    hipLaunchKernelGGL(split3_bf16_cl_kernel, grid, dim3(256), 0, (hipStream_t)stream, x, (unsigned short*)out, C, (int)HW, (int64_t)N * HW * C);
    return pg::launch_status();
}

/* dw of y = conv2d(x, w) from the split operands: x3 = pg_split3_bf16_cl(x) [3][N][H][W][Cin], dy3 = pg_split3_bf16_cl(dy) [3][N][OH][OW][Cout]; dw float32
 * [Cout][Cin][KH][KW]; `splits` = pg_conv2d16_wgrad_plan(6 * N, ...), workspace = splits * KH*KW * Cout * Cin floats.  Geometries of pg_conv2d16_wgrad. */
PG_EXPORT int pg_conv2d16_wgrad_x3(const void* x3, const void* dy3, float* dw, float* workspace,
                                   int N, int Cin, int H, int W, int Cout, int KH, int KW, int stride, int pad_y, int pad_x, int OH, int OW,
                                   int splits, void* stream) {
    if (N <= 0 || (int64_t)N * 6 > 0x7fffffffLL) return PG_ERR_INVALID_ARG;
    return wgrad16_run(x3, dy3, dw, workspace, PG_BF16, 6 * N, Cin, H, W, Cout, KH, KW, stride, pad_y, pad_x, OH, OW, splits, stream, N);
}

