// Streaming 1x1 convolution for gfx950 on v_mfma_f32_32x32x2_f32: the fp32 NCHW 1x1 layers of the synthesis network
// (merge_conv on cat([x, garment features]), the SPADE blocks' skip convolutions; reference networks.py:287-316 Conv2dLayer /
// :1586-1635 Spade_Conv2dLayer with kernel_size = 1) -- HBM-bound layers (one output per 64..192 MACs) that the tiled kernel of
// conv2d_kernel.h runs at ~3 TB/s because its per-chunk structure (halo staging, barrier, operand reads) dominates at K = Cin.
//
// A 1x1 convolution needs no halo, so the B operand goes from global memory straight into the MFMA:
//   * pixels are the flattened H*W axis; MFMA column c of a wave = the pixels p0 + E*c .. p0 + E*c + E-1, i.e. lane c loads ONE E-float
//     vector per input channel (16 bytes for E = 4: a wave-instruction reads 2 x 512 contiguous bytes) and feeds E MFMAs with it;
//     lanes 32-63 take the odd channel of the pair;
//   * D rows are couts, and for a fixed cout a lane holds the E consecutive pixels of its column: 16-byte (E = 4) stores;
//   * weights of the workgroup's 32*MT couts sit in LDS as [Cin][32*MT] (conflict-free A reads), with the per-sample input scale of a
//     modulated convolution multiplied in when they are loaded (per (n, cout block), not per pixel);
//   * no barrier and no LDS traffic for activations in the K loop; channel-pair loads are requested a group (4 pairs) ahead;
//   * MT x E = 8 accumulators of 16 registers: (MT 2, E 4) for Cout <= 64 -- the form that is used: 64->64 at 512^2 275 vs 311 us,
//     128->64 412 vs 417 us against the tiled kernel; (MT 4, E 2) for Cout <= 128 measured 8-12 % slower (8-byte accesses, 244
//     VGPRs), two 64-cout passes over the same pixel tile (input from L2 the second time) -3 ... +7 %: neither is dispatched.  One workgroup of 8 waves per CU (226 VGPRs): 2 waves per SIMD is what limits it to ~3.9 TB/s;
//   * two-source mode (channels [split, Cin) from x2), epilogue: * out_scale, + bias, act, gain, clamp, + residual -- as conv2d_kernel.h.
// Roofline: HBM; algorithmic bytes 4 * N * H*W * (Cin + Cout) (+ residual).
#pragma once
#include "conv2d_kernel.h"

namespace pgconv {

template <int E> struct PixVec;
template <> struct PixVec<4> { typedef float type __attribute__((ext_vector_type(4))); };
template <> struct PixVec<2> { typedef float type __attribute__((ext_vector_type(2))); };

template <int MT, int E>
__global__ __launch_bounds__(512, 1) void conv1x1_stream(ConvParams p) {
    typedef typename PixVec<E>::type vec_t;
    constexpr int BM = 32 * MT, WPIX = 32 * E, WAVES = 8, GRP = 4;          // pixels per wave tile; channel pairs per load group
    extern __shared__ __attribute__((aligned(16))) float smem[];           // ws[cin_loop][BM], then ep_scale[BM], ep_bias[BM]
    const int cin_loop = (p.Cin + 2 * GRP - 1) / (2 * GRP) * (2 * GRP);
    float* ws = smem;
    float* ep_scale = smem + cin_loop * BM;
    float* ep_bias = ep_scale + BM;
    const int t = threadIdx.x, lane = t & 63;
    const int wave = __builtin_amdgcn_readfirstlane(t >> 6);
    const int half = lane >> 5, l31 = lane & 31;
    const int64_t HW = (int64_t)p.H * p.W;
    const int ptiles = (int)((HW + WPIX * WAVES - 1) / (WPIX * WAVES));
    const int split = p.f.x2 ? p.f.cin_split : p.Cin;
    const float gain = p.f.gain, slope = act_slope(p.f.act, p.f.alpha);
    const float cl = p.f.clamp >= 0.f ? p.f.clamp : __builtin_inff();
    const bool plain_tail = slope == 1.f && gain == 1.f && p.f.clamp < 0.f;

    int cur_key = -1;
    // a contiguous share of the tile range per workgroup: consecutive pixel tiles of one (n, m-block) -> the weights are loaded once
    const int per = (p.total_tiles + (int)gridDim.x - 1) / (int)gridDim.x;
    const int t_end = min(p.total_tiles, ((int)blockIdx.x + 1) * per);
    for (int tile = (int)blockIdx.x * per; tile < t_end; tile++) {
        // tile -> (n, m-block, pixel tile), pixel tile fastest: the weights in LDS change only with (n, m-block)
        const int pt = tile % ptiles;
        const int key = tile / ptiles;
        const int mb = key % p.mblocks, n = key / p.mblocks;
        const int m0 = mb * BM;
        if (key != cur_key) {
            __syncthreads();                                               // every wave is done with the previous weights
            for (int e = t; e < cin_loop * BM; e += 512) {
                const int ci = e / BM, co = e % BM;
                float v = (ci < p.Cin && m0 + co < p.CoutP) ? p.wp[(int64_t)ci * p.CoutP + m0 + co] : 0.f;
                if (p.f.in_scale && ci < p.Cin) v *= p.f.in_scale[(int64_t)n * p.Cin + ci];
                ws[e] = v;
            }
            if (t < BM) {
                const int co = m0 + t;
                const bool ok = co < p.Cout;
                ep_scale[t] = ok ? (p.f.out_scale ? p.f.out_scale[(int64_t)n * p.Cout + co] : 1.f) : 0.f;
                ep_bias[t] = (ok && p.f.bias) ? p.f.bias[co] : 0.f;
            }
            __syncthreads();
            cur_key = key;
        }
        const int64_t p0 = ((int64_t)pt * WAVES + wave) * WPIX;            // this wave's first pixel
        if (p0 >= HW) continue;                                            // (wave-uniform; no barrier below)
        const int64_t pl = p0 + (int64_t)E * l31;                          // this lane's first pixel
        const bool lane_ok = pl + E <= HW;                                 // HW % E == 0 (host): a lane's vector is inside or outside as a whole
        const int64_t plc = lane_ok ? pl : 0;
        const float* x1 = p.x + (int64_t)n * split * HW + plc;
        const float* x2 = p.f.x2 ? p.f.x2 + (int64_t)n * (p.Cin - split) * HW + plc : x1;

        f32x16 acc[MT][E];
#pragma unroll
        for (int mt = 0; mt < MT; mt++)
#pragma unroll
            for (int e = 0; e < E; e++)
#pragma unroll
                for (int k = 0; k < 16; k++) acc[mt][e][k] = 0.f;

        auto load_group = [&](int j0, vec_t (&dst)[GRP]) __attribute__((always_inline)) {
#pragma unroll
            for (int u = 0; u < GRP; u++) {
                int ci = 2 * (j0 + u) + half;
                if (ci >= p.Cin) ci = p.Cin - 1;                          // padded tail: its weights are zero
                const float* src = ci < split ? x1 + (int64_t)ci * HW : x2 + (int64_t)(ci - split) * HW;
                dst[u] = *(const vec_t*)src;
            }
        };
        const int npairs = cin_loop / 2;
        vec_t cur[GRP], nxt[GRP];
        load_group(0, cur);
        for (int j0 = 0; j0 < npairs; j0 += GRP) {
            if (j0 + GRP < npairs) load_group(j0 + GRP, nxt);
            __builtin_amdgcn_sched_barrier(0);
#pragma unroll
            for (int u = 0; u < GRP; u++) {
                float a[MT];
#pragma unroll
                for (int mt = 0; mt < MT; mt++) a[mt] = ws[(2 * (j0 + u) + half) * BM + mt * 32 + l31];
#pragma unroll
                for (int mt = 0; mt < MT; mt++)
#pragma unroll
                    for (int e = 0; e < E; e++) acc[mt][e] = __builtin_amdgcn_mfma_f32_32x32x2f32(a[mt], cur[u][e], acc[mt][e], 0, 0, 0);
            }
            __builtin_amdgcn_sched_barrier(0);
#pragma unroll
            for (int u = 0; u < GRP; u++) cur[u] = nxt[u];
        }

        // ---- epilogue: D col = lane & 31 = pixel column (E consecutive pixels across the E accumulators), row = cout
        if (lane_ok) {
            float* yb = p.y + (int64_t)n * p.ys[0] + pl;
            const float* rb = p.f.residual ? p.f.residual + (int64_t)n * p.ys[0] + pl : nullptr;
#pragma unroll
            for (int mt = 0; mt < MT; mt++)
#pragma unroll
                for (int k = 0; k < 16; k++) {
                    const int row = mt * 32 + (k & 3) + 8 * (k >> 2) + 4 * half;
                    const int co = m0 + row;
                    if (co < p.Cout) {
                        const float sc = ep_scale[row], bi = ep_bias[row];
                        vec_t v;
#pragma unroll
                        for (int e = 0; e < E; e++) {
                            float r = fmaf(acc[mt][e][k], sc, bi);
                            if (!plain_tail) { r = r > 0.f ? r : r * slope; r = fminf(fmaxf(r * gain, -cl), cl); }
                            v[e] = r;
                        }
                        if (rb) v += *(const vec_t*)(rb + (int64_t)co * p.ys[1]);
                        *(vec_t*)(yb + (int64_t)co * p.ys[1]) = v;
                    }
                }
        }
    }
}

// ---- Ring form (round 3): the same mapping, (MT 2, E 4), for Cout a multiple of 64 (one pass over the pixels per 64-cout block: the
// blocks of an image run side by side on different workgroups, so all but one pass come from L2 / MALL), Cin % 32 == 0, whole
// 1024-pixel tile rows, no residual.
// The plain-C++ prefetch of conv1x1_stream above does not prefetch: `cur = nxt` copies registers whose loads are in flight, so hipcc
// puts `s_waitcnt vmcnt(0)` between the loads of group g + 1 and the MFMAs of group g (found in the assembly) and every group exposes a
// full memory round trip with two waves per SIMD to cover it: 3.9 TB/s.  Here the input words are inline-asm loads into a RING of four
// groups (4 channel pairs x 16 bytes each), requested THREE groups (12 KB per wave, 96 KB per CU) ahead with hand-counted vmcnt, and the
// request stream runs across tile boundaries: the first three groups of the next tile are in flight while this tile's 32 stores issue.
// vmcnt returns loads and stores in issue order, so the wait for group g counts what is younger than its four loads: the three groups
// behind it (12) and, for the first three groups after an epilogue, that epilogue's 32 stores (44).  Every store of the epilogue is
// unconditional (64-cout blocks, full tiles) -- the count is exact; anything else the compiler issues in between (the weight reload at an
// image change) only makes a wait conservative.
constexpr int S1_RING = 4, S1_AHEAD = 3, S1_GRP = 4;
__global__ __launch_bounds__(512, 1) void conv1x1_stream_ring(ConvParams p) {
    constexpr int BM = 64, E = 4, WPIX = 32 * E, WAVES = 8, GRP = S1_GRP;
    extern __shared__ __attribute__((aligned(16))) float smem[];           // ws[Cin][BM], then ep_scale[BM], ep_bias[BM]
    float* ws = smem;
    float* ep_scale = smem + p.Cin * BM;
    float* ep_bias = ep_scale + BM;
    const int t = threadIdx.x, lane = t & 63;
    const int wave = __builtin_amdgcn_readfirstlane(t >> 6);
    const int half = lane >> 5, l31 = lane & 31;
    const int HW = p.H * p.W;                                              // (host: Cin * HW * 4 < 2^31)
    const int ptiles = HW / (WPIX * WAVES);
    const int split = p.f.x2 ? p.f.cin_split : p.Cin;
    const int ngroups = p.Cin / (2 * GRP);
    const float gain = p.f.gain, slope = act_slope(p.f.act, p.f.alpha);
    const float cl = p.f.clamp >= 0.f ? p.f.clamp : __builtin_inff();
    const bool plain_tail = slope == 1.f && gain == 1.f && p.f.clamp < 0.f;

    typedef float f32x4v __attribute__((ext_vector_type(4)));
    f32x4v ring[S1_RING][GRP];

    // where a tile's input lives: per-image bases (uniform) + this lane's byte offset (its 4 pixels, its channel of a pair)
    struct Src { const float* b1; const float* b2; unsigned voff; };
    auto src_of = [&](int tile) {
        const int pt = tile % ptiles, n = (tile / ptiles) / p.mblocks;
        Src s;
        s.b1 = p.x + (int64_t)n * split * HW;
        s.b2 = p.f.x2 ? p.f.x2 + (int64_t)n * (p.Cin - split) * HW : s.b1;
        s.voff = (unsigned)(((pt * WAVES + wave) * WPIX + E * l31) + half * HW) * 4u;
        return s;
    };
    auto request = [&](int slot, const Src& s, int g) __attribute__((always_inline)) {
        const int c0 = 2 * GRP * g;                                        // first channel of the group; split % 8 == 0: one source per group
        const float* base = c0 < split ? s.b1 : s.b2;
        const unsigned goff = (unsigned)((c0 < split ? c0 : c0 - split) * HW) * 4u;
        asm volatile("s_nop 4" ::: "memory");       // `base` may come straight from a spill lane (v_readlane -> SGPR -> VMEM needs 5 wait states; asm operands are invisible to the hazard recognizer)
#pragma unroll
        for (int u = 0; u < GRP; u++) {
            const unsigned off = s.voff + goff + (unsigned)(2 * u * HW) * 4u;
            asm volatile("global_load_dwordx4 %0, %1, %2" : "=v"(ring[slot][u]) : "v"(off), "s"(base) : "memory");
        }
    };

    const int per = (p.total_tiles + (int)gridDim.x - 1) / (int)gridDim.x;
    const int t_begin = (int)blockIdx.x * per, t_end = min(p.total_tiles, t_begin + per);
    if (t_begin >= t_end) return;
    Src cur = src_of(t_begin);
#pragma unroll
    for (int g = 0; g < S1_AHEAD; g++) request(g, cur, g);                 // (Cin >= 32: ngroups >= 4 > S1_AHEAD)
    int cur_key = -1;
    bool after_store = false;
    for (int tile = t_begin; tile < t_end; tile++) {
        const int key = tile / ptiles;
        const int mb = key % p.mblocks, n = key / p.mblocks;
        const int m0 = mb * BM;
        const int pt = tile % ptiles;
        const Src nxt = tile + 1 < t_end ? src_of(tile + 1) : cur;        // past the end: re-request this tile (keeps the counts static)
        if (key != cur_key) {
            __syncthreads();                                               // every wave is done with the previous weights
            for (int e = t; e < p.Cin * BM; e += 512) {
                const int ci = e / BM, co = e % BM;
                float v = p.wp[(int64_t)ci * p.CoutP + m0 + co];
                if (p.f.in_scale) v *= p.f.in_scale[(int64_t)n * p.Cin + ci];
                ws[e] = v;
            }
            if (t < BM) {
                const int co = m0 + t;
                ep_scale[t] = p.f.out_scale ? p.f.out_scale[(int64_t)n * p.Cout + co] : 1.f;
                ep_bias[t] = p.f.bias ? p.f.bias[co] : 0.f;
            }
            __syncthreads();
            cur_key = key;
        }
        f32x16 acc[2][E];
#pragma unroll
        for (int mt = 0; mt < 2; mt++)
#pragma unroll
            for (int e = 0; e < E; e++)
#pragma unroll
                for (int k = 0; k < 16; k++) acc[mt][e][k] = 0.f;

        for (int j = 0; j < ngroups; j += S1_RING) {
#pragma unroll
            for (int s = 0; s < S1_RING; s++) {
                const int g = j + s, gp = g + S1_AHEAD;
                if (gp < ngroups) request((s + S1_AHEAD) % S1_RING, cur, gp);
                else request((s + S1_AHEAD) % S1_RING, nxt, gp - ngroups);
                if (s < S1_AHEAD && j == 0 && after_store)
                    asm volatile("s_waitcnt vmcnt(%4)" : "+v"(ring[s][0]), "+v"(ring[s][1]), "+v"(ring[s][2]), "+v"(ring[s][3]) : "n"(S1_AHEAD * GRP + 32) : "memory");
                else
                    asm volatile("s_waitcnt vmcnt(%4)" : "+v"(ring[s][0]), "+v"(ring[s][1]), "+v"(ring[s][2]), "+v"(ring[s][3]) : "n"(S1_AHEAD * GRP) : "memory");
#pragma unroll
                for (int u = 0; u < GRP; u++) {
                    const float* wr = ws + (2 * (GRP * g + u) + half) * BM + l31;
                    const float a0 = wr[0], a1 = wr[32];
#pragma unroll
                    for (int e = 0; e < E; e++) acc[0][e] = __builtin_amdgcn_mfma_f32_32x32x2f32(a0, ring[s][u][e], acc[0][e], 0, 0, 0);
#pragma unroll
                    for (int e = 0; e < E; e++) acc[1][e] = __builtin_amdgcn_mfma_f32_32x32x2f32(a1, ring[s][u][e], acc[1][e], 0, 0, 0);
                }
            }
        }

        // ---- epilogue: exactly 32 unconditional 16-byte stores (the vmcnt(44) above counts them)
        float* yb = p.y + (int64_t)n * p.ys[0] + ((pt * WAVES + wave) * WPIX + E * l31);
#pragma unroll
        for (int mt = 0; mt < 2; mt++)
#pragma unroll
            for (int k = 0; k < 16; k++) {
                const int row = mt * 32 + (k & 3) + 8 * (k >> 2) + 4 * half;
                const float sc = ep_scale[row], bi = ep_bias[row];
                f32x4v v;
#pragma unroll
                for (int e = 0; e < E; e++) {
                    float r = fmaf(acc[mt][e][k], sc, bi);
                    if (!plain_tail) { r = r > 0.f ? r : r * slope; r = fminf(fmaxf(r * gain, -cl), cl); }
                    v[e] = r;
                }
                float* dst = yb + (int64_t)(m0 + row) * p.ys[1];
                asm volatile("global_store_dwordx4 %0, %1, off" :: "v"(dst), "v"(v) : "memory");
            }
        after_store = true;
        cur = nxt;
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");                       // the three groups requested past the end
}

inline bool s1x1_ring_ok(const ConvParams& p) {
    const int64_t HW = (int64_t)p.H * p.W;
    if (p.Cout % 64 != 0 || p.CoutP != p.Cout || p.Cin % 32 != 0 || p.Cin < 32 || HW % 1024 != 0 || HW < 4096 || p.f.residual) return false;
    if (p.in_xform || p.f.noise || p.f.spade_x || p.ksplit > 1) return false;
    if (p.osy != 1 || p.osx != 1 || p.ooy != 0 || p.oox != 0 || p.pad_y != 0 || p.pad_x != 0) return false;
    if (p.ys[3] != 1 || p.ys[2] != p.OW || p.ys[1] != HW || p.OH != p.H || p.OW != p.W) return false;     // flattened pixel axis
    if ((((uintptr_t)p.x) | ((uintptr_t)p.y) | ((uintptr_t)p.f.x2)) & 15) return false;
    if ((p.ys[0] & 3) != 0) return false;
    if (p.f.x2 && (p.f.cin_split % 8 != 0 || p.f.cin_split <= 0 || p.f.cin_split >= p.Cin)) return false;
    if ((int64_t)p.Cin * HW * 4 > 0x7fffffffLL) return false;              // 32-bit lane offsets within one image
    return (size_t)(p.Cin + 2) * 64 * 4 <= 150 * 1024;
}

inline int launch_s1x1_ring(const ConvParams& p0, hipStream_t s) {
    ConvParams p = p0;
    const int64_t HW = (int64_t)p.H * p.W;
    const int64_t ptiles = HW / 1024;
    p.mblocks = p.Cout / 64;                                            // each 64-cout block streams the pixels again (the other blocks' workgroups read the same image at about the same time: L2 / MALL)
    const int64_t tiles = ptiles * p.mblocks * p.N;
    if (tiles > 0x7fffffffLL) return PG_ERR_TOO_LARGE;
    p.total_tiles = (int)tiles;
    const size_t lds = (size_t)(p.Cin + 2) * 64 * sizeof(float);
    static PerDeviceOnce attr;
    const hipError_t e = attr.run([] { return hipFuncSetAttribute((const void*)conv1x1_stream_ring, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024); });
    if (e != hipSuccess) return (int)e;
    const int64_t blocks = tiles < (int64_t)num_cu() ? tiles : (int64_t)num_cu();
    hipLaunchKernelGGL(conv1x1_stream_ring, dim3((unsigned)blocks), dim3(512), lds, s, p);
    return launch_status();
}

// True if the streaming kernel takes this launch (else the caller uses the tiled kernel).
inline bool s1x1_ok(const ConvParams& p) {
    const int64_t HW = (int64_t)p.H * p.W;
    if (p.in_xform || p.f.noise || p.f.spade_x || p.ksplit > 1) return false;
    if (p.osy != 1 || p.osx != 1 || p.ooy != 0 || p.oox != 0 || p.pad_y != 0 || p.pad_x != 0) return false;
    if (p.ys[3] != 1 || p.ys[2] != p.OW || p.ys[1] != HW || p.OH != p.H || p.OW != p.W) return false;     // flattened pixel axis
    if (p.Cout > 64 || p.Cout < 32) return false;            // measured: the (MT 4, E 2) form for 65..128 couts is 8-12 % SLOWER than the tiled kernel
    const int E = p.Cout <= 64 ? 4 : 2;
    if (HW % E != 0 || HW < 4096) return false;
    if ((((uintptr_t)p.x) | ((uintptr_t)p.y) | ((uintptr_t)p.f.x2) | ((uintptr_t)p.f.residual)) & 15) return false;
    if ((p.ys[0] & 3) != 0) return false;
    const int BM = p.Cout <= 64 ? 64 : 128;
    const int cin_loop = (p.Cin + 7) / 8 * 8;
    return (size_t)(cin_loop + 2) * BM * 4 <= 150 * 1024;
}

template <int MT, int E>
int launch_s1x1_t(const ConvParams& p0, hipStream_t s) {
    ConvParams p = p0;
    constexpr int BM = 32 * MT;
    const int64_t HW = (int64_t)p.H * p.W;
    const int64_t ptiles = (HW + 32 * E * 8 - 1) / (32 * E * 8);
    p.mblocks = (p.Cout + BM - 1) / BM;
    const int64_t tiles = ptiles * p.mblocks * p.N;
    if (tiles > 0x7fffffffLL) return PG_ERR_TOO_LARGE;
    p.total_tiles = (int)tiles;
    const int cin_loop = (p.Cin + 7) / 8 * 8;
    const size_t lds = (size_t)(cin_loop + 2) * BM * sizeof(float);
    static PerDeviceOnce attr;
    const hipError_t e = attr.run([] { return hipFuncSetAttribute((const void*)conv1x1_stream<MT, E>, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024); });
    if (e != hipSuccess) return (int)e;
    int per_cu = (int)((160 * 1024) / lds);
    if (per_cu > 1) per_cu = 1;                              // 512 threads x ~170 VGPRs
    if (per_cu < 1) per_cu = 1;
    const int64_t blocks = tiles < (int64_t)num_cu() * per_cu ? tiles : (int64_t)num_cu() * per_cu;
    hipLaunchKernelGGL((conv1x1_stream<MT, E>), dim3((unsigned)blocks), dim3(512), lds, s, p);
    return launch_status();
}

}  // namespace pgconv
