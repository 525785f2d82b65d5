// 16-bit `up = 2` modulated 3x3 layer in ONE launch with the FIR's x half in the epilogue (round 5).
//
// What it replaces in the reference: conv2d_resample's up-by-2 branch (torch_utils/ops/conv2d_resample.py:125-142): a stride-2
// conv_transpose2d of the 3x3 kernel -> (2H+1) x (2W+1) intermediate, then upfirdn2d with the 4-tap filter (padding 1, gain 4)
// -> 2H x 2W, then noise / bias_act (training/networks.py:73-94, 170-179).  The composite route of conv2d_kernel16.h folds BOTH
// filter axes into the weights: four 3x3 phase kernels of the low-resolution input = 36 tap-products per input position where
// the reference's transposed convolution has 9.  Here only the y axis is folded (the filter is separable, f = fy (x) fx):
//
//     zq[2q+a][X] = sum_{ty, kx} Ky_a[ty][kx] * x[q + ty - 1][(X - kx) / 2]     Ky = (2 fy) (*)_y w, phase a of it: 3 y taps
//                   -- in x the plain transposed convolution: even X = 2r has 2 taps (x[r-1], x[r]), odd X = 2r+1 has one (x[r])
//     out[2q+a][o] = sum_k (2 fx)[o + 2 - X] * zq[2q+a][X],  X = o-1 .. o+2     -- the x half of the FIR, in registers
//
// = 2 (a) x 3 (ty) x 3 (x tap / parity pairs) = 18 tap-products per position.  The MFMA's N axis runs along x (lane = position r,
// D column), and a lane holds BOTH x parities of its position (phases (a, 0), (a, 1) are two accumulators), so the x filter needs
// the neighbouring lanes' values only: two DPP wave shifts per accumulator register pair, no LDS exchange, no intermediate
// rounding (the reference's half-precision path rounds the (2H+1)^2 intermediate to 16 bit; this does not), no barrier.
// The price is one overlap column on either side of a tile: 30 of the 32 lanes of a tile produce output.
//
// GEMM view (as conv2d_kernel16.h): M = 4 phases x 32 couts (phase p = 2a + b: output row parity a, x parity b of the
// INTERMEDIATE), N = 16 rows x 32 positions, K = Cin x (3 x 2 taps); taps with tx = 0 (sample r-1) exist for b = 0 only, so a
// 16-channel step is 36 MFMAs per wave (24 + 12) behind 18 weight and 8 activation fragment reads.  Workgroup = 8 multiplying
// waves (2 position rows each: 128 accumulator registers) + 4 loader waves (the two-role form of conv2d_kernel16.h: chunk requests
// by 16-byte LDS-DMA two chunks ahead through three staging buffers, per-tile side loads), one workgroup per CU, persistent.
//
// Weights: pg_conv2d16_pack_weight of the [Cin, 4 * Cout, 3, 2] stack (phase-major along Cout; built on the host,
// training/networks.py `_up2_fused_weights`); the loader gathers a workgroup's 4 x 32 rows from the four phase blocks and
// never requests the (b = 1, tx = 0) taps (they are zero by construction; the range check of the descriptor fills them).
//
// Roofline: HBM at the top resolutions (2 * (numel(x) + numel(y)) bytes), MFMA below; executed flops = 2 * 18 * N*H*W*Cin*Cout
// x 32/30 (overlap) against 2 * 9 * ... of SURVEY 8d's count for the transposed convolution.

#pragma once
#include "conv2d_kernel16.h"

namespace pgconv16 {

struct Up2fParams {
    Conv16Params c;         // x [N,H,W,Cin]; wp = packed 3x2 stack; y [N, 2H, 2W, Cout] (ys strides); Cout = channels of y; f.phase_cout = Cout
    float fir[4];           // 2 * fx: the x half of the filter with its share of the gain
};

// MW multiplying waves (2 position rows each) + MW / 2 loader waves.  MW = 8: one workgroup per CU (16-row tiles, three staging buffers);
// MW = 4: TWO workgroups per CU (8-row tiles, two staging buffers, <= 80 KB of LDS each) -- the two run out of step, so one's epilogue
// (vector work: the x filter, the activation, the stores) overlaps the other's K loop instead of stopping the matrix pipe and the load stream.
template <int MW, int NB>
struct Up2fGeo {
    static constexpr int LDW = MW / 2, NTHREADS = (MW + LDW) * 64, LT = LDW * 64;      // loader waves / threads
    static constexpr int WG_PER_CU = 8 / MW;
    static constexpr int TH = 2 * MW, LW = 32, UW = 30;    // position rows, lanes (positions incl. one overlap column per side), useful positions
    static constexpr int T = 6, KC = 16, NBUF = NB;
    static constexpr int NSTEP = 18;                       // (ty, tx, phase) weight fragments with a non-zero tap: per ty, tx = 1 x phases 0..3, then tx = 0 x phases 0, 2
    static constexpr int IH_T = TH + 2, IW_T = LW + 1;
    static constexpr int NPIX = IH_T * IW_T;               // halo pixels
    // The halo is staged in 1 KB groups of 32 consecutive pixels, [group][k-half][32 pixels] x 16 bytes: a wave's ds_read_b128 (lanes 0-31: 32 consecutive
    // pixels of k-half 0, lanes 32-63: of k-half 1) walks whole 256-byte bank rows per 16 lanes -- conflict-free without an XOR swizzle -- and ONE LDS-DMA
    // instruction (64 consecutive slots) fetches BOTH 16-byte halves of 32 pixels' 32-byte channel chunk, so lanes l and l + 32 ask for adjacent bytes.
    // (First cut: two whole k-half planes -- every 16-byte request its own cache line, twice the L2 requests for the same bytes: the request stream of a
    // step with nothing else running took 53-68 us on the five config-5 shapes, 36-57 us in this layout; tools/up2f_probe.py, PG_CONV16_DBG=12.)
    static constexpr int NGRP = (NPIX + 31) / 32;
    static constexpr int NXS = NGRP * 64;
    static constexpr int NXS_PAD = (NXS + LT - 1) / LT * LT;
    static constexpr int NWS = NSTEP * 2 * 32;             // 1152 weight slots: [step][k-half][32 couts]
    static constexpr int NWS_PAD = (NWS + LT - 1) / LT * LT;
    static constexpr int LDS_BUF = NXS_PAD + NWS_PAD;
    static constexpr int NREQ_X = NXS_PAD / LT, NREQ = LDS_BUF / LT;
    static constexpr int EPS = 64, NOISE = 4 * TH * LW;
    static constexpr int EP_FLOATS = 2 * EPS + NOISE;
    static constexpr size_t LDS_BYTES = (size_t)NBUF * LDS_BUF * 16 + (size_t)2 * EP_FLOATS * 4 + 256;
    static_assert(MW == 8 || MW == 4, "one or two workgroups per CU");
    static_assert(LDS_BYTES <= 160 * 1024 / WG_PER_CU, "LDS budget");
};

template <int CTRL> __device__ __forceinline__ float dpp_f(float v) {
    return __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), CTRL, 0xf, 0xf, true));
}
constexpr int DPP_WAVE_SHL1 = 0x130, DPP_WAVE_SHR1 = 0x138;     // lane l reads lane l + 1 / lane l - 1
// acc += tap * v[lane + 1] / v[lane - 1] in ONE instruction (v_fmac_f32 with a DPP source; bound_ctrl: lanes without a neighbour read 0).  The compiler's own
// DPP combine leaves v_mov_b32_dpp + v_fmac pairs here (three extra instructions per value pair); `v` must not have been written by the few vector
// instructions in front of this one (a DPP read needs two wait states after a VALU write): the callers pass accumulator registers the MFMAs wrote long ago.
__device__ __forceinline__ void fmac_shl(float& acc, float v, float tap) {
    asm volatile("v_fmac_f32_dpp %0, %1, %2 wave_shl:1 row_mask:0xf bank_mask:0xf bound_ctrl:1" : "+v"(acc) : "v"(v), "v"(tap));
}
__device__ __forceinline__ void fmac_shr(float& acc, float v, float tap) {
    asm volatile("v_fmac_f32_dpp %0, %1, %2 wave_shr:1 row_mask:0xf bank_mask:0xf bound_ctrl:1" : "+v"(acc) : "v"(v), "v"(tap));
}

template <typename T, int MW, int NB>
__global__ __launch_bounds__((MW + MW / 2) * 64, 3) void conv2d_up2f16(Up2fParams pp) {
    const Conv16Params& p = pp.c;
    typedef Up2fGeo<MW, NB> G;
    typedef Half16<T> HT;
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    typedef const __attribute__((address_space(3))) i32x4* lds_v4;
    typedef const __attribute__((address_space(3))) f32x4* lds_f4;
    typedef const __attribute__((address_space(3))) float* lds_f;

    const int t = threadIdx.x, lane = t & 63;
    const int wave = __builtin_amdgcn_readfirstlane(t >> 6);
    const int half = lane >> 5, l31 = lane & 31;
    const int total = p.total_tiles;
    const int q8 = total >> 3, r8 = total & 7;
    const int nchunks = p.Cin / G::KC;
    const int pc = p.f.phase_cout;
    const int dbg_ = PG_CONV16_STAMPS ? p.dbg : 0;      // dev ablations (diagnostic build only; results wrong by design): 1 no stores, 4 no MFMA, 8 no epilogue, 16 no per-cout constant reads, 32 no shifted multiply-adds, 64 no activation, 128 no halo DMA

    // tile -> (n, first position row, position of lane 0, first cout), XCD-aware like conv2d_mfma16 (scalar unit)
    auto decode = [&](int tile, int& n, int& q0, int& r0, int& m0) __attribute__((always_inline)) {
        const int xcd = tile & 7;
        unsigned L = (unsigned)((xcd < r8 ? xcd * (q8 + 1) : r8 * (q8 + 1) + (xcd - r8) * q8) + (tile >> 3));
        unsigned q = div_magic(L, p.m_mblocks); const int mb = (int)(L - q * p.mblocks); L = q;
        q = div_magic(L, p.m_tilesX); const int tx = (int)(L - q * p.tilesX); L = q;
        q = div_magic(L, p.m_tilesY); const int ty = (int)(L - q * p.tilesY);
        n = (int)q; q0 = ty * G::TH; r0 = tx * G::UW - 1; m0 = mb * 32;      // r0: the left overlap column
    };

    if (wave >= MW) {
        // ---------------------------------------------------------------- loader waves: requests only
        const int lw = wave - MW;
        const unsigned smem_b = __builtin_amdgcn_readfirstlane(lds_offset(smem));
        const unsigned side_b = smem_b + (unsigned)(G::NBUF * G::LDS_BUF) * 16u;
        const unsigned dump_b = side_b + 2u * G::EP_FLOATS * 4u;
        const i32x4 xrsrc = make_rsrc(p.x, (int64_t)p.N * p.H * p.W * p.xC * 2);
        const i32x4 wrsrc = make_rsrc(p.wp, p.w_bytes);
        const int lt = lw * 64 + lane;
        unsigned rel[G::NREQ], hyx[G::NREQ_X];
#pragma unroll
        for (int j = 0; j < G::NREQ; j++) {
            const int s = j * G::LT + lt;
            rel[j] = SENTINEL;
            if (j < G::NREQ_X) {
                hyx[j] = 0x4000u;
                const int c = (s >> 5) & 1, qh = (s >> 6) * 32 + (s & 31);
                const int hy = qh / G::IW_T, hx = qh % G::IW_T;
                if (qh < G::NPIX && s < G::NXS) {
                    rel[j] = (unsigned)((hy * p.W + hx) * p.xC + c * 8) * 2u;
                    hyx[j] = (unsigned)hy | ((unsigned)hx << 16);
                }
            } else {
                const int e = s - G::NXS_PAD;                          // [step i][k-half][32 couts]
                const int i = e >> 6, kh = (e >> 5) & 1, c = e & 31;
                const int ty = i / 6, r = i % 6, tx = r < 4 ? 1 : 0, ph = r < 4 ? r : (r - 4) * 2;
                if (e < G::NWS) rel[j] = (unsigned)(((ty * 2 + tx) * 2 + kh) * p.CoutP + ph * pc + c) * 16u;
            }
        }
        const bool side_scale = lw == 0, side_bias = lw == 1;
        const i32x4 sbrsrc = side_scale ? make_rsrc(p.f.out_scale, p.f.out_scale ? (int64_t)p.N * pc * 4 : 0)
                                        : make_rsrc(p.f.bias, (p.f.bias && side_bias) ? (int64_t)pc * 4 : 0);
        const i32x4 nrsrc = make_rsrc(p.f.noise, p.f.noise ? ((int64_t)(p.N - 1) * p.f.noise_batch_stride + 3 * p.f.noise_phase_stride + (int64_t)p.H * p.W) * 4 : 0);

        int c_tile = blockIdx.x, c_chunk = 0, c_ahead = 0, ibuf = 0;
        unsigned voff[G::NREQ_X];
        unsigned w_soff = 0, x_soff0 = 0;
        auto issue_next = [&]() __attribute__((always_inline)) {
            if (c_chunk == 0) {
                int n, q0, r0, m0;
                decode(c_tile, n, q0, r0, m0);
                const int ty0 = q0 - 1, tx0 = r0 - 1;
                const unsigned org = (unsigned)((ty0 * p.W + tx0) * p.xC * 2);
                const bool interior = ty0 >= 0 && tx0 >= 0 && ty0 + G::IH_T <= p.H && tx0 + G::IW_T <= p.W;
#pragma unroll
                for (int j = 0; j < G::NREQ_X; j++) {
                    if (interior) {
                        voff[j] = org + rel[j];
                    } else {
                        const unsigned gy = (unsigned)(ty0 + (int)(hyx[j] & 0xffffu)), gx = (unsigned)(tx0 + (int)(hyx[j] >> 16));
                        voff[j] = (gy < (unsigned)p.H && gx < (unsigned)p.W) ? org + rel[j] : SENTINEL;
                    }
                }
                x_soff0 = (unsigned)((int64_t)n * p.H * p.W * p.xC * 2);
                w_soff = (unsigned)(((int64_t)n * p.w_nstride + (int64_t)m0 * 8) * 2);
                // per-tile side loads: 32 demodulation scales, 32 biases, 4 x TH x 32 noise samples (tile T's into side buffer T & 1)
                const unsigned sb_ = side_b + (unsigned)((c_ahead & 1) * G::EP_FLOATS) * 4u;
                const bool live = (side_scale || side_bias) && lane < 32 && m0 + lane < pc;
                dma4(sbrsrc, (side_scale || side_bias) ? sb_ + (unsigned)(lw * 64) * 4u : dump_b, live ? (unsigned)lane * 4u : SENTINEL,
                     (unsigned)(m0 + (side_scale ? n * pc : 0)) * 4u);
#pragma unroll
                for (int j = 0; j < G::TH * G::LW / G::LT; j++) {
                    const int tt = (j * G::LDW + lw) * 64 + lane;
                    const int nq = q0 + tt / G::LW, nr = r0 + tt % G::LW;
                    const unsigned nvo = (nq < p.H && nr >= 0 && nr < p.W) ? (unsigned)(nq * p.W + nr) * 4u : SENTINEL;
#pragma unroll
                    for (int ph = 0; ph < 4; ph++)
                        dma4(nrsrc, sb_ + (unsigned)(2 * G::EPS + ph * (G::TH * G::LW) + (j * G::LDW + lw) * 64) * 4u, nvo,
                             (unsigned)(n * p.f.noise_batch_stride + ph * p.f.noise_phase_stride) * 4u);
                }
            }
            const unsigned x_soff = x_soff0 + (unsigned)(c_chunk * G::KC) * 2u;
            const unsigned wk_soff = w_soff + (unsigned)(c_chunk * G::T * 2) * (unsigned)p.CoutP * 16u;
            const unsigned buf_b = smem_b + (unsigned)(ibuf * G::LDS_BUF) * 16u;
#pragma unroll
            for (int j = 0; j < G::NREQ; j++) {
                const bool is_w = j >= G::NREQ_X;
                dma16(is_w ? wrsrc : xrsrc, buf_b + (unsigned)(j * G::LT + lw * 64) * 16u, is_w ? rel[j] : ((dbg_ & 128) ? SENTINEL : voff[j]), is_w ? wk_soff : x_soff);
            }
            ibuf = ibuf == G::NBUF - 1 ? 0 : ibuf + 1;
            if (++c_chunk == nchunks) { c_chunk = 0; c_tile += gridDim.x; c_ahead++; }
        };
        const int my_chunks = ((total - (int)blockIdx.x + (int)gridDim.x - 1) / (int)gridDim.x) * nchunks;
        int req = 0;
#pragma unroll
        for (int j = 0; j < G::NBUF - 1; j++)
            if (req < my_chunks) { issue_next(); req++; }
        for (int c = 0; c < my_chunks; c++) {
            // chunk c must have landed; with three buffers chunk c + 1's requests (and, in front of them, its tile's side loads) may stay in flight
            if (G::NBUF > 2 && req > c + 1) vm_wait<G::NREQ>(); else vm_wait<0>();
            __builtin_amdgcn_s_barrier();
            if (req < my_chunks) { issue_next(); req++; }               // into the buffer of chunk c - 1: every multiplying wave is past it
        }
        return;
    }

    // -------------------------------------------------------------------- multiplying waves
    const __amdgpu_buffer_rsrc_t yrsrc = __builtin_amdgcn_make_buffer_rsrc(p.y, 0, (int)(p.y_bytes > 0x7fffffffLL ? 0x7fffffffLL : p.y_bytes), 0x00020000);
    f32x16 acc[4][2];
#pragma unroll
    for (int ph = 0; ph < 4; ph++)
#pragma unroll
        for (int nt = 0; nt < 2; nt++)
#pragma unroll
            for (int k = 0; k < 16; k++) acc[ph][nt][k] = 0.f;

    // operand addresses: the weights ONE lane address + immediates; the activations by halo pixel q = q_lane + (row, tap) offset: group q >> 5, k-half, pixel q & 31
    const unsigned a_lane = (unsigned)(G::NXS_PAD + half * 32 + l31) * 16u;

    // One 16-channel chunk: 18 weight fragments (ty, tx, phase), each multiplied with the two position rows of the wave.  The loop is
    // written as a software pipeline by hand -- the weight fragment of step i + 1 is requested before the MFMAs of step i, the activation
    // fragments rotate through FOUR register sets (row ty leaves after its last use, row ty + 2 takes its place) -- and pinned with
    // sched_barrier: left to itself the scheduler (at the register limit) sinks every read to its first use and waits lgkmcnt(0) there.
    auto compute_chunk = [&](int buf) __attribute__((always_inline)) {
        int lane_k;                                                    // (fresh lane id, asm volatile: the eight activation addresses are rebuilt per chunk, 4 vector
        asm volatile("v_mbcnt_lo_u32_b32 %0, -1, 0\n\tv_mbcnt_hi_u32_b32 %0, -1, %0" : "=v"(lane_k));      // instructions each -- hoisted out of the loop they were spilled and came back through scratch loads)
        const int q_lane = (2 * wave) * G::IW_T + (lane_k & 31);
        const unsigned char* bb = smem + (size_t)buf * G::LDS_BUF * 16 + (lane_k >> 5) * 512;
        const unsigned char* ab = smem + (size_t)buf * G::LDS_BUF * 16 + a_lane;
        auto b_frag = [&](int hr, int tx) __attribute__((always_inline)) {
            const int q = q_lane + hr * G::IW_T + tx;
            return *(lds_v4)(bb + (size_t)(((q >> 5) << 10) + ((q & 31) << 4)));
        };
        auto a_frag = [&](int i) __attribute__((always_inline)) { return *(lds_v4)(ab + (size_t)(i * 64) * 16); };
        i32x4 b1[4], b0[4];                                           // [halo row] for tx = 1 / tx = 0; at most four of the eight are live
        b1[0] = b_frag(0, 1); b1[1] = b_frag(1, 1);
        i32x4 a_cur = a_frag(0), a_nxt;
        b0[0] = b_frag(0, 0); b0[1] = b_frag(1, 0);
#pragma unroll
        for (int i = 0; i < G::NSTEP; i++) {
            const int ty = i / 6, r = i % 6, tx = r < 4 ? 1 : 0, ph = r < 4 ? r : (r - 4) * 2;
            if (i + 1 < G::NSTEP) a_nxt = a_frag(i + 1);
            if (ty < 2 && r == 3) b1[ty + 2] = b_frag(ty + 2, 1);      // (a new register set: row ty of this column has its last use in this step)
            if (ty < 2 && r == 5) b0[ty + 2] = b_frag(ty + 2, 0);
            const i32x4 blo = tx ? b1[ty] : b0[ty], bhi = tx ? b1[ty + 1] : b0[ty + 1];
            acc[ph][0] = HT::mma(a_cur, blo, acc[ph][0]);
            acc[ph][1] = HT::mma(a_cur, bhi, acc[ph][1]);
            a_cur = a_nxt;
            __builtin_amdgcn_sched_barrier(0);
        }
    };

    // ---- epilogue constants.  v = clamp(act((h * scale + noise + bias)) * gain) with a positively homogeneous activation (linear / relu / lrelu, gain > 0):
    // u = h * (scale * gain) + (bias + noise) * gain, v = med3(max(u, u * slope), -cl, cl).  Without a per-cout scale (per-sample weight packs carry the
    // demodulation) the gain rides in the filter taps and the bias + noise term is the filter sum's start value: 4 multiply-adds + 1 add per value.
    const float gain = p.f.gain;
    const float cl = p.f.clamp >= 0.f ? p.f.clamp : __builtin_inff();
    const float slope = act_slope(p.f.act, p.f.alpha);
    const bool has_scale = p.f.out_scale != nullptr;
    const float ng = (p.f.noise ? p.f.noise_gain : 0.f) * gain;
    const float tg = has_scale ? 1.f : gain;
    const float f0 = pp.fir[0] * tg, f1 = pp.fir[1] * tg, f2 = pp.fir[2] * tg, f3 = pp.fir[3] * tg;

    auto write_tile = [&](int tile, int dpar, auto scaled) __attribute__((always_inline)) {
        float f0v, f1v, f3v;                       // the taps a DPP multiply-add takes from a VECTOR register (VOP2); made here: not live through the K loop
        asm volatile("v_mov_b32 %0, %3\n\tv_mov_b32 %1, %4\n\tv_mov_b32 %2, %5" : "=v"(f0v), "=v"(f1v), "=v"(f3v) : "s"(f0), "s"(f1), "s"(f3));
        constexpr bool SCALED = decltype(scaled)::value;
        // lane-derived values are rebuilt here from a fresh lane id (asm volatile: not hoisted): kept across the K loop they were 18 spilled registers
        int lane_e;
        asm volatile("v_mbcnt_lo_u32_b32 %0, -1, 0\n\tv_mbcnt_hi_u32_b32 %0, -1, %0" : "=v"(lane_e));
        const int half = lane_e >> 5, l31 = lane_e & 31;
        int e_n, e_q0, e_r0, e_m0;
        decode(tile, e_n, e_q0, e_r0, e_m0);
        const unsigned char* side = smem + (size_t)G::NBUF * G::LDS_BUF * 16 + (size_t)dpar * G::EP_FLOATS * 4;
        // One (position row, output-row parity) at a time: its two accumulator pairs (32 registers) are dead after the block, so the register pressure
        // falls as the tile is written out.  Everything a store needs (pixel offset, masks) is rebuilt AT the store from a fresh lane id: values kept
        // across the arithmetic were spilled, and a scratch reload is a vector-memory load -- its s_waitcnt vmcnt(0) also waits for the block's own
        // output stores (an HBM round trip, sixteen times per tile: that, not the instruction count, was most of the 24 k cycles of this epilogue).
#pragma unroll
        for (int nt = 0; nt < 2; nt++) {
            const int row_l = 2 * wave + nt, q = e_q0 + row_l;
#pragma unroll
            for (int a = 0; a < 2; a++) {
                const float nze = *(lds_f)(side + (size_t)(2 * G::EPS + (2 * a) * (G::TH * G::LW) + row_l * G::LW + l31) * 4) * ng;
                const float nzo = *(lds_f)(side + (size_t)(2 * G::EPS + (2 * a + 1) * (G::TH * G::LW) + row_l * G::LW + l31) * 4) * ng;
                u32x2 pe, po;                                              // group g - 1's packed results (even g), waiting for their exchange partner
#pragma unroll
                for (int g = 0; g < 4; g++) {
                    const int r0c = 8 * g + 4 * half;
                    f32x4 bgv = {gain, gain, gain, gain}, sgv = {gain, gain, gain, gain};
                    if (!(dbg_ & 16)) {
                        bgv = *(lds_f4)(side + (size_t)(G::EPS + r0c) * 4) * gain;
                        if constexpr (SCALED) sgv = *(lds_f4)(side + (size_t)r0c * 4) * gain;
                    }
                    float ve[4], vo[4];
#pragma unroll
                    for (int j = 0; j < 4; j++) {
                        const float z0 = acc[2 * a][nt][4 * g + j], z1 = acc[2 * a + 1][nt][4 * g + j];
                        float ue, uo;
                        if constexpr (SCALED) {
                            float he = fmaf(f2, z0, f1 * z1), ho = fmaf(f3, z0, f2 * z1);
                            if (!(dbg_ & 32)) {
                                fmac_shr(he, z1, f3v); fmac_shl(he, z0, f0v);
                                fmac_shl(ho, z0, f1v); fmac_shl(ho, z1, f0v);
                            }
                            ue = fmaf(he, sgv[j], bgv[j] + nze); uo = fmaf(ho, sgv[j], bgv[j] + nzo);
                        } else {
                            ue = fmaf(f2, z0, fmaf(f1, z1, bgv[j] + nze)); uo = fmaf(f3, z0, fmaf(f2, z1, bgv[j] + nzo));
                            if (!(dbg_ & 32)) {
                                fmac_shr(ue, z1, f3v); fmac_shl(ue, z0, f0v);
                                fmac_shl(uo, z0, f1v); fmac_shl(uo, z1, f0v);
                            }
                        }
                        if (dbg_ & 64) { ve[j] = ue; vo[j] = uo; continue; }
                        ve[j] = __builtin_amdgcn_fmed3f(fmaxf(ue, ue * slope), -cl, cl);
                        vo[j] = __builtin_amdgcn_fmed3f(fmaxf(uo, uo * slope), -cl, cl);
                    }
                    u32x2 ce = {HT::pack(ve[0], ve[1]), HT::pack(ve[2], ve[3])}, co2 = {HT::pack(vo[0], vo[1]), HT::pack(vo[2], vo[3])};
                    if (g & 1) {
                        // lanes 32-63 of group g - 1 <-> lanes 0-31 of group g: 8 consecutive couts of one pixel per lane
#pragma unroll
                        for (int d = 0; d < 2; d++) {
                            const auto re = __builtin_amdgcn_permlane32_swap(pe[d], ce[d], false, false);
                            pe[d] = re[0]; ce[d] = re[1];
                            const auto ro = __builtin_amdgcn_permlane32_swap(po[d], co2[d], false, false);
                            po[d] = ro[0]; co2[d] = ro[1];
                        }
                        int lane_s;
                        asm volatile("v_mbcnt_lo_u32_b32 %0, -1, 0\n\tv_mbcnt_hi_u32_b32 %0, -1, %0" : "=v"(lane_s));
                        const int ls = lane_s & 31, rs = e_r0 + ls;
                        const int co = e_m0 + 8 * (g - 1 + (lane_s >> 5));
                        const bool ok = ls >= 1 && ls <= G::UW && rs < p.W && q < p.H && co < pc && !(dbg_ & 1);
                        const unsigned pix_off = (unsigned)((int64_t)e_n * p.ys[0] + (int64_t)(2 * q + a) * p.ys[2] + (int64_t)(2 * rs) * p.ys[3]);
                        const unsigned se = ok ? (pix_off + (unsigned)co) * 2u : SENTINEL;
                        const unsigned so = ok ? (pix_off + (unsigned)p.ys[3] + (unsigned)co) * 2u : SENTINEL;
                        __builtin_amdgcn_raw_buffer_store_b128(u32x4{pe[0], pe[1], ce[0], ce[1]}, yrsrc, (int)se, 0, 0);
                        __builtin_amdgcn_raw_buffer_store_b128(u32x4{po[0], po[1], co2[0], co2[1]}, yrsrc, (int)so, 0, 0);
                    } else {
                        pe = ce; po = co2;
                    }
                    __builtin_amdgcn_sched_barrier(0);      // (one group of 2 x 4 values at a time: interleaving the groups of a tile for ILP costs more registers than the wave has beside its 128 accumulators)
                }
            }
        }
    };

    int tile = blockIdx.x, cbuf = 0, dpar = 0;
    for (;;) {
        for (int k = 0; k < nchunks; k++) {
            __builtin_amdgcn_s_barrier();
            if (!(dbg_ & 4)) compute_chunk(cbuf);
            cbuf = cbuf == G::NBUF - 1 ? 0 : cbuf + 1;
        }
        if (!(dbg_ & 8)) {
            if (has_scale) write_tile(tile, dpar, std::true_type{}); else write_tile(tile, dpar, std::false_type{});
        }
        if (tile + (int)gridDim.x >= total) break;
        tile += gridDim.x;
        dpar ^= 1;
#pragma unroll
        for (int ph = 0; ph < 4; ph++)
#pragma unroll
            for (int nt = 0; nt < 2; nt++)
#pragma unroll
                for (int k = 0; k < 16; k++) acc[ph][nt][k] = 0.f;
    }
}

template <typename T, int MW, int NB>
int launch_up2f16(const Up2fParams& pp0, hipStream_t s) {
    typedef Up2fGeo<MW, NB> G;
    Up2fParams pp = pp0;
    Conv16Params& p = pp.c;
    p.tilesX = (p.W + G::UW - 1) / G::UW;
    p.tilesY = (p.H + G::TH - 1) / G::TH;
    p.mblocks = p.f.phase_cout / 32;
    const int64_t tiles = (int64_t)p.N * p.tilesX * p.tilesY * p.mblocks;
    if (tiles > 0x7fffffffLL) return PG_ERR_TOO_LARGE;
    p.total_tiles = (int)tiles;
    auto magic = [&](int d) -> unsigned { return d <= 1 ? 0u : (unsigned)((0x100000000ULL + (unsigned)d - 1) / (unsigned)d); };
    const int dmax = std::max(std::max(p.tilesX, p.tilesY), p.mblocks);
    if (tiles * dmax >= 0x100000000LL) return PG_ERR_TOO_LARGE;
    p.m_tilesX = magic(p.tilesX); p.m_tilesY = magic(p.tilesY); p.m_mblocks = magic(p.mblocks); p.m_ksplit = 0;
    const int64_t resident = (int64_t)num_cu() * G::WG_PER_CU;        // persistent: WG_PER_CU workgroups per CU
    const int64_t blocks = tiles < resident ? tiles : resident;
    auto kern = conv2d_up2f16<T, MW, NB>;
    static PerDeviceOnce lds_attr;
    const hipError_t e = lds_attr.run([&] { return hipFuncSetAttribute((const void*)kern, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024 / G::WG_PER_CU); });
    if (e != hipSuccess) return (int)e;
    hipLaunchKernelGGL(kern, dim3((unsigned)blocks), dim3(G::NTHREADS), G::LDS_BYTES, s, pp);
    return launch_status();
}

// ------------------------------------------------------------------------------------------------------------------------
// The ping-pong form (round 5, second half).  Ablations of the kernel above (tools/up2f_probe.py): of a 4-chunk tile's 45 k cycles 24 k are the
// epilogue -- vector work (x filter, activation, packing: ~10 instructions per value, 128 values per lane) during which the matrix pipe AND the
// load stream stand still, because every wave of the workgroup is in it at the same time (the loaders may run two chunks ahead, no further).  Two
// workgroups per CU did not fix it (twice the weight traffic, one chunk of run-ahead each: slower); two accumulator sets per wave, or one set beside
// any control flow, do not fit the 168 registers of three waves per SIMD (the allocator kept whole 16-register tuples in scratch).
// Here a workgroup is EIGHT waves -- two per SIMD, 256 registers each -- in TWO GROUPS of four (one wave per SIMD each) that take turns by tile:
// group T & 1 multiplies tile T (alone: one wave per SIMD with its operands prefetched keeps the matrix pipe as busy as two did) while the other
// group SERVES: it writes out ITS previous tile, one block of 8 values per lane at a time (EB blocks per K chunk), and issues the LDS-DMA requests of
// the chunk three ahead -- a bf16 MFMA stream and a vector stream from different waves of a SIMD co-issue (tools/probes/mfma_valu_coexec_bf16.hip).
// The chunk stream never stops: one sequence of tiles, 8 rows x 32 lanes x (4 phases x 32 couts) each, through FOUR 32 KB staging buffers; all eight
// waves meet at one barrier per chunk.  Chunk c is requested in step c - 3 by whoever serves then and must have landed before barrier c: the requester
// (it may be multiplying by then) waits with a count of what IT has issued since -- vector memory operations return in order.
struct Up2pGeo {
    static constexpr int GW = 4, NTHREADS = 2 * GW * 64, LT = GW * 64;      // waves per group; threads; requesting threads
    static constexpr int TH = 8, LW = 32, UW = 30;
    static constexpr int T = 6, KC = 16, NBUF = 4, NSIDE = 4, AHEAD = NBUF - 1;
    static constexpr int NSTEP = 18;                       // weight fragments with a non-zero tap: per ty, tx = 1 x phases 0..3, then tx = 0 x phases 0, 2
    static constexpr int IH_T = TH + 2, IW_T = LW + 1;
    static constexpr int NPIX = IH_T * IW_T;
    static constexpr int NGRP = (NPIX + 31) / 32;          // 1 KB groups of 32 pixels x 2 k-halves (see Up2fGeo)
    static constexpr int NXS = NGRP * 64;
    static constexpr int NXS_PAD = (NXS + LT - 1) / LT * LT;
    static constexpr int NWS = NSTEP * 2 * 32;
    static constexpr int NWS_PAD = (NWS + LT - 1) / LT * LT;
    static constexpr int LDS_BUF = NXS_PAD + NWS_PAD;
    static constexpr int NREQ_X = NXS_PAD / LT, NREQ = LDS_BUF / LT;
    static constexpr int NSIDE_REQ = 5;                    // a tile's side loads per requesting wave: scale or bias, 4 noise planes
    static constexpr int EPS = 64, NOISE = 4 * TH * LW;
    static constexpr int EP_FLOATS = 2 * EPS + NOISE;
    static constexpr int NBLK = 16;                        // epilogue blocks per tile and wave: (position row nt, output-row parity a, register group g)
    static constexpr size_t LDS_BYTES = (size_t)NBUF * LDS_BUF * 16 + (size_t)NSIDE * EP_FLOATS * 4 + 256;
    static_assert(LDS_BYTES <= 160 * 1024, "LDS budget");
    static_assert(TH * LW == LT, "one noise sample per requesting thread and phase");
};

template <int V> struct int_c { static constexpr int value = V; };

// s_waitcnt vmcnt(n) for a run-time n: the largest available immediate <= n (waiting for fewer operations to remain is always safe)
__device__ __forceinline__ void vm_wait_dyn(int n) {
    if (n >= 48) vm_wait<48>();
    else if (n >= 40) vm_wait<40>();
    else if (n >= 32) vm_wait<32>();
    else if (n >= 28) vm_wait<28>();
    else if (n >= 24) vm_wait<24>();
    else if (n >= 20) vm_wait<20>();
    else if (n >= 16) vm_wait<16>();
    else if (n >= 13) vm_wait<13>();
    else if (n >= 12) vm_wait<12>();
    else if (n >= 10) vm_wait<10>();
    else if (n >= 8) vm_wait<8>();
    else if (n >= 6) vm_wait<6>();
    else if (n >= 4) vm_wait<4>();
    else if (n >= 2) vm_wait<2>();
    else vm_wait<0>();
}

template <typename T>
__global__ __launch_bounds__(Up2pGeo::NTHREADS, 2) void conv2d_up2f16p(Up2fParams pp) {
    const Conv16Params& p = pp.c;
    typedef Up2pGeo G;
    typedef Half16<T> HT;
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    typedef const __attribute__((address_space(3))) i32x4* lds_v4;
    typedef const __attribute__((address_space(3))) f32x4* lds_f4;
    typedef const __attribute__((address_space(3))) float* lds_f;

    const int t = threadIdx.x, lane = t & 63;
    const int wave = __builtin_amdgcn_readfirstlane(t >> 6);
    const int grp = wave >> 2, wr = wave & 3;                              // group; position rows 2 wr, 2 wr + 1 of a tile / requesting wave index
    const int half = lane >> 5, l31 = lane & 31;
    const int total = p.total_tiles;
    const int q8 = total >> 3, r8 = total & 7;
    const int nchunks = p.Cin / G::KC;
    const int pc = p.f.phase_cout;
    const int dbg_ = PG_CONV16_STAMPS ? p.dbg : 0;      // dev ablations (diagnostic build only; results wrong by design): 1 no stores, 4 no MFMA, 8 no epilogue, 128 no halo DMA

    auto decode = [&](int tile, int& n, int& q0, int& r0, int& m0) __attribute__((always_inline)) {
        const int xcd = tile & 7;
        unsigned L = (unsigned)((xcd < r8 ? xcd * (q8 + 1) : r8 * (q8 + 1) + (xcd - r8) * q8) + (tile >> 3));
        unsigned q = div_magic(L, p.m_mblocks); const int mb = (int)(L - q * p.mblocks); L = q;
        q = div_magic(L, p.m_tilesX); const int tx = (int)(L - q * p.tilesX); L = q;
        q = div_magic(L, p.m_tilesY); const int ty = (int)(L - q * p.tilesY);
        n = (int)q; q0 = ty * G::TH; r0 = tx * G::UW - 1; m0 = mb * 32;
    };

    // ---------------------------------------------------------------- serving role, part 1: the chunk requests
    const unsigned smem_b = __builtin_amdgcn_readfirstlane(lds_offset(smem));
    const unsigned side_b = smem_b + (unsigned)(G::NBUF * G::LDS_BUF) * 16u;
    const unsigned dump_b = side_b + (unsigned)(G::NSIDE * G::EP_FLOATS) * 4u;
    const i32x4 xrsrc = make_rsrc(p.x, (int64_t)p.N * p.H * p.W * p.xC * 2);
    const i32x4 wrsrc = make_rsrc(p.wp, p.w_bytes);
    const int lt = wr * 64 + lane;
    unsigned rel[G::NREQ], hyx[G::NREQ_X];
#pragma unroll
    for (int j = 0; j < G::NREQ; j++) {
        const int s = j * G::LT + lt;
        rel[j] = SENTINEL;
        if (j < G::NREQ_X) {
            hyx[j] = 0x4000u;
            const int c = (s >> 5) & 1, qh = (s >> 6) * 32 + (s & 31);
            const int hy = qh / G::IW_T, hx = qh % G::IW_T;
            if (qh < G::NPIX && s < G::NXS) {
                rel[j] = (unsigned)((hy * p.W + hx) * p.xC + c * 8) * 2u;
                hyx[j] = (unsigned)hy | ((unsigned)hx << 16);
            }
        } else {
            const int e = s - G::NXS_PAD;                                  // [step i][k-half][32 couts]
            const int i = e >> 6, kh = (e >> 5) & 1, cc = e & 31;
            const int ty = i / 6, r = i % 6, tx = r < 4 ? 1 : 0, ph = r < 4 ? r : (r - 4) * 2;
            if (e < G::NWS) rel[j] = (unsigned)(((ty * 2 + tx) * 2 + kh) * p.CoutP + ph * pc + cc) * 16u;
        }
    }
    const bool side_scale = wr == 0, side_bias = wr == 1;
    const i32x4 sbrsrc = side_scale ? make_rsrc(p.f.out_scale, p.f.out_scale ? (int64_t)p.N * pc * 4 : 0)
                                    : make_rsrc(p.f.bias, (p.f.bias && side_bias) ? (int64_t)pc * 4 : 0);
    const i32x4 nrsrc = make_rsrc(p.f.noise, p.f.noise ? ((int64_t)(p.N - 1) * p.f.noise_batch_stride + 3 * p.f.noise_phase_stride + (int64_t)p.H * p.W) * 4 : 0);

    int vm_issued = 0;                                                     // vector memory operations this wave has issued (requests + stores), and its value
    int mark0 = 0, mark1 = 0, mark2 = 0, mark3 = 0;                        // right after the requests of the chunk that sits in staging buffer 0 .. 3
    unsigned voff[G::NREQ_X];
    unsigned w_soff = 0, x_soff0 = 0;
    // request chunk (tile index ti of this workgroup, chunk ck) into staging buffer `buf`
    auto issue_chunk = [&](int ti, int ck, int buf) __attribute__((always_inline)) {
        if (ck == 0) {
            int n, q0, r0, m0;
            decode((int)blockIdx.x + ti * (int)gridDim.x, n, q0, r0, m0);
            const int ty0 = q0 - 1, tx0 = r0 - 1;
            const unsigned org = (unsigned)((ty0 * p.W + tx0) * p.xC * 2);
            const bool interior = ty0 >= 0 && tx0 >= 0 && ty0 + G::IH_T <= p.H && tx0 + G::IW_T <= p.W;
#pragma unroll
            for (int j = 0; j < G::NREQ_X; j++) {
                if (interior) {
                    voff[j] = org + rel[j];
                } else {
                    const unsigned gy = (unsigned)(ty0 + (int)(hyx[j] & 0xffffu)), gx = (unsigned)(tx0 + (int)(hyx[j] >> 16));
                    voff[j] = (gy < (unsigned)p.H && gx < (unsigned)p.W) ? org + rel[j] : SENTINEL;
                }
            }
            x_soff0 = (unsigned)((int64_t)n * p.H * p.W * p.xC * 2);
            w_soff = (unsigned)(((int64_t)n * p.w_nstride + (int64_t)m0 * 8) * 2);
            // per-tile side loads (tile ti's into side buffer ti & 3: its epilogue runs during tile ti + 1's K loop)
            const unsigned sb_ = side_b + (unsigned)((ti & (G::NSIDE - 1)) * G::EP_FLOATS) * 4u;
            const bool live = (side_scale || side_bias) && lane < 32 && m0 + lane < pc;
            dma4(sbrsrc, (side_scale || side_bias) ? sb_ + (unsigned)(wr * 64) * 4u : dump_b, live ? (unsigned)lane * 4u : SENTINEL,
                 (unsigned)(m0 + (side_scale ? n * pc : 0)) * 4u);
            const int nq = q0 + lt / G::LW, nr = r0 + lt % G::LW;
            const unsigned nvo = (nq < p.H && nr >= 0 && nr < p.W) ? (unsigned)(nq * p.W + nr) * 4u : SENTINEL;
#pragma unroll
            for (int ph = 0; ph < 4; ph++)
                dma4(nrsrc, sb_ + (unsigned)(2 * G::EPS + ph * (G::TH * G::LW) + wr * 64) * 4u, nvo,
                     (unsigned)(n * p.f.noise_batch_stride + ph * p.f.noise_phase_stride) * 4u);
            vm_issued += G::NSIDE_REQ;
        }
        const unsigned x_soff = x_soff0 + (unsigned)(ck * G::KC) * 2u;
        const unsigned wk_soff = w_soff + (unsigned)(ck * G::T * 2) * (unsigned)p.CoutP * 16u;
        const unsigned buf_b = smem_b + (unsigned)(buf * G::LDS_BUF) * 16u;
#pragma unroll
        for (int j = 0; j < G::NREQ; j++) {
            const bool is_w = j >= G::NREQ_X;
            dma16(is_w ? wrsrc : xrsrc, buf_b + (unsigned)(j * G::LT + wr * 64) * 16u, is_w ? rel[j] : ((dbg_ & 128) ? SENTINEL : voff[j]), is_w ? wk_soff : x_soff);
        }
        vm_issued += G::NREQ;
        mark0 = buf == 0 ? vm_issued : mark0; mark1 = buf == 1 ? vm_issued : mark1;
        mark2 = buf == 2 ? vm_issued : mark2; mark3 = buf == 3 ? vm_issued : mark3;
    };

    // ---------------------------------------------------------------- multiplying role
    const __amdgpu_buffer_rsrc_t yrsrc = __builtin_amdgcn_make_buffer_rsrc(p.y, 0, (int)(p.y_bytes > 0x7fffffffLL ? 0x7fffffffLL : p.y_bytes), 0x00020000);
    f32x16 acc[4][2];                                                      // [phase 2a + b][position row nt]
    const int q_lane = (2 * wr) * G::IW_T + l31;                           // halo pixel of this lane's first position row, tap (0, 0)
    const unsigned a_lane = (unsigned)(G::NXS_PAD + half * 32 + l31) * 16u;

    // One 16-channel chunk: 18 weight fragments (ty, tx, phase), two MFMAs each (the two position rows).  A software pipeline by hand, pinned with
    // sched_barrier: the weight fragments are requested TWO steps ahead (one multiplying wave per SIMD: nobody else covers an exposed LDS round trip),
    // the activation fragments rotate through four register sets (row ty leaves after its last use, row ty + 2 takes its place).
    auto compute_chunk = [&](int buf) __attribute__((always_inline)) {
        const unsigned char* bb = smem + (size_t)buf * G::LDS_BUF * 16 + half * 512;
        const unsigned char* ab = smem + (size_t)buf * G::LDS_BUF * 16 + a_lane;
        auto b_frag = [&](int hr, int tx) __attribute__((always_inline)) {
            const int q = q_lane + hr * G::IW_T + tx;
            return *(lds_v4)(bb + (size_t)(((q >> 5) << 10) + ((q & 31) << 4)));
        };
        auto a_frag = [&](int i) __attribute__((always_inline)) { return *(lds_v4)(ab + (size_t)(i * 64) * 16); };
        i32x4 b1[4], b0[4];
        b1[0] = b_frag(0, 1); b1[1] = b_frag(1, 1);
        i32x4 a0 = a_frag(0), a1 = a_frag(1), a2;
        b0[0] = b_frag(0, 0); b0[1] = b_frag(1, 0);
#pragma unroll
        for (int i = 0; i < G::NSTEP; i++) {
            const int ty = i / 6, r = i % 6, tx = r < 4 ? 1 : 0, ph = r < 4 ? r : (r - 4) * 2;
            if (i + 2 < G::NSTEP) a2 = a_frag(i + 2);
            if (ty < 2 && r == 2) b1[ty + 2] = b_frag(ty + 2, 1);
            if (ty < 2 && r == 4) b0[ty + 2] = b_frag(ty + 2, 0);
            const i32x4 blo = tx ? b1[ty] : b0[ty], bhi = tx ? b1[ty + 1] : b0[ty + 1];
            acc[ph][0] = HT::mma(a0, blo, acc[ph][0]);
            acc[ph][1] = HT::mma(a0, bhi, acc[ph][1]);
            a0 = a1; a1 = a2;
            __builtin_amdgcn_sched_barrier(0);
        }
    };

    // ---------------------------------------------------------------- serving role, part 2: the epilogue of this group's previous tile, in blocks
    //   u = h * (scale * gain) + (bias + noise) * gain,  v = med3(max(u, u * slope), -cl, cl)  (positively homogeneous activations, gain > 0)
    const float gain = p.f.gain;
    const float cl = p.f.clamp >= 0.f ? p.f.clamp : __builtin_inff();
    const float slope = act_slope(p.f.act, p.f.alpha);
    const bool has_scale = p.f.out_scale != nullptr;
    const float ng = (p.f.noise ? p.f.noise_gain : 0.f) * gain;
    const float tg = has_scale ? 1.f : gain;                               // without a per-cout scale the gain rides in the taps
    const float f0 = pp.fir[0] * tg, f1 = pp.fir[1] * tg, f2 = pp.fir[2] * tg, f3 = pp.fir[3] * tg;
    u32x2 pe = {0u, 0u}, po = {0u, 0u};                                    // an even block's packed results, waiting for their exchange partner (the next block)

    // block blk = (nt, a, g) of the finished tile (e_n, e_q0, e_r0, e_m0): 2 x 4 values per lane; the odd blocks also exchange and store (2 stores).
    // ONE body for all sixteen blocks: a switch copies the block's eight accumulator values into plain registers, everything else takes (nt, a, g) as scalars.
    // (Sixteen unrolled bodies x two scale variants were ~30 KB of straight-line code next to the other group's MFMA loop and the request code: the serving
    // waves ran at ~15 cycles per instruction -- instruction fetch, not the vector ALU.)
    auto epi_block = [&](int blk, int e_n, int e_q0, int e_r0, int e_m0, int sidx) __attribute__((always_inline)) {
        const int nt = blk >> 3, a = (blk >> 2) & 1, g = blk & 3;
        float z0[4], z1[4];
        switch (blk) {
#define PG_UP2_BLK(B) case B: { _Pragma("unroll") for (int j = 0; j < 4; j++) { z0[j] = acc[2 * (((B) >> 2) & 1)][(B) >> 3][4 * ((B) & 3) + j]; z1[j] = acc[2 * (((B) >> 2) & 1) + 1][(B) >> 3][4 * ((B) & 3) + j]; } } break;
            PG_UP2_BLK(0) PG_UP2_BLK(1) PG_UP2_BLK(2) PG_UP2_BLK(3) PG_UP2_BLK(4) PG_UP2_BLK(5) PG_UP2_BLK(6) PG_UP2_BLK(7)
            PG_UP2_BLK(8) PG_UP2_BLK(9) PG_UP2_BLK(10) PG_UP2_BLK(11) PG_UP2_BLK(12) PG_UP2_BLK(13) PG_UP2_BLK(14)
            default: { _Pragma("unroll") for (int j = 0; j < 4; j++) { z0[j] = acc[3 - 1][1][12 + j]; z1[j] = acc[3][1][12 + j]; } } break;
#undef PG_UP2_BLK
        }
        // (a DPP read of a register needs two wait states after the vector instruction that wrote it: the copies above are followed by the LDS reads and their wait)
        const unsigned char* side = smem + (size_t)G::NBUF * G::LDS_BUF * 16 + (size_t)sidx * G::EP_FLOATS * 4;
        const int row_l = 2 * wr + nt, q = e_q0 + row_l;
        const float nze = *(lds_f)(side + (size_t)(2 * G::EPS + (2 * a) * (G::TH * G::LW) + row_l * G::LW + l31) * 4) * ng;
        const float nzo = *(lds_f)(side + (size_t)(2 * G::EPS + (2 * a + 1) * (G::TH * G::LW) + row_l * G::LW + l31) * 4) * ng;
        const int r0c = 8 * g + 4 * half;
        const f32x4 bgv = *(lds_f4)(side + (size_t)(G::EPS + r0c) * 4) * gain;
        f32x4 sgv = {gain, gain, gain, gain};
        if (has_scale) sgv = *(lds_f4)(side + (size_t)r0c * 4) * gain;
        float ve[4], vo[4];
#pragma unroll
        for (int j = 0; j < 4; j++) {
            const float z1m = dpp_f<DPP_WAVE_SHR1>(z1[j]), z0p = dpp_f<DPP_WAVE_SHL1>(z0[j]), z1p = dpp_f<DPP_WAVE_SHL1>(z1[j]);
            const float he = fmaf(f3, z1m, fmaf(f2, z0[j], fmaf(f1, z1[j], f0 * z0p)));
            const float ho = fmaf(f3, z0[j], fmaf(f2, z1[j], fmaf(f1, z0p, f0 * z1p)));
            // (taps carry the gain when there is no per-cout scale: sgv == 1 then -- one code path)
            const float ue = fmaf(he, has_scale ? sgv[j] : 1.f, bgv[j] + nze), uo = fmaf(ho, has_scale ? sgv[j] : 1.f, bgv[j] + nzo);
            ve[j] = __builtin_amdgcn_fmed3f(fmaxf(ue, ue * slope), -cl, cl);
            vo[j] = __builtin_amdgcn_fmed3f(fmaxf(uo, uo * slope), -cl, cl);
        }
        u32x2 ce = {HT::pack(ve[0], ve[1]), HT::pack(ve[2], ve[3])}, co2 = {HT::pack(vo[0], vo[1]), HT::pack(vo[2], vo[3])};
        if (g & 1) {
            // lanes 32-63 of group g - 1 <-> lanes 0-31 of group g: 8 consecutive couts of one pixel per lane
#pragma unroll
            for (int d = 0; d < 2; d++) {
                const auto re = __builtin_amdgcn_permlane32_swap(pe[d], ce[d], false, false);
                pe[d] = re[0]; ce[d] = re[1];
                const auto ro = __builtin_amdgcn_permlane32_swap(po[d], co2[d], false, false);
                po[d] = ro[0]; co2[d] = ro[1];
            }
            const int r = e_r0 + l31, oy = 2 * q + a;
            const bool pos_ok = l31 >= 1 && l31 <= G::UW && r < p.W && q < p.H && !(dbg_ & 1);
            const unsigned pix_off = (unsigned)((int64_t)e_n * p.ys[0] + (int64_t)oy * p.ys[2] + (int64_t)(2 * r) * p.ys[3]);
            const int co = e_m0 + 8 * (g - 1 + half);
            const bool ok = pos_ok && co < pc;
            const unsigned se = ok ? (pix_off + (unsigned)co) * 2u : SENTINEL;
            const unsigned so = ok ? (pix_off + (unsigned)p.ys[3] + (unsigned)co) * 2u : SENTINEL;
            __builtin_amdgcn_raw_buffer_store_b128(u32x4{pe[0], pe[1], ce[0], ce[1]}, yrsrc, (int)se, 0, 0);
            __builtin_amdgcn_raw_buffer_store_b128(u32x4{po[0], po[1], co2[0], co2[1]}, yrsrc, (int)so, 0, 0);
            vm_issued += 2;
        } else {
            pe = ce; po = co2;
        }
    };
    auto epi_blocks = [&](int first, int count, int e_n, int e_q0, int e_r0, int e_m0, int sidx) __attribute__((always_inline)) {
#pragma unroll 1
        for (int blk = first; blk < first + count && blk < G::NBLK; blk++) epi_block(blk, e_n, e_q0, e_r0, e_m0, sidx);
    };

    // ---------------------------------------------------------------- the step loop: one barrier per chunk of the workgroup's tile sequence
    const int EB = (G::NBLK + nchunks - 1) / nchunks;                      // blocks per step: all 16 within the other group's K loop
    const int my_tiles = (total - (int)blockIdx.x + (int)gridDim.x - 1) / (int)gridDim.x;
    const int C = my_tiles * nchunks;
    // group 1 serves tile 0's steps: it also fills the pipeline
    int q_ti = 0, q_ck = 0, q_c = 0;                                        // request cursor: (tile index, chunk) of global chunk q_c
    auto issue_cursor = [&]() __attribute__((always_inline)) {
        issue_chunk(q_ti, q_ck, q_c & (G::NBUF - 1));
    };
    auto advance_cursor = [&]() __attribute__((always_inline)) {            // (every wave keeps the cursor, whoever issues)
        q_c++;
        if (++q_ck == nchunks) { q_ck = 0; q_ti++; }
    };
    for (int j = 0; j < G::AHEAD; j++) {
        if (q_c < C) { if (grp == 1) issue_cursor(); advance_cursor(); }
    }
    int ti = 0, ck = 0;                                                     // (tile index, chunk) of step c
    int e_n = 0, e_q0 = 0, e_r0 = 0, e_m0 = 0;
    for (int c = 0; c < C; c++) {
        // chunk c was requested in step max(c - AHEAD, 0) by that step's serving group; if that was this wave, its requests must have landed before the barrier
        {
            const int rc = c >= G::AHEAD ? c - G::AHEAD : 0;
            const int rt = (int)div_magic((unsigned)rc, p.m_ksplit);       // tile index of step rc (m_ksplit: the magic of nchunks)
            if ((1 - (rt & 1)) == grp) {
                const int b = c & (G::NBUF - 1);
                const int mk = b == 0 ? mark0 : (b == 1 ? mark1 : (b == 2 ? mark2 : mark3));
                vm_wait_dyn(vm_issued - mk);
            }
        }
        __builtin_amdgcn_s_barrier();
        if ((ti & 1) == grp) {                                              // this group multiplies tile ti
            if (ck == 0) {
#pragma unroll
                for (int ph = 0; ph < 4; ph++)
#pragma unroll
                    for (int nt = 0; nt < 2; nt++)
#pragma unroll
                        for (int e = 0; e < 16; e++) acc[ph][nt][e] = 0.f;
            }
            if (!(dbg_ & 4)) compute_chunk(c & (G::NBUF - 1));
        } else {                                                           // it serves: its tile ti - 1 goes out, the chunk AHEAD steps on is requested
            const bool pending = ti > 0 && !(dbg_ & 8);
            if (pending && ck == 0) decode((int)blockIdx.x + (ti - 1) * (int)gridDim.x, e_n, e_q0, e_r0, e_m0);
            if (pending && ck * EB < G::NBLK) epi_blocks(ck * EB, EB, e_n, e_q0, e_r0, e_m0, (ti - 1) & (G::NSIDE - 1));
            if (q_c < C) issue_cursor();
        }
        if (q_c < C) advance_cursor();
        if (++ck == nchunks) { ck = 0; ti++; }
    }
    if (((my_tiles - 1) & 1) == grp && !(dbg_ & 8)) {                      // the last tile's epilogue has nobody to hide behind
        decode((int)blockIdx.x + (my_tiles - 1) * (int)gridDim.x, e_n, e_q0, e_r0, e_m0);
        epi_blocks(0, G::NBLK, e_n, e_q0, e_r0, e_m0, (my_tiles - 1) & (G::NSIDE - 1));
    }
}

template <typename T>
int launch_up2f16p(const Up2fParams& pp0, hipStream_t s) {
    typedef Up2pGeo G;
    Up2fParams pp = pp0;
    Conv16Params& p = pp.c;
    p.tilesX = (p.W + G::UW - 1) / G::UW;
    p.tilesY = (p.H + G::TH - 1) / G::TH;
    p.mblocks = p.f.phase_cout / 32;
    const int64_t tiles = (int64_t)p.N * p.tilesX * p.tilesY * p.mblocks;
    if (tiles > 0x7fffffffLL) return PG_ERR_TOO_LARGE;
    p.total_tiles = (int)tiles;
    auto magic = [&](int d) -> unsigned { return d <= 1 ? 0u : (unsigned)((0x100000000ULL + (unsigned)d - 1) / (unsigned)d); };
    const int dmax = std::max(std::max(p.tilesX, p.tilesY), p.mblocks);
    if (tiles * dmax >= 0x100000000LL) return PG_ERR_TOO_LARGE;
    const int nchunks = p.Cin / G::KC;
    const int64_t blocks = tiles < (int64_t)num_cu() ? tiles : (int64_t)num_cu();
    const int64_t steps = ((tiles + blocks - 1) / blocks) * nchunks;       // of one workgroup: exactness bound of the step -> tile division
    if (steps * nchunks >= 0x100000000LL) return PG_ERR_TOO_LARGE;
    p.m_tilesX = magic(p.tilesX); p.m_tilesY = magic(p.tilesY); p.m_mblocks = magic(p.mblocks); p.m_ksplit = magic(nchunks);
    auto kern = conv2d_up2f16p<T>;
    static PerDeviceOnce lds_attr;
    const hipError_t e = lds_attr.run([&] { return hipFuncSetAttribute((const void*)kern, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024); });
    if (e != hipSuccess) return (int)e;
    hipLaunchKernelGGL(kern, dim3((unsigned)blocks), dim3(G::NTHREADS), G::LDS_BYTES, s, pp);
    return launch_status();
}

int launch16_up2f(const Up2fParams& p, int dtype, hipStream_t s);      // conv2d16_inst_up2f.hip

}  // namespace pgconv16
