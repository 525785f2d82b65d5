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
#include "pg_act.h"

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
    int tilesX, tilesY, mblocks;
    int in_xform;          // prologue bias/act/gain/clamp stage on
    pg_conv2d_fusion f;
};

constexpr int TH = 8, TW = 32;   // output tile of one workgroup (rows x cols)

template <int KH, int KW, int S, int BM, int KC>
struct Geo {
    static constexpr int T = KH * KW;
    static constexpr int IH_T = (TH - 1) * S + KH;
    static constexpr int IW_T = (TW - 1) * S + KW;
    static constexpr int PLANE = IH_T * IW_T;
    static constexpr int NX = KC * PLANE;                 // staged input floats per chunk
    static constexpr int XPT = (NX + 255) / 256;          // per thread
    static constexpr int NW4 = KC * T * BM / 4;           // staged weight float4s per chunk
    static constexpr int WPT = (NW4 + 255) / 256;
    static constexpr int MT = BM / 32;
    static constexpr int NT = 2;
    static constexpr int LDS_X = NX;                      // floats
    static constexpr int LDS_W = KC * T * BM;
    static constexpr int LDS_BUF = LDS_X + LDS_W;          // one staging buffer (floats)
    static constexpr size_t LDS_BYTES = (size_t)2 * LDS_BUF * 4;   // double buffered
};

// The fused activations are linear / relu / lrelu (everything the synthesis path uses): one select,
// v > 0 ? v : v * slope.  Other activations are rejected by pg_conv2d_forward (PG_ERR_UNSUPPORTED); callers
// run bias_act separately for those.
__device__ __forceinline__ float act_slope(int act, float alpha) { return act == PG_ACT_LINEAR ? 1.f : (act == PG_ACT_RELU ? 0.f : alpha); }

template <int KH, int KW, int S, int BM, int KC>
__global__ __launch_bounds__(256, 4) void conv2d_mfma(ConvParams p) {
    typedef Geo<KH, KW, S, BM, KC> G;
    extern __shared__ __attribute__((aligned(16))) float smem[];
    // two staging buffers, each { xs[KC][IH_T][IW_T], ws[KC][T][BM] }, then the per-channel constants
    float* cs = smem + 2 * G::LDS_BUF;          // [cin_loop] prologue scale  (1 where off)
    const int cin_loop = ((p.Cin + KC - 1) / KC) * KC;
    float* cb = cs + cin_loop;                  // [cin_loop] prologue bias   (0 where off)

    // ---- workgroup -> (n, tile, m-block), XCD-aware: each XCD gets a contiguous range of
    // logical tiles so neighbouring tiles / m-blocks of one tile share that XCD's L2.
    const int total = gridDim.x, id = blockIdx.x;
    const int q = total >> 3, r8 = total & 7, xcd = id & 7;
    int L = (xcd < r8 ? xcd * (q + 1) : r8 * (q + 1) + (xcd - r8) * q) + (id >> 3);
    const int mb = L % p.mblocks; L /= p.mblocks;
    const int tx = L % p.tilesX; L /= p.tilesX;
    const int ty = L % p.tilesY;
    const int n = L / p.tilesY;

    const int t = threadIdx.x, lane = t & 63, wave = t >> 6;
    const int half = lane >> 5, l31 = lane & 31;
    const int oy0 = ty * TH, ox0 = tx * TW, m0 = mb * BM;
    const int HW = p.H * p.W;

    // ---- per-channel prologue constants -> LDS (so the staging loop has no dependent global loads)
    {
        const float* in_scale = p.f.in_scale ? p.f.in_scale + (int64_t)n * p.Cin : nullptr;
        for (int c = t; c < cin_loop; c += 256) {
            const bool ok = c < p.Cin;
            cs[c] = (in_scale && ok) ? in_scale[c] : 1.f;
            cb[c] = (p.f.in_bias && ok) ? p.f.in_bias[c] : 0.f;
        }
    }

    // ---- per-thread staging map of the input halo tile (independent of the chunk): BYTE offsets into
    // image n, or a sentinel >= 2^31 for halo elements outside the image.  The tile is fetched with raw
    // buffer loads whose hardware range check returns 0 for the sentinel AND for channels >= Cin, so zero
    // padding costs neither a branch nor a select.
    unsigned xoff[G::XPT];
    unsigned xok = 0;
#pragma unroll
    for (int i = 0; i < G::XPT; i++) {
        const int e = t + 256 * i;
        const int c = e / G::PLANE, rem = e % G::PLANE;
        const int rr = rem / G::IW_T, cc = rem % G::IW_T;
        const int gy = oy0 * S - p.pad_y + rr, gx = ox0 * S - p.pad_x + cc;
        const bool ok = e < G::NX && gy >= 0 && gy < p.H && gx >= 0 && gx < p.W;
        xoff[i] = ok ? (unsigned)(c * HW + gy * p.W + gx) * 4u : 0x80000000u;
        xok |= ok ? (1u << i) : 0u;
    }
    const float* xn = p.x + (int64_t)n * p.Cin * HW;
    const __amdgpu_buffer_rsrc_t xrsrc = __builtin_amdgcn_make_buffer_rsrc((void*)xn, 0, p.Cin * HW * 4, 0x00020000);

    float xr[G::XPT];
    f32x4 wr[G::WPT];

    // Unconditional loads from always-valid addresses + selects: no exec-mask branches, so the
    // compiler issues the whole batch back to back and waits once.
    auto load_chunk = [&](int c0) {
        const int soff = c0 * HW * 4;                       // wave-uniform chunk offset
#pragma unroll
        for (int i = 0; i < G::XPT; i++)
            xr[i] = __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(xrsrc, xoff[i], soff, 0));
        const float* wb = p.wp + (int64_t)c0 * G::T * p.CoutP + m0;
#pragma unroll
        for (int i = 0; i < G::WPT; i++) {
            int e4 = t + 256 * i;
            if (G::NW4 % 256 != 0 && e4 >= G::NW4) e4 = G::NW4 - 1;     // clamp: harmless duplicate read
            const int row = (e4 * 4) / BM, col = (e4 * 4) % BM;
            wr[i] = *(const f32x4*)(wb + (int64_t)row * p.CoutP + col);
        }
    };

    const float in_slope = act_slope(p.f.in_act, p.f.in_alpha);
    const float in_cl = p.f.in_clamp >= 0.f ? p.f.in_clamp : __builtin_inff();

    auto store_chunk = [&](int c0, int buf) {
        float* xs = smem + buf * G::LDS_BUF;
        float* ws = xs + G::LDS_X;
        if (!p.in_xform) {
#pragma unroll
            for (int i = 0; i < G::XPT; i++) {
                const int e = t + 256 * i;
                const float v = xr[i] * cs[c0 + e / G::PLANE];             // padding is 0 and stays 0
                if (G::NX % 256 == 0 || e < G::NX) xs[e] = v;
            }
        } else {
#pragma unroll
            for (int i = 0; i < G::XPT; i++) {
                const int e = t + 256 * i;
                const int c = c0 + e / G::PLANE;
                float v = xr[i] * cs[c] + cb[c];
                v = v > 0.f ? v : v * in_slope;
                v = fminf(fmaxf(v * p.f.in_gain, -in_cl), in_cl);
                const bool ok = ((xok >> i) & 1u) && c < p.Cin;            // zero padding stays zero (conv pads AFTER the activation)
                if (G::NX % 256 == 0 || e < G::NX) xs[e] = ok ? v : 0.f;
            }
        }
#pragma unroll
        for (int i = 0; i < G::WPT; i++) {
            const int e4 = t + 256 * i;
            if (G::NW4 % 256 == 0 || e4 < G::NW4) *(f32x4*)(ws + e4 * 4) = wr[i];
        }
    };

    f32x16 acc[G::MT][G::NT];
#pragma unroll
    for (int mt = 0; mt < G::MT; mt++)
#pragma unroll
        for (int nt = 0; nt < G::NT; nt++)
#pragma unroll
            for (int k = 0; k < 16; k++) acc[mt][nt][k] = 0.f;

    // operand base addresses inside LDS buffer 0 (floats)
    const float* a_base0 = smem + G::LDS_X + half * (G::T * BM) + l31;
    const float* b_base0 = smem + half * G::PLANE + (wave * 2 * S) * G::IW_T + l31 * S;

    // Pipeline (one barrier per chunk): while chunk k is multiplied out of buffer k&1, chunk k+1 -- fetched
    // during the previous iteration -- is written to the other buffer, and chunk k+2 is in flight to registers.
    const int nchunks = cin_loop / KC;
    load_chunk(0);
    __syncthreads();                                       // cs / cb visible
    store_chunk(0, 0);
    if (nchunks > 1) load_chunk(KC);
    __syncthreads();

    for (int k = 0; k < nchunks; k++) {
        if (k + 1 < nchunks) store_chunk((k + 1) * KC, (k + 1) & 1);   // that buffer was last read in iteration k-1 (barrier below)
        if (k + 2 < nchunks) load_chunk((k + 2) * KC);                 // lands during the MFMAs
        const float* a_base = a_base0 + (k & 1) * G::LDS_BUF;
        const float* b_base = b_base0 + (k & 1) * G::LDS_BUF;
#pragma unroll
        for (int cp = 0; cp < KC / 2; cp++) {
#pragma unroll
            for (int ky = 0; ky < KH; ky++) {
#pragma unroll
                for (int kx = 0; kx < KW; kx++) {
                    float a[G::MT], b[G::NT];
#pragma unroll
                    for (int mt = 0; mt < G::MT; mt++) a[mt] = a_base[((2 * cp) * G::T + ky * KW + kx) * BM + mt * 32];
#pragma unroll
                    for (int nt = 0; nt < G::NT; nt++) b[nt] = b_base[(2 * cp) * G::PLANE + (nt * S + ky) * G::IW_T + kx];
#pragma unroll
                    for (int mt = 0; mt < G::MT; mt++)
#pragma unroll
                        for (int nt = 0; nt < G::NT; nt++)
                            acc[mt][nt] = __builtin_amdgcn_mfma_f32_32x32x2f32(a[mt], b[nt], acc[mt][nt], 0, 0, 0);
                }
            }
        }
        __syncthreads();
    }

    // ---- epilogue.  Per-cout constants go through LDS (free now); every load below is unconditional
    // from a clamped, always-valid address, so nothing serialises behind a branch.
    float* ep_scale = smem;            // [BM]
    float* ep_bias = smem + BM;        // [BM]
    if (t < BM) {
        const int co = m0 + t;
        const bool ok = co < p.Cout;
        const int cc = ok ? co : 0;
        const float sc = p.f.out_scale ? p.f.out_scale[(int64_t)n * p.Cout + cc] : 1.f;
        const float bi = p.f.bias ? p.f.bias[cc] : 0.f;
        ep_scale[t] = ok ? sc : 0.f;
        ep_bias[t] = ok ? bi : 0.f;
    }
    __syncthreads();

    const int ox = ox0 + l31;
    const int oxc = ox < p.OW ? ox : p.OW - 1;
    const float gain = p.f.gain;
    const float cl = p.f.clamp >= 0.f ? p.f.clamp : __builtin_inff();
    const float slope = act_slope(p.f.act, p.f.alpha);
#pragma unroll
    for (int nt = 0; nt < G::NT; nt++) {
        const int oy = oy0 + wave * 2 + nt;
        const bool pix_ok = oy < p.OH && ox < p.OW;
        const int oyc = oy < p.OH ? oy : p.OH - 1;
        float nz = 0.f;
        if (p.f.noise) nz = p.f.noise[(int)(n * p.f.noise_batch_stride) + oyc * p.OW + oxc] * p.f.noise_gain;
        const int pix_off = (int)((int64_t)n * p.ys[0] + (int64_t)(oyc * p.osy + p.ooy) * p.ys[2] + (int64_t)(oxc * p.osx + p.oox) * p.ys[3]);
        const int cstride = (int)p.ys[1];
#pragma unroll
        for (int mt = 0; mt < G::MT; mt++) {
#pragma unroll
            for (int kq = 0; kq < 4; kq++) {                 // 4 consecutive couts at a time keeps the live state small
                const int row0 = mt * 32 + 8 * kq + 4 * half;
                const f32x4 sc4 = *(const f32x4*)(ep_scale + row0);
                const f32x4 bi4 = *(const f32x4*)(ep_bias + row0);
                int off[4];
                float rv[4];
#pragma unroll
                for (int j = 0; j < 4; j++) {
                    const int co = m0 + row0 + j;
                    off[j] = pix_off + (co < p.Cout ? co : p.Cout - 1) * cstride;
                    rv[j] = 0.f;
                }
                if (p.f.residual) {
#pragma unroll
                    for (int j = 0; j < 4; j++) rv[j] = p.f.residual[off[j]];
                }
#pragma unroll
                for (int j = 0; j < 4; j++) {
                    float v = acc[mt][nt][4 * kq + j] * sc4[j] + nz + bi4[j];
                    v = v > 0.f ? v : v * slope;
                    v = fminf(fmaxf(v * gain, -cl), cl) + rv[j];
                    if (pix_ok && m0 + row0 + j < p.Cout) p.y[off[j]] = v;
                }
            }
        }
    }
}

template <int KH, int KW, int S, int BM, int KC>
int launch_conv(const ConvParams& p0, hipStream_t s) {
    typedef Geo<KH, KW, S, BM, KC> G;
    ConvParams p = p0;
    p.tilesX = (p.OW + TW - 1) / TW;
    p.tilesY = (p.OH + TH - 1) / TH;
    p.mblocks = p.CoutP / BM;
    const int64_t blocks = (int64_t)p.N * p.tilesX * p.tilesY * p.mblocks;
    if (blocks > 0x7fffffffLL) return PG_ERR_TOO_LARGE;
    static bool attr_set = false;
    if (!attr_set) {
        hipError_t e = hipFuncSetAttribute((const void*)conv2d_mfma<KH, KW, S, BM, KC>, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
        if (e != hipSuccess) return (int)e;
        attr_set = true;
    }
    const int cin_loop = ((p.Cin + KC - 1) / KC) * KC;
    const size_t lds = G::LDS_BYTES + (size_t)2 * cin_loop * sizeof(float);
    if (lds > 160 * 1024) return PG_ERR_UNSUPPORTED;
    hipLaunchKernelGGL((conv2d_mfma<KH, KW, S, BM, KC>), dim3((unsigned)blocks), dim3(256), lds, s, p);
    return launch_status();
}

template <int KH, int KW, int S, int KC>
int launch_bm(const ConvParams& p, hipStream_t s) {
    if (p.CoutP % 64 == 0) return launch_conv<KH, KW, S, 64, KC>(p, s);
    return launch_conv<KH, KW, S, 32, KC>(p, s);
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

}  // namespace pgconv
