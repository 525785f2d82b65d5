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
    static constexpr size_t LDS_BYTES = (size_t)(LDS_X + LDS_W) * 4;
};

template <int KH, int KW, int S, int BM, int KC>
__global__ __launch_bounds__(256, 2) void conv2d_mfma(ConvParams p) {
    typedef Geo<KH, KW, S, BM, KC> G;
    extern __shared__ __attribute__((aligned(16))) float smem[];
    float* xs = smem;                 // [KC][IH_T][IW_T]
    float* ws = smem + G::LDS_X;      // [KC][T][BM]

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

    // ---- per-thread staging map of the input halo tile (independent of the chunk)
    int xoff[G::XPT];
    unsigned xok = 0;
#pragma unroll
    for (int i = 0; i < G::XPT; i++) {
        const int e = t + 256 * i;
        const int c = e / G::PLANE, rem = e % G::PLANE;
        const int rr = rem / G::IW_T, cc = rem % G::IW_T;
        const int gy = oy0 * S - p.pad_y + rr, gx = ox0 * S - p.pad_x + cc;
        const bool ok = e < G::NX && gy >= 0 && gy < p.H && gx >= 0 && gx < p.W;
        xoff[i] = ok ? c * HW + gy * p.W + gx : 0;
        xok |= ok ? (1u << i) : 0u;
    }
    const float* xn = p.x + (int64_t)n * p.Cin * HW;
    const float* in_scale = p.f.in_scale ? p.f.in_scale + (int64_t)n * p.Cin : nullptr;

    float xr[G::XPT];
    f32x4 wr[G::WPT];

    auto load_chunk = [&](int c0) {
        const float* xb = xn + (int64_t)c0 * HW;
#pragma unroll
        for (int i = 0; i < G::XPT; i++) {
            const int c = c0 + (t + 256 * i) / G::PLANE;
            const bool ok = ((xok >> i) & 1u) && c < p.Cin;
            xr[i] = ok ? xb[xoff[i]] : 0.f;
        }
        const float* wb = p.wp + (int64_t)c0 * G::T * p.CoutP + m0;
#pragma unroll
        for (int i = 0; i < G::WPT; i++) {
            const int e4 = t + 256 * i;
            if (G::NW4 % 256 == 0 || e4 < G::NW4) {
                const int row = (e4 * 4) / BM, col = (e4 * 4) % BM;
                wr[i] = *(const f32x4*)(wb + (int64_t)row * p.CoutP + col);
            }
        }
    };

    auto store_chunk = [&](int c0) {
#pragma unroll
        for (int i = 0; i < G::XPT; i++) {
            const int e = t + 256 * i;
            float v = xr[i];
            const int c = c0 + e / G::PLANE;
            if (((xok >> i) & 1u) && c < p.Cin) {          // zero padding stays zero
                if (in_scale) v *= in_scale[c];
                if (p.in_xform) {
                    if (p.f.in_bias) v += p.f.in_bias[c];
                    v = act_forward(p.f.in_act, v, p.f.in_alpha) * p.f.in_gain;
                    if (p.f.in_clamp >= 0.f) v = clampf(v, p.f.in_clamp);
                }
            }
            if (G::NX % 256 == 0 || e < G::NX) xs[e] = v;
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

    // operand base addresses inside LDS (floats)
    const float* a_base = ws + half * (G::T * BM) + l31;
    const float* b_base = xs + half * G::PLANE + (wave * 2 * S) * G::IW_T + l31 * S;

    const int cin_loop = ((p.Cin + KC - 1) / KC) * KC;
    load_chunk(0);
    store_chunk(0);
    __syncthreads();

    for (int c0 = 0; c0 < cin_loop; c0 += KC) {
        const bool more = c0 + KC < cin_loop;
        if (more) load_chunk(c0 + KC);                     // in flight during the MFMAs below

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
        __syncthreads();                                   // everyone done reading this chunk
        if (more) {
            store_chunk(c0 + KC);
            __syncthreads();
        }
    }

    // ---- epilogue: D layout col = lane&31 (pixel), row = (reg&3) + 8*(reg>>2) + 4*(lane>>5) (cout)
    const int ox = ox0 + l31;
    const float gain = p.f.gain;
    const float* out_scale = p.f.out_scale ? p.f.out_scale + (int64_t)n * p.Cout : nullptr;
#pragma unroll
    for (int nt = 0; nt < G::NT; nt++) {
        const int oy = oy0 + wave * 2 + nt;
        const bool pix_ok = oy < p.OH && ox < p.OW;
        float nz = 0.f;
        if (p.f.noise && pix_ok) nz = p.f.noise[(int64_t)n * p.f.noise_batch_stride + (int64_t)oy * p.OW + ox] * p.f.noise_gain;
        const int64_t pix_off = (int64_t)n * p.ys[0] + (int64_t)(oy * p.osy + p.ooy) * p.ys[2] + (int64_t)(ox * p.osx + p.oox) * p.ys[3];
#pragma unroll
        for (int mt = 0; mt < G::MT; mt++) {
#pragma unroll
            for (int k = 0; k < 16; k++) {
                const int co = m0 + mt * 32 + (k & 3) + 8 * (k >> 2) + 4 * half;
                if (pix_ok && co < p.Cout) {
                    float v = acc[mt][nt][k];
                    if (out_scale) v *= out_scale[co];
                    v += nz;
                    if (p.f.bias) v += p.f.bias[co];
                    v = act_forward(p.f.act, v, p.f.alpha) * gain;
                    if (p.f.clamp >= 0.f) v = clampf(v, p.f.clamp);
                    const int64_t o = pix_off + (int64_t)co * p.ys[1];
                    if (p.f.residual) v += p.f.residual[o];
                    p.y[o] = v;
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
        hipError_t e = hipFuncSetAttribute((const void*)conv2d_mfma<KH, KW, S, BM, KC>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)G::LDS_BYTES);
        if (e != hipSuccess) return (int)e;
        attr_set = true;
    }
    hipLaunchKernelGGL((conv2d_mfma<KH, KW, S, BM, KC>), dim3((unsigned)blocks), dim3(256), G::LDS_BYTES, s, p);
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
