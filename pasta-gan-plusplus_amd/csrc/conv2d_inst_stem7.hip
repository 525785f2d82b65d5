// The 7x7 three-channel stem convolution on the bf16 matrix pipe: instantiation + C ABI (own translation unit: see conv2d_kernel.h on build time).
#include "conv2d_stem7x3.h"

/* Round 6 -- conv2d(x [N,3,H,W], w [Cout,3,7,7] * scale, stride 1, padding 3) with the fused bias / activation / gain / clamp of pg_conv2d_forward
 * (networks.py:170-179 around conv2d_resample.py:145-147), float32 operands as exact sums of three bf16 values, six plane products per float32 product on
 * v_mfma_f32_32x32x16_bf16, float32 accumulation: float32-class results (csrc/conv2d_stem7x3.h).  `packed` = pg_conv2d_stem7x3_pack_weight of the OIHW
 * kernel (pg_conv2d_stem7x3_packed_size(Cout) BYTES, once per weight version; flip_hw as pg_conv2d_pack_weight).  Fusion stages other than bias / act in
 * {linear, relu, lrelu} / gain / clamp are declined with PG_ERR_UNSUPPORTED (callers then use pg_conv2d_forward). */
PG_EXPORT int64_t pg_conv2d_stem7x3_packed_size(int Cout) {
    return Cout > 0 ? pgconv::stem7x3_packed_bytes(Cout) : 0;
}

PG_EXPORT int pg_conv2d_stem7x3_pack_weight(const float* w, void* packed, int Cout, float scale, int flip_hw, void* stream) {
    if (!w || !packed || Cout <= 0 || (((uintptr_t)packed) & 15) != 0) return PG_ERR_INVALID_ARG;
    const int mtiles = (Cout + 63) / 64 * 2;
    const int total = mtiles * pgconv::S7_KS * 64 * 8;
    int blocks = (total + 255) / 256;
    if (blocks > pg::max_stream_blocks()) blocks = pg::max_stream_blocks();
    hipLaunchKernelGGL(pgconv::stem7x3_pack_kernel, dim3((unsigned)blocks), dim3(256), 0, (hipStream_t)stream, w, (unsigned short*)packed, Cout, mtiles, scale, flip_hw);
    return pg::launch_status();
}

PG_EXPORT int pg_conv2d_stem7x3_forward(const float* x, const void* packed, float* y, int N, int H, int W, int Cout, const int64_t ystride[4],
                                        const pg_conv2d_fusion* fusion, void* stream) {
    if (!x || !packed || !y || !ystride || N <= 0 || H <= 0 || W <= 0 || Cout <= 0) return PG_ERR_INVALID_ARG;
    if ((((uintptr_t)x) & 3) != 0 || (((uintptr_t)packed) & 15) != 0) return PG_ERR_INVALID_ARG;
    pgconv::Stem7Params p;
    p.x = x; p.y = y; p.wx3 = (const unsigned char*)packed; p.bias = nullptr;
    p.N = N; p.H = H; p.W = W; p.Cout = Cout;
    for (int i = 0; i < 4; i++) p.ys[i] = ystride[i];
    p.act = PG_ACT_LINEAR; p.alpha = 0.f; p.gain = 1.f; p.clamp = -1.f;
    if (fusion) {
        const pg_conv2d_fusion& f = *fusion;
        if (f.in_scale || f.in_bias || (f.in_act != 0 && f.in_act != PG_ACT_LINEAR) || f.in_clamp >= 0.f || (f.in_gain != 0.f && f.in_gain != 1.f) || f.out_scale || f.noise || f.residual ||
            f.spade_x || f.spade_mean || f.spade_rstd || f.x2 || f.stats_partial)
            return PG_ERR_UNSUPPORTED;
        const int act = f.act == 0 ? PG_ACT_LINEAR : f.act;
        if (act != PG_ACT_LINEAR && act != PG_ACT_RELU && act != PG_ACT_LRELU) return PG_ERR_UNSUPPORTED;
        if (act == PG_ACT_LRELU && !(f.alpha >= 0.f && f.alpha <= 1.f)) return PG_ERR_UNSUPPORTED;
        p.bias = f.bias; p.act = act; p.alpha = f.alpha; p.gain = f.gain == 0.f ? 1.f : f.gain; p.clamp = f.clamp;
    }
    return pgconv::launch_stem7x3(p, (hipStream_t)stream);
}
