// Winograd F(4x4, 3x3), X3 form (the transform-domain GEMM as six bf16 products of exact three-term splits, conv2d_wino4.h): run-time-tail instantiations + dispatch.
// hipcc-flags: -fno-slp-vectorize
#include "conv2d_wino4.h"

namespace pgconv {
int launch_wino4x3_spade(const ConvParams& p, hipStream_t s);      // conv2d_inst_wino4xs.hip
int launch_wino4x3_plain(const ConvParams& p, hipStream_t s);      // conv2d_inst_wino4xs.hip
int launch_wino4x3_stats(const ConvParams& p, hipStream_t s);      // conv2d_inst_wino4xt.hip
int launch_wino4x3(const ConvParams& p, hipStream_t s) {
    // acceptance rules: launch_wino4's (conv2d_inst_wino4.hip) -- the two forms differ in the K loop's arithmetic only
    if (p.W % 4 != 0 || (((uintptr_t)p.x) & 15) != 0 || p.in_xform || p.f.x2 || p.pad_x < 0 || p.pad_x > 4) return PG_ERR_UNSUPPORTED;
    if (!(p.f.gain > 0.f) || (p.f.act == PG_ACT_LRELU && !(p.f.alpha >= 0.f && p.f.alpha <= 1.f))) return PG_ERR_UNSUPPORTED;
    if (p.f.stats_partial && (p.f.spade_x || p.f.in_scale || p.f.residual || p.f.noise)) return PG_ERR_UNSUPPORTED;
    if (p.f.spade_x) return p.f.in_scale ? PG_ERR_UNSUPPORTED : launch_wino4x3_spade(p, s);
    if (p.f.stats_partial) return launch_wino4x3_stats(p, s);
    if (!p.f.in_scale && !p.f.residual && !p.f.noise) return launch_wino4x3_plain(p, s);
    return p.f.in_scale ? launch_wino4_mode<1, W4_TAIL_ANY, true>(p, s) : launch_wino4_mode<0, W4_TAIL_ANY, true>(p, s);
}
}  // namespace pgconv
