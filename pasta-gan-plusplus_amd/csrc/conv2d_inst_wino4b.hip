// Winograd F(4x4, 3x3), two-workgroups-per-CU form (conv2d_wino4b.h): the run-time-tail instantiations + dispatch.
// hipcc-flags: -fno-slp-vectorize
// (scalar fp32 transforms on purpose: packed fp32 VALU is slow beside MFMAs on gfx950)
#include "conv2d_wino4b.h"

namespace pgconv {
int launch_wino4b_spade(const ConvParams& p, hipStream_t s);      // conv2d_inst_wino4bs.hip
int launch_wino4b_plain(const ConvParams& p, hipStream_t s);      // conv2d_inst_wino4bs.hip
int launch_wino4b(const ConvParams& p, hipStream_t s) {
    // same acceptance as launch_wino4: 16-byte halo words and patch reads, no input pre-activation, a tail activation max() can express
    if (p.W % 4 != 0 || (((uintptr_t)p.x) & 15) != 0 || p.in_xform || p.f.x2 || p.pad_x < 0 || p.pad_x > 4) return PG_ERR_UNSUPPORTED;
    if (!(p.f.gain > 0.f) || (p.f.act == PG_ACT_LRELU && !(p.f.alpha >= 0.f && p.f.alpha <= 1.f))) return PG_ERR_UNSUPPORTED;
    if (p.f.spade_x) return p.f.in_scale ? PG_ERR_UNSUPPORTED : launch_wino4b_spade(p, s);
    if (!p.f.in_scale && !p.f.residual && !p.f.noise) return launch_wino4b_plain(p, s);
    return p.f.in_scale ? launch_wino4b_mode<1, W4_TAIL_ANY>(p, s) : launch_wino4b_mode<0, W4_TAIL_ANY>(p, s);
}
}  // namespace pgconv
