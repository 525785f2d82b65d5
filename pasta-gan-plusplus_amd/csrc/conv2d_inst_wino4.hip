// Winograd F(4x4, 3x3) instantiations (own translation units: see conv2d_kernel.h on build time): the run-time-tail fallbacks + dispatch.
// hipcc-flags: -fno-slp-vectorize
// (scalar fp32 transforms on purpose: packed fp32 VALU is slow beside MFMAs on gfx950)
#include "conv2d_wino4.h"

namespace pgconv {
int launch_wino4_spade(const ConvParams& p, hipStream_t s);      // conv2d_inst_wino4s.hip
int launch_wino4_plain(const ConvParams& p, hipStream_t s);      // conv2d_inst_wino4s.hip
int launch_wino4_stats(const ConvParams& p, hipStream_t s);      // conv2d_inst_wino4t.hip
int launch_wino4(const ConvParams& p, hipStream_t s) {
    // 16-byte halo DMA and 16-byte patch reads: every 4-column word of a row is inside or outside the image as a whole
    if (p.W % 4 != 0 || (((uintptr_t)p.x) & 15) != 0 || p.in_xform || p.f.x2 || p.pad_x < 0 || p.pad_x > 4) return PG_ERR_UNSUPPORTED;
    // the tail's activation is max(v * gain, v * gain * slope): exact only for gain > 0 and a slope in [0, 1] -- anything else
    // is declined here and runs on F(2x2) / the direct kernel, whose tails use the select form
    if (!(p.f.gain > 0.f) || (p.f.act == PG_ACT_LRELU && !(p.f.alpha >= 0.f && p.f.alpha <= 1.f))) return PG_ERR_UNSUPPORTED;
    static const bool split = [] { const char* e = getenv("PG_WINO4_TAILS"); return e ? atoi(e) != 0 : true; }();      // A/B switch: 0 = the run-time tail for everything but SPADE
    if (p.f.stats_partial && (p.f.spade_x || p.f.in_scale || p.f.residual || p.f.noise)) return PG_ERR_UNSUPPORTED;      // output statistics: plain tail only
    if (p.f.spade_x) return p.f.in_scale ? PG_ERR_UNSUPPORTED : launch_wino4_spade(p, s);
    // (residual-only and modulated + noise instantiations were built and measured too: +1 % and -1 ... +5 % against the run-time tail -- not kept)
    if (p.f.stats_partial) return launch_wino4_stats(p, s);
    if (split && !p.f.in_scale && !p.f.residual && !p.f.noise) return launch_wino4_plain(p, s);
    return p.f.in_scale ? launch_wino4_mode<1, W4_TAIL_ANY>(p, s) : launch_wino4_mode<0, W4_TAIL_ANY>(p, s);
}
}  // namespace pgconv
