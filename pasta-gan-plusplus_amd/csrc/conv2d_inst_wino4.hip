// Winograd F(4x4, 3x3) instantiations (own translation unit: see conv2d_kernel.h on build time).
// hipcc-flags: -fno-slp-vectorize
// (scalar fp32 transforms on purpose: packed fp32 VALU is slow beside MFMAs on gfx950)
#include "conv2d_wino4.h"

namespace pgconv {
int launch_wino4_spade(const ConvParams& p, hipStream_t s);      // conv2d_inst_wino4s.hip
int launch_wino4(const ConvParams& p, hipStream_t s) {
    // 16-byte halo DMA and 16-byte patch reads: every 4-column word of a row is inside or outside the image as a whole
    if (p.W % 4 != 0 || (((uintptr_t)p.x) & 15) != 0 || p.in_xform || p.f.x2 || p.pad_x < 0 || p.pad_x > 4) return PG_ERR_UNSUPPORTED;
    if (p.f.spade_x) return p.f.in_scale ? PG_ERR_UNSUPPORTED : launch_wino4_spade(p, s);
    return p.f.in_scale ? launch_wino4_mode<1>(p, s) : launch_wino4_mode<0>(p, s);
}
}  // namespace pgconv
