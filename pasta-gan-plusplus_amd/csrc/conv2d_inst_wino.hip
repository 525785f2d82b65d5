// Winograd F(2x2, 3x3) instantiations (own translation unit: see conv2d_kernel.h on build time).
// hipcc-flags: -fno-slp-vectorize
// (the K-loop transform is scalar fp32 on purpose -- packed fp32 VALU is slow beside MFMAs -- and must not be re-packed)
#include "conv2d_wino.h"

namespace pgconv {
int launch_wino(const ConvParams& p, hipStream_t s) {
    // 16-byte halo DMA needs every 4-column word of a row to be inside or outside the image as a whole
    const bool vec = p.W % 4 == 0 && (((uintptr_t)p.x) & 15) == 0;
    if (p.in_xform) return vec ? launch_wino_xf<2, true>(p, s) : launch_wino_xf<2, false>(p, s);
    if (p.f.in_scale) return vec ? launch_wino_xf<1, true>(p, s) : launch_wino_xf<1, false>(p, s);
    return vec ? launch_wino_xf<0, true>(p, s) : launch_wino_xf<0, false>(p, s);
}
}  // namespace pgconv
