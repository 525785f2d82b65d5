// Winograd F(2x2, 3x3) instantiations (own translation unit: see conv2d_kernel.h on build time).
#include "conv2d_wino.h"

namespace pgconv {
int launch_wino(const ConvParams& p, hipStream_t s) {
    return p.in_xform ? launch_wino_xf<true>(p, s) : launch_wino_xf<false>(p, s);
}
}  // namespace pgconv
