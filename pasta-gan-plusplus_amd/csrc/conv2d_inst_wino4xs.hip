// Winograd F(4x4, 3x3), X3 form: the SPADE-combine tail / the plain tail (own translation unit: see conv2d_kernel.h on build time).
// hipcc-flags: -fno-slp-vectorize
#include "conv2d_wino4.h"

namespace pgconv {
int launch_wino4x3_spade(const ConvParams& p, hipStream_t s) { return launch_wino4_mode<0, W4_TAIL_SPADE, true>(p, s); }
int launch_wino4x3_plain(const ConvParams& p, hipStream_t s) { return launch_wino4_mode<0, W4_TAIL_PLAIN, true>(p, s); }
}  // namespace pgconv
