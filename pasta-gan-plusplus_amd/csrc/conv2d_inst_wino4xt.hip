// Winograd F(4x4, 3x3), X3 form: the plain tail that also gathers the output's instance-norm statistics (own translation unit: see conv2d_kernel.h on build time).
// hipcc-flags: -fno-slp-vectorize
#include "conv2d_wino4.h"

namespace pgconv {
int launch_wino4x3_stats(const ConvParams& p, hipStream_t s) { return launch_wino4_mode<0, W4_TAIL_STATS, true>(p, s); }
}  // namespace pgconv
