// Winograd F(4x4, 3x3), two-workgroups-per-CU form: plain input with the SPADE-combine tail / the plain tail (own translation unit: build time).
// hipcc-flags: -fno-slp-vectorize
#include "conv2d_wino4b.h"

namespace pgconv {
int launch_wino4b_spade(const ConvParams& p, hipStream_t s) { return launch_wino4b_mode<0, W4_TAIL_SPADE>(p, s); }
int launch_wino4b_plain(const ConvParams& p, hipStream_t s) { return launch_wino4b_mode<0, W4_TAIL_PLAIN>(p, s); }
}  // namespace pgconv
