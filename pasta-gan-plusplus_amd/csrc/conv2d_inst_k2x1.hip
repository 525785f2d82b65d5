// conv2d_mfma<KH, KW, STRIDE, BM, KC> instantiations for geometry k2x1 (see conv2d_kernel.h).
#include "conv2d_kernel.h"
namespace pgconv {
int launch_k2x1(const ConvParams& p, hipStream_t s) { return launch_bm<2, 1, 1, kc_for(2, 1, 1)>(p, s); }
}
