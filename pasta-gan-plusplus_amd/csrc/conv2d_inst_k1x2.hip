// conv2d_mfma<KH, KW, STRIDE, BM, KC> instantiations for geometry k1x2 (see conv2d_kernel.h).
#include "conv2d_kernel.h"
namespace pgconv {
int launch_k1x2(const ConvParams& p, hipStream_t s) { return launch_bm<1, 2, 1, kc_for(1, 2, 1)>(p, s); }
}
