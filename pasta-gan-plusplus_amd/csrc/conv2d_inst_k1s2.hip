// conv2d_mfma<KH, KW, STRIDE, BM, KC> instantiations for geometry k1s2 (see conv2d_kernel.h).
#include "conv2d_kernel.h"
namespace pgconv {
int launch_k1s2(const ConvParams& p, hipStream_t s) { return launch_bm<1, 1, 2, kc_for(1, 1, 2)>(p, s); }
}
