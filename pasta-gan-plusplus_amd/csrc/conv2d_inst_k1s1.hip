// conv2d_mfma<KH, KW, STRIDE, BM, KC> instantiations for geometry k1s1 (see conv2d_kernel.h).
#include "conv2d_kernel.h"
namespace pgconv {
int launch_k1s1(const ConvParams& p, hipStream_t s) { return launch_bm<1, 1, 1, kc_for(1, 1, 1), true>(p, s); }
}
