// conv2d_mfma<KH, KW, STRIDE, BM, KC> instantiations for geometry k7s1 (see conv2d_kernel.h).
#include "conv2d_kernel.h"
namespace pgconv {
int launch_k7s1(const ConvParams& p, hipStream_t s) { return launch_bm<7, 7, 1, kc_for(7, 7, 1)>(p, s); }
}
