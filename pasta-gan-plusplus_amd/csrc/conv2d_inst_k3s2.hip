// conv2d_mfma<KH, KW, STRIDE, BM, KC> instantiations for geometry k3s2 (see conv2d_kernel.h).
#include "conv2d_kernel.h"
namespace pgconv {
int launch_k3s2(const ConvParams& p, hipStream_t s) { return launch_bm<3, 3, 2, kc_for(3, 3, 2)>(p, s); }
}
