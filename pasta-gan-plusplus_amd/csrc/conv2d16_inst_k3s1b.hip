// conv2d_mfma16 instantiations for geometry k3s1 with 32-channel K chunks: two-role form, two staging buffers (see conv2d_kernel16.h).
#include "conv2d_kernel16.h"
namespace pgconv16 {
int launch16_k3s1_kc32(const Conv16Params& p, int dtype, hipStream_t s) { return launch16_dt<3, 3, 1, 2, 32, 2>(p, dtype, s); }
}
