// conv2d_mfma16<T, KH, KW, STRIDE, ...> instantiations for geometry k2x1 (see conv2d_kernel16.h): <KH, KW, stride, N tiles per wave,
// channels per chunk[, staging buffers]> x {bf16, fp16} x {32, 64}-cout blocks.
#include "conv2d_kernel16.h"
namespace pgconv16 {
int launch16_k2x1(const Conv16Params& p, int dtype, hipStream_t s) { return launch16_dt<2, 1, 1, 2, 32>(p, dtype, s); }
}
