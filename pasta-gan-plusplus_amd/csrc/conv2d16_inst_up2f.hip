// conv2d_up2f16<T> instantiations (see conv2d_up2f16.h): the 16-bit up = 2 layer with the y half of the FIR in the weights and the x half in the epilogue.
#include "conv2d_up2f16.h"
namespace pgconv16 {
int launch16_up2f(const Up2fParams& p, int dtype, hipStream_t s) {
    if (dtype == PG_BF16) return launch_up2f16<bf16_t>(p, s);
    if (dtype == PG_F16) return launch_up2f16<f16_t>(p, s);
    return PG_ERR_INVALID_ARG;
}
}
