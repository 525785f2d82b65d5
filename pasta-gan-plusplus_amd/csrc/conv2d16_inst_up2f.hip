// conv2d_up2f16<T> instantiations (see conv2d_up2f16.h): the 16-bit up = 2 layer with the y half of the FIR in the weights and the x half in the epilogue.
#include "conv2d_up2f16.h"
namespace pgconv16 {
int launch16_up2f(const Up2fParams& p, int dtype, hipStream_t s) {
    // PG_UP2F_WG=1 (default): one 12-wave workgroup per CU (16-row tiles, three staging buffers, the epilogue after the K loop); 2: two 6-wave workgroups per CU (8-row tiles, two buffers); 0: conv2d_up2f16p
    static const int per_cu = [] { const char* e = getenv("PG_UP2F_WG"); return e ? atoi(e) : 1; }();
    if (per_cu == 0) {                                                     // experimental: the eight-wave ping-pong form (conv2d_up2f16p; parity-green, slower for now)
        if (dtype == PG_BF16) return launch_up2f16p<bf16_t>(p, s);
        if (dtype == PG_F16) return launch_up2f16p<f16_t>(p, s);
    } else if (per_cu == 1) {
        if (dtype == PG_BF16) return launch_up2f16<bf16_t, 8, 3>(p, s);
        if (dtype == PG_F16) return launch_up2f16<f16_t, 8, 3>(p, s);
    } else {
        if (dtype == PG_BF16) return launch_up2f16<bf16_t, 4, 2>(p, s);
        if (dtype == PG_F16) return launch_up2f16<f16_t, 4, 2>(p, s);
    }
    return PG_ERR_INVALID_ARG;
}
}
