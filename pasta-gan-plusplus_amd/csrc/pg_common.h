// Shared device/host helpers for the gfx950 kernels of this package.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>
#include "pasta_gan_ops.h"

#define PG_EXPORT extern "C" __attribute__((visibility("default")))

namespace pg {

// Storage type -> arithmetic type (fp32 for 16-bit storage; the reference accumulates
// half in float too, bias_act.cu:15-18 / upfirdn2d.cu:15-18).
template <typename T> struct acc_of { typedef float type; };
template <> struct acc_of<double> { typedef double type; };

typedef _Float16 f16_t;
typedef __bf16 bf16_t;

template <typename T, int N> struct alignas(sizeof(T) * N) vec_t { T v[N]; };

static inline int launch_status() {
    hipError_t e = hipGetLastError();
    return e == hipSuccess ? PG_OK : (int)e;
}

__host__ __device__ static inline bool aligned16(const void* p) { return (reinterpret_cast<uintptr_t>(p) & 15u) == 0; }

// MI355X: 256 CUs; memory-bound grids are capped at 256 CUs x 8 blocks and grid-stride the rest.
constexpr int kNumCU = 256;
constexpr int kMaxStreamBlocks = kNumCU * 8;

}  // namespace pg
