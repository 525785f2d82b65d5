// Shared device/host helpers for the gfx950 kernels of this package.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <atomic>
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

// Compute units of the CURRENT device (256 on MI355X), queried once per device; memory-bound grids are capped at
// CUs x 8 blocks and grid-stride the rest.
constexpr int kMaxDevices = 64;
inline int current_device() {
    int d = 0;
    return hipGetDevice(&d) == hipSuccess && d >= 0 && d < kMaxDevices ? d : 0;
}
// CUs a data-parallel training step keeps free of this library's persistent grids (round 5): the convolution / weight-gradient kernels are one workgroup
// per CU with nearly all of its LDS, so RCCL's channel workgroups -- launched on a side stream while the backward pass runs (training/ddp.py) -- could only
// start when a CU drained.  With a reservation every grid and planner of this plugin sizes itself for CUs - reserved (pg_conv2d_reserve_cus).
inline std::atomic<int>& reserved_cus() {
    static std::atomic<int> r{0};
    return r;
}
inline int num_cu() {
    static std::atomic<int> cached[kMaxDevices];
    const int d = current_device();
    int v = cached[d].load(std::memory_order_relaxed);
    if (v <= 0) {
        if (hipDeviceGetAttribute(&v, hipDeviceAttributeMultiprocessorCount, d) != hipSuccess || v <= 0) v = 256;
        cached[d].store(v, std::memory_order_relaxed);
    }
    const int r = reserved_cus().load(std::memory_order_relaxed);
    return v - r >= 8 ? v - r : (v >= 8 ? 8 : v);
}
inline int max_stream_blocks() { return num_cu() * 8; }

// hipFuncAttributeMaxDynamicSharedMemorySize is a per-DEVICE attribute of a kernel: one flag per (kernel, device).
struct PerDeviceOnce {
    std::atomic<bool> done[kMaxDevices];
    template <typename F> hipError_t run(F&& f) {
        const int d = current_device();
        if (done[d].load(std::memory_order_acquire)) return hipSuccess;
        const hipError_t e = f();
        if (e == hipSuccess) done[d].store(true, std::memory_order_release);
        return e;
    }
};

}  // namespace pg
