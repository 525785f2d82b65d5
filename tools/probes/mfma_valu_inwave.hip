// Dev probe (GPU box): can ONE wave overlap its own vector-ALU work with its own MFMAs?  One wave per SIMD runs `iters` rounds of 8
// independent v_mfma_f32_32x32x2_f32 plus NV v_fma_f32 (independent of the MFMAs) placed either in one block before the MFMAs
// (mode 0) or spread between them, NV / 8 after each MFMA (mode 1).  Prints cycles per round; 8 MFMAs alone = 512.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
typedef float f32x16 __attribute__((ext_vector_type(16)));
template <int NV, int MODE>
__global__ __launch_bounds__(256) void k(float* out, long long* cyc, int iters) {
    f32x16 acc[8];
    for (int i = 0; i < 8; i++) for (int j = 0; j < 16; j++) acc[i][j] = 0.f;
    float a = threadIdx.x * 1e-3f, b = 1.0f;
    float v[8];
    for (int i = 0; i < 8; i++) v[i] = threadIdx.x * 1e-3f + i;
    const long long t0 = __builtin_readcyclecounter();
    for (int it = 0; it < iters; it++) {
        if (MODE == 0) {
#pragma unroll
            for (int r = 0; r < NV; r++) asm volatile("v_fma_f32 %0, %0, %1, %1" : "+v"(v[r & 7]) : "v"(b));
            __builtin_amdgcn_sched_barrier(0);
#pragma unroll
            for (int i = 0; i < 8; i++) acc[i] = __builtin_amdgcn_mfma_f32_32x32x2f32(a, b, acc[i], 0, 0, 0);
            __builtin_amdgcn_sched_barrier(0);
        } else {
#pragma unroll
            for (int i = 0; i < 8; i++) {
                acc[i] = __builtin_amdgcn_mfma_f32_32x32x2f32(a, b, acc[i], 0, 0, 0);
                __builtin_amdgcn_sched_barrier(0);
#pragma unroll
                for (int r = 0; r < NV / 8; r++) asm volatile("v_fma_f32 %0, %0, %1, %1" : "+v"(v[(i + r) & 7]) : "v"(b));
                __builtin_amdgcn_sched_barrier(0);
            }
        }
    }
    const long long t1 = __builtin_readcyclecounter();
    float s = 0.f;
    for (int i = 0; i < 8; i++) s += acc[i][0] + v[i];
    if (s == 123.456f) out[0] = s;
    if ((threadIdx.x & 63) == 0) cyc[blockIdx.x * 4 + (threadIdx.x >> 6)] = t1 - t0;
}
template <int NV, int MODE>
void run(float* out, long long* cyc, long long* h) {
    const int iters = 2000;
    hipLaunchKernelGGL((k<NV, MODE>), dim3(256), dim3(256), 0, 0, out, cyc, iters);
    hipDeviceSynchronize();
    hipMemcpy(h, cyc, 1024 * 8, hipMemcpyDeviceToHost);
    double s = 0; for (int i = 0; i < 1024; i++) s += (double)h[i];
    printf("NV=%3d %s: %.1f cycles per round of 8 MFMAs (+%d v_fma)\n", NV, MODE ? "interleaved" : "block      ", s / 1024 / iters, NV);
}
int main() {
    float* out; long long* cyc; static long long h[1024];
    hipMalloc(&out, 64); hipMalloc(&cyc, 1024 * 8);
    run<0, 0>(out, cyc, h);
    run<16, 0>(out, cyc, h); run<16, 1>(out, cyc, h);
    run<32, 0>(out, cyc, h); run<32, 1>(out, cyc, h);
    run<64, 0>(out, cyc, h); run<64, 1>(out, cyc, h);
    run<96, 1>(out, cyc, h); run<120, 1>(out, cyc, h);
    return 0;
}
