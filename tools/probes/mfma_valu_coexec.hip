// Dev probe (GPU box): does vector-ALU work issued by ANOTHER wave on the same SIMD take matrix-pipe time from a wave that issues
// fp32 MFMAs back to back?  One workgroup per CU of 4 or 8 waves: waves 0-3 (one per SIMD) run `iters` rounds of 8 independent
// v_mfma_f32_32x32x2_f32; waves 4-7, when present, run `iters` rounds of `nvalu` dependent-free v_fma_f32 (mode 1) or nothing
// but s_sleep (mode 2).  Prints the MFMA waves' cycles per MFMA.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
typedef float f32x16 __attribute__((ext_vector_type(16)));
template <int NVALU>
__global__ __launch_bounds__(768) void k(float* out, long long* cyc, int iters, int mode, int nmf) {
    const int wave = threadIdx.x >> 6;
    if (wave < nmf) {
        f32x16 acc[8];
        for (int i = 0; i < 8; i++) for (int j = 0; j < 16; j++) acc[i][j] = 0.f;
        float a = threadIdx.x * 1e-3f, b = 1.0f;
        const long long t0 = __builtin_readcyclecounter();
        for (int it = 0; it < iters; it++) {
#pragma unroll
            for (int i = 0; i < 8; i++) acc[i] = __builtin_amdgcn_mfma_f32_32x32x2f32(a, b, acc[i], 0, 0, 0);
        }
        const long long t1 = __builtin_readcyclecounter();
        float s = 0.f;
        for (int i = 0; i < 8; i++) s += acc[i][0];
        if (s == 123.456f) out[0] = s;
        if ((threadIdx.x & 63) == 0) cyc[blockIdx.x * 4 + (wave & 3)] = t1 - t0;
    } else if (mode == 1) {
        float v[8];
        for (int i = 0; i < 8; i++) v[i] = threadIdx.x * 1e-3f + i;
        const long long t0 = __builtin_readcyclecounter();
        for (int it = 0; it < iters; it++) {
#pragma unroll
            for (int r = 0; r < NVALU / 8; r++)
#pragma unroll
                for (int i = 0; i < 8; i++) v[i] = __builtin_fmaf(v[i], 1.0001f, 0.5f);
        }
        const long long t1 = __builtin_readcyclecounter();
        float s = 0.f;
        for (int i = 0; i < 8; i++) s += v[i];
        if (s == 123.456f) out[1] = s;
        if ((threadIdx.x & 63) == 0) cyc[1024 + blockIdx.x * 4 + (wave & 3)] = t1 - t0;
    }
}
template <int NVALU>
void run(int waves, int mode, const char* what, int nmf = 4) {
    float* out; long long* cyc;
    hipMalloc(&out, 64); hipMalloc(&cyc, 2 * 256 * 4 * 8); hipMemset(cyc, 0, 2 * 256 * 4 * 8);
    const int iters = 4000;
    for (int rep = 0; rep < 2; rep++) hipLaunchKernelGGL((k<NVALU>), dim3(256), dim3(waves * 64), 0, 0, out, cyc, iters, mode, nmf);
    hipDeviceSynchronize();
    long long h[2048];
    hipMemcpy(h, cyc, 2048 * 8, hipMemcpyDeviceToHost);
    double m = 0, v = 0; for (int i = 0; i < 1024; i++) { m += (double)h[i]; v += (double)h[1024 + i]; }
    m /= 1024.0; v /= 1024.0;
    printf("%-58s %7.1f clock ticks per MFMA", what, m / (iters * 8.0));
    if (mode == 1) printf("   partner: %6.2f ticks per v_fma (alone: 4 on a 16-lane SIMD pass)", v / ((double)iters * NVALU));
    printf("\n");
    hipFree(out); hipFree(cyc);
}
int main() {
    run<8>(4, 0, "MFMA waves alone (1 per SIMD)");
    run<8>(8, 2, "+ an idle partner wave per SIMD");
    run<8>(8, 1, "+ partner: 8 v_fma per 8 MFMAs");
    run<32>(8, 1, "+ partner: 32 v_fma per 8 MFMAs");
    run<64>(8, 1, "+ partner: 64 v_fma per 8 MFMAs");
    run<128>(8, 1, "+ partner: 128 v_fma per 8 MFMAs (VALU-saturating)");
    run<8>(8, 0, "TWO MFMA waves per SIMD, no partner", 8);
    run<32>(12, 1, "TWO MFMA waves per SIMD + partner: 32 v_fma per 8 MFMAs", 8);
    run<128>(12, 1, "TWO MFMA waves per SIMD + partner: 128 v_fma per 8 MFMAs", 8);
    return 0;
}
