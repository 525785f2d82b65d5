// Dev probe (GPU box): semantics of `buffer_load_dwordx4 ... offen lds` on gfx950 -- LDS placement (M0 + lane*16),
// out-of-range lanes (sentinel offset / straddling num_records) and 4-byte-aligned (not 16-byte-aligned) source addresses.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>
#include <vector>
typedef int i32x4 __attribute__((ext_vector_type(4)));
__global__ void k(const float* x, float* y, int nbytes, const unsigned* voffs) {
    extern __shared__ float sm[];
    for (int i = threadIdx.x; i < 1024; i += 64) sm[i] = -7.f;
    __syncthreads();
    i32x4 rsrc;
    const uint64_t base = (uint64_t)(uintptr_t)x;
    rsrc[0] = __builtin_amdgcn_readfirstlane((int)(unsigned)base);
    rsrc[1] = __builtin_amdgcn_readfirstlane((int)(unsigned)(base >> 32) & 0xffff);
    rsrc[2] = nbytes; rsrc[3] = 0x00020000;
    unsigned keep; unsigned lds = 68; unsigned voff = voffs[threadIdx.x]; int soff = 0;
    asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %1\n\ts_nop 0\n\tbuffer_load_dwordx4 %2, %3, %4 offen lds\n\ts_mov_b32 m0, %0"
                 : "=&s"(keep) : "s"(lds), "v"(voff), "s"(rsrc), "s"(soff) : "memory");
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
    for (int i = threadIdx.x; i < 1024; i += 64) y[i] = sm[i];
}
int main() {
    const int n = 4096;
    std::vector<float> hx(n); for (int i = 0; i < n; i++) hx[i] = (float)i;
    std::vector<unsigned> hv(64);
    for (int l = 0; l < 64; l++) hv[l] = (unsigned)(l * 16 * 4 + 4);      // 4-byte aligned, not 16: floats 16l+1 .. 16l+4
    hv[3] = 0x80000000u;                                                  // sentinel
    hv[5] = (unsigned)(1000 * 4 - 8);                                     // straddles num_records = 1000 floats
    hv[7] = (unsigned)(1000 * 4);                                         // starts at the end
    float *dx, *dy; unsigned* dv;
    hipMalloc(&dx, n * 4); hipMalloc(&dy, 1024 * 4); hipMalloc(&dv, 64 * 4);
    hipMemcpy(dx, hx.data(), n * 4, hipMemcpyHostToDevice); hipMemcpy(dv, hv.data(), 64 * 4, hipMemcpyHostToDevice);
    hipLaunchKernelGGL(k, dim3(1), dim3(64), 4096, 0, dx, dy, 1000 * 4, dv);
    std::vector<float> hy(1024); hipMemcpy(hy.data(), dy, 1024 * 4, hipMemcpyDeviceToHost);
    printf("before M0 region (floats 12..15): %g %g %g %g\n", hy[12], hy[13], hy[14], hy[15]);
    for (int l : {0, 1, 2, 3, 4, 5, 6, 7, 8, 63}) printf("(M0 = 68: +1 float) lane %2d -> LDS floats %d+1..: %g %g %g %g\n", l, 16 + 4 * l, hy[17 + 4 * l], hy[18 + 4 * l], hy[19 + 4 * l], hy[20 + 4 * l]);
    printf("after (float %d): %g\n", 16 + 256, hy[16 + 256]);
    return 0;
}
