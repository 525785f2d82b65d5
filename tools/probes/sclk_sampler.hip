// Dev probe (GPU box): what shader clock does the GPU actually run at while another process (bench.py) loads it?  A one-wave kernel
// spins for ~200 us and reports d(s_memtime) / d(s_memrealtime); s_memrealtime ticks at a constant 100 MHz, s_memtime with the shader
// clock (the MFMA probe next to this file measures exactly 64.0 of its ticks per v_mfma_f32_32x32x2_f32).  Prints one line per
// half second: median / min / max MHz over that window.   usage: sclk_sampler <seconds>
#include <hip/hip_runtime.h>
#include <algorithm>
#include <chrono>
#include <cstdio>
#include <cstdlib>
#include <thread>
#include <vector>
__global__ void k(unsigned long long* out, int spin_ticks) {
    unsigned long long c0, r0, c1, r1;
    asm volatile("s_memtime %0\n\ts_memrealtime %1\n\ts_waitcnt lgkmcnt(0)" : "=s"(c0), "=s"(r0));
    do {
        asm volatile("s_sleep 8\n\ts_memtime %0\n\ts_memrealtime %1\n\ts_waitcnt lgkmcnt(0)" : "=s"(c1), "=s"(r1));
    } while ((long long)(r1 - r0) < spin_ticks);
    if (threadIdx.x == 0) { out[0] = c1 - c0; out[1] = r1 - r0; }
}
int main(int argc, char** argv) {
    const double seconds = argc > 1 ? atof(argv[1]) : 10.0;
    unsigned long long* d;
    hipHostMalloc(&d, 16);
    const auto t0 = std::chrono::steady_clock::now();
    std::vector<double> win;
    double next_print = 0.5;
    while (true) {
        hipLaunchKernelGGL(k, dim3(1), dim3(64), 0, 0, d, 20000);      // 200 us at 100 MHz
        hipDeviceSynchronize();
        win.push_back(100.0 * (double)d[0] / (double)d[1]);
        const double t = std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count();
        if (t >= next_print) {
            std::sort(win.begin(), win.end());
            printf("t=%5.1fs  sclk MHz median %7.1f  min %7.1f  max %7.1f  (%zu samples)\n", t, win[win.size() / 2], win.front(), win.back(), win.size());
            fflush(stdout);
            win.clear();
            next_print += 0.5;
        }
        if (t > seconds) break;
        std::this_thread::sleep_for(std::chrono::milliseconds(5));
    }
    return 0;
}
