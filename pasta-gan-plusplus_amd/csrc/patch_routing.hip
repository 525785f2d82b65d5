// Patch-routing warps of the data loader on gfx950: OpenCV's warpPerspective (8-bit, INTER_LINEAR,
// BORDER_CONSTANT 0) restated bit for bit, batched over jobs, plus the erode-and-paste step of the de-normalisation.
// What it replaces: cv2.warpPerspective / cv2.erode calls of training/dataset.py:2555-2700 (44 warps per sample, single-threaded on
// the DataLoader's main thread in the reference).  Integer/byte work: one thread per destination pixel, coordinates in fp64
// with the evaluation order of OpenCV's WarpPerspectiveInvoker (imgwarp.cpp) -- no fused multiply-adds, round half to even --
// weights 15-bit integers; HBM/L2-bound gather (each destination pixel reads a 2x2 source patch).
#include "pg_common.h"
#include <cstdint>

namespace {

using namespace pg;

constexpr int kInterBits = 5, kTab = 1 << kInterBits;

__global__ __launch_bounds__(256) void warp_perspective_u8_kernel(const pg_warp_job* __restrict__ jobs) {
    const pg_warp_job jb = jobs[blockIdx.y];
    const int npix = jb.dst_h * jb.dst_w;
    for (int i = blockIdx.x * 256 + threadIdx.x; i < npix; i += gridDim.x * 256) {
        const int y = i / jb.dst_w, x = i - y * jb.dst_w;
        const int xb = (x / jb.block_w) * jb.block_w, x1 = x - xb;       // OpenCV walks blocks: origin term + offset term
        const double* m = jb.minv;
        const double X0 = __dadd_rn(__dadd_rn(__dmul_rn(m[0], (double)xb), __dmul_rn(m[1], (double)y)), m[2]);
        const double Y0 = __dadd_rn(__dadd_rn(__dmul_rn(m[3], (double)xb), __dmul_rn(m[4], (double)y)), m[5]);
        const double W0 = __dadd_rn(__dadd_rn(__dmul_rn(m[6], (double)xb), __dmul_rn(m[7], (double)y)), m[8]);
        double W = __dadd_rn(W0, __dmul_rn(m[6], (double)x1));
        W = W != 0.0 ? __ddiv_rn((double)kTab, W) : 0.0;
        double fX = __dmul_rn(__dadd_rn(X0, __dmul_rn(m[0], (double)x1)), W);
        double fY = __dmul_rn(__dadd_rn(Y0, __dmul_rn(m[3], (double)x1)), W);
        fX = fmax(-2147483648.0, fmin(2147483647.0, fX));
        fY = fmax(-2147483648.0, fmin(2147483647.0, fY));
        const int X = (int)rint(fX), Y = (int)rint(fY);                  // cvRound: round half to even
        int sx = X >> kInterBits, sy = Y >> kInterBits;
        sx = sx < -32768 ? -32768 : (sx > 32767 ? 32767 : sx);           // saturate_cast<short>
        sy = sy < -32768 ? -32768 : (sy > 32767 ? 32767 : sy);
        const int fx = X & (kTab - 1), fy = Y & (kTab - 1);
        int w00 = (32 - fx) * (32 - fy) * 32, w01 = fx * (32 - fy) * 32, w10 = (32 - fx) * fy * 32, w11 = fx * fy * 32;
        if ((fx | fy) == 0) { w00 = 32767; w11 = 1; }                    // initInterTab2D: 1.0 saturates, the correction lands on the last tap
        const bool y0 = sy >= 0 && sy < jb.src_h, y1 = sy + 1 >= 0 && sy + 1 < jb.src_h;
        const bool x0 = sx >= 0 && sx < jb.src_w, x1ok = sx + 1 >= 0 && sx + 1 < jb.src_w;
        const int C = jb.channels;
        const uint8_t* s = jb.src;
        uint8_t* d = jb.dst + (int64_t)i * C;
        for (int c = 0; c < C; c++) {
            const int p00 = (y0 && x0) ? s[((int64_t)sy * jb.src_w + sx) * C + c] : 0;
            const int p01 = (y0 && x1ok) ? s[((int64_t)sy * jb.src_w + sx + 1) * C + c] : 0;
            const int p10 = (y1 && x0) ? s[((int64_t)(sy + 1) * jb.src_w + sx) * C + c] : 0;
            const int p11 = (y1 && x1ok) ? s[((int64_t)(sy + 1) * jb.src_w + sx + 1) * C + c] : 0;
            int v = (p00 * w00 + p01 * w01 + p10 * w10 + p11 * w11 + (1 << 14)) >> 15;
            d[c] = (uint8_t)(v < 0 ? 0 : (v > 255 ? 255 : v));
        }
    }
}

// canvas[p] = (erode8x8(mask channel 0)[p] == 255) ? patch[p] : canvas[p]      (dataset.py:2624-2630)
// erode: minimum over the 8x8 window anchored at (4, 4), pixels outside the image ignored -> "== 255" iff every in-range tap is 255
__global__ __launch_bounds__(256) void patch_compose_u8_kernel(const uint8_t* __restrict__ patch, const uint8_t* __restrict__ mask, uint8_t* __restrict__ canvas,
                                                               uint8_t* __restrict__ canvas2, int h, int w, int mc) {
    const int npix = h * w;
    for (int i = blockIdx.x * 256 + threadIdx.x; i < npix; i += gridDim.x * 256) {
        const int y = i / w, x = i - y * w;
        bool all = true;
        for (int ky = 0; ky < 8 && all; ky++) {
            const int yy = y + ky - 4;
            if (yy < 0 || yy >= h) continue;
            for (int kx = 0; kx < 8; kx++) {
                const int xx = x + kx - 4;
                if (xx < 0 || xx >= w) continue;
                if (mask[((int64_t)yy * w + xx) * mc] != 255) { all = false; break; }
            }
        }
        if (all) {
#pragma unroll
            for (int c = 0; c < 3; c++) {
                const uint8_t v = patch[(int64_t)i * 3 + c];
                canvas[(int64_t)i * 3 + c] = v;
                if (canvas2) canvas2[(int64_t)i * 3 + c] = v;
            }
        }
    }
}

// The whole paste sequence of one canvas in one pass (round 6): the reference pastes its parts one after the other, later parts overwriting earlier ones
// (dataset.py:2620-2633), i.e. a pixel ends up with the patch of the LAST part whose eroded mask is set there, or 0.  One job = one canvas (+ the copy that
// skips the sleeve parts); one thread = one pixel, walking the job's parts in order.  Every canvas pixel is written: the canvases need no zero fill.
__global__ __launch_bounds__(256) void patch_compose_ordered_u8_kernel(const pg_compose_job* __restrict__ jobs, int h, int w, int mc) {
    const pg_compose_job* jb = jobs + blockIdx.y;
    const int npix = h * w, nparts = jb->nparts;
    for (int i = blockIdx.x * 256 + threadIdx.x; i < npix; i += gridDim.x * 256) {
        const int y = i / w, x = i - y * w;
        int last = -1, last2 = -1;
        for (int p = 0; p < nparts; p++) {
            const uint8_t* mask = jb->mask[p];
            bool all = mask[(int64_t)i * mc] == 255;          // (the window's own pixel first: most pixels lie outside a part)
            for (int ky = 0; ky < 8 && all; ky++) {
                const int yy = y + ky - 4;
                if (yy < 0 || yy >= h) continue;
                for (int kx = 0; kx < 8; kx++) {
                    const int xx = x + kx - 4;
                    if (xx < 0 || xx >= w) continue;
                    if (mask[((int64_t)yy * w + xx) * mc] != 255) { all = false; break; }
                }
            }
            if (all) {
                last = p;
                if (jb->to_canvas2[p]) last2 = p;
            }
        }
#pragma unroll
        for (int c = 0; c < 3; c++) {
            jb->canvas[(int64_t)i * 3 + c] = last >= 0 ? jb->patch[last][(int64_t)i * 3 + c] : (uint8_t)0;
            if (jb->canvas2) jb->canvas2[(int64_t)i * 3 + c] = last2 >= 0 ? jb->patch[last2][(int64_t)i * 3 + c] : (uint8_t)0;
        }
    }
}

}  // namespace

PG_EXPORT int pg_patch_routing_abi_version(void) { return PG_ABI_VERSION; }

PG_EXPORT int pg_patch_compose_ordered_u8(const pg_compose_job* jobs_device, int njobs, int h, int w, int mask_channels, void* stream) {
    if (!jobs_device || njobs <= 0 || h <= 0 || w <= 0 || mask_channels <= 0) return PG_ERR_INVALID_ARG;
    if ((int64_t)h * w > 0x3fffffffLL) return PG_ERR_TOO_LARGE;
    if (njobs > 65535) return PG_ERR_TOO_LARGE;
    int bx = (h * w + 255) / 256;
    if (bx > 1024) bx = 1024;
    hipLaunchKernelGGL(patch_compose_ordered_u8_kernel, dim3((unsigned)bx, (unsigned)njobs), dim3(256), 0, (hipStream_t)stream, jobs_device, h, w, mask_channels);
    return pg::launch_status();
}

PG_EXPORT int pg_warp_perspective_u8(const pg_warp_job* jobs_device, int njobs, int max_dst_pixels, void* stream) {
    if (!jobs_device || njobs <= 0 || max_dst_pixels <= 0) return PG_ERR_INVALID_ARG;
    if (njobs > 65535) return PG_ERR_TOO_LARGE;
    int bx = (max_dst_pixels + 255) / 256;
    if (bx > 1024) bx = 1024;
    hipLaunchKernelGGL(warp_perspective_u8_kernel, dim3((unsigned)bx, (unsigned)njobs), dim3(256), 0, (hipStream_t)stream, jobs_device);
    return pg::launch_status();
}

PG_EXPORT int pg_patch_compose_u8(const uint8_t* patch, const uint8_t* mask, uint8_t* canvas, uint8_t* canvas2, int h, int w, int mask_channels, void* stream) {
    if (!patch || !mask || !canvas || h <= 0 || w <= 0 || mask_channels <= 0) return PG_ERR_INVALID_ARG;
    if ((int64_t)h * w > 0x3fffffffLL) return PG_ERR_TOO_LARGE;
    int bx = (h * w + 255) / 256;
    if (bx > pg::max_stream_blocks()) bx = pg::max_stream_blocks();
    hipLaunchKernelGGL(patch_compose_u8_kernel, dim3((unsigned)bx), dim3(256), 0, (hipStream_t)stream, patch, mask, canvas, canvas2, h, w, mask_channels);
    return pg::launch_status();
}
