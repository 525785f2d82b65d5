// Fused four-parity stride-2 transposed 3x3 convolution: instantiation + C ABI (own translation unit: see conv2d_kernel.h on build time).
#include "conv2d_up2x3.h"

/* y[n, co, 2 iy + ky, 2 ix + kx] += x[n, ci, iy, ix] * in_scale[n, ci] * w[co, ci, ky, kx], then * out_scale[n, co]:
 * conv_transpose2d(stride 2, padding 0) of a 3x3 kernel (conv2d_gradfix.py:46-53 behind conv2d_resample.py:125-142) with the
 * modulation / demodulation of networks.py:73-94 around it.  `packed` = pg_conv2d_pack_weight of the OIHW kernel w (3x3).
 * y is [N, Cout, 2H+1, 2W+1] with strides ystride (elements); an even row pitch gives 8-byte stores.  Every element of y is written
 * (the last column ox = 2W by the edge tiles of the same launch, conv2d_up2.h). */
static int up2_forward(const float* x, const float* packed, float* y, int N, int Cin, int H, int W, int Cout,
                       const int64_t ystride[4], const float* in_scale, const float* out_scale, float* workspace, int ksplit, void* stream) {
    if (!x || !packed || !y || !ystride || N <= 0 || Cin <= 0 || H <= 0 || W <= 0 || Cout <= 0) return PG_ERR_INVALID_ARG;
    if ((((uintptr_t)x) & 3) != 0 || (((uintptr_t)packed) & 15) != 0) return PG_ERR_INVALID_ARG;
    if ((int64_t)Cin * H * W * 4 > 0x7fffffffLL) return PG_ERR_TOO_LARGE;             // one image through a 32-bit buffer descriptor
    pgconv::Up2Params p;
    p.x = x; p.wp = packed; p.y = y; p.in_scale = in_scale; p.out_scale = out_scale;
    p.N = N; p.Cin = Cin; p.H = H; p.W = W; p.Cout = Cout; p.CoutP = (Cout + 31) / 32 * 32;
    for (int i = 0; i < 4; i++) p.ys[i] = ystride[i];
    p.ksplit = 1; p.cpk = 0; p.ws_slice = 0;
    if (ksplit <= 1) return pgconv::launch_up2(p, (hipStream_t)stream);
    // split-K: the shares write slices laid out like y (one slice = N * ystride[0] floats: y dense over n), then one pass adds them into y
    const int64_t slice = (int64_t)N * ystride[0];
    if (!workspace || slice % 4 != 0 || (((uintptr_t)workspace) & 15) != 0 || (((uintptr_t)y) & 15) != 0) return PG_ERR_INVALID_ARG;
    p.y = workspace; p.ksplit = ksplit; p.ws_slice = slice;
    const int st = pgconv::launch_up2(p, (hipStream_t)stream);
    if (st != PG_OK) return st;
    int64_t blocks = (slice / 4 + 255) / 256;
    if (blocks > (int64_t)pg::max_stream_blocks()) blocks = pg::max_stream_blocks();
    hipLaunchKernelGGL(pgconv::up2_sum_slices, dim3((unsigned)blocks), dim3(256), 0, (hipStream_t)stream, workspace, y, ksplit, slice / 4);
    return pg::launch_status();
}

PG_EXPORT int pg_conv2d_up2_forward(const float* x, const float* packed, float* y, int N, int Cin, int H, int W, int Cout,
                                    const int64_t ystride[4], const float* in_scale, const float* out_scale, void* stream) {
    return up2_forward(x, packed, y, N, Cin, H, W, Cout, ystride, in_scale, out_scale, nullptr, 1, stream);
}

/* Split-K form for the low-resolution layers: pg_conv2d_up2_splitk_plan = the share count the launch wants (1 = use pg_conv2d_up2_forward);
 * `workspace` = ksplit * N * ystride[0] floats (16-byte aligned; y must be dense over n with N * ystride[0] % 4 == 0, as conv_up2_forward allocates it). */
PG_EXPORT int pg_conv2d_up2_splitk_plan(int N, int Cin, int H, int W, int Cout) {
    if (N <= 0 || Cin <= 0 || H <= 0 || W <= 0 || Cout <= 0) return 1;
    return pgconv::up2_splitk_plan(N, Cin, H, W, Cout);
}

PG_EXPORT int pg_conv2d_up2_forward_splitk(const float* x, const float* packed, float* y, int N, int Cin, int H, int W, int Cout,
                                           const int64_t ystride[4], const float* in_scale, const float* out_scale, float* workspace, int ksplit, void* stream) {
    if (ksplit < 1) return PG_ERR_INVALID_ARG;
    return up2_forward(x, packed, y, N, Cin, H, W, Cout, ystride, in_scale, out_scale, workspace, ksplit, stream);
}

/* Round 6 -- the same layer with its multiplies on the bf16 matrix pipe (csrc/conv2d_up2x3.h): float32 operands as exact sums of three bf16 values, six plane
 * products per float32 product on v_mfma_f32_32x32x16_bf16, float32 accumulation; float32-class results.  `packed_x3` = pg_conv2d_up2x3_pack_weight of `packed`
 * (pg_conv2d_up2x3_packed_size(Cout, Cin) bytes), made once per weight version.  Serves W > 16, W % 4 == 0, Cin % 16 == 0, 16-byte aligned x
 * (PG_ERR_UNSUPPORTED otherwise: callers use pg_conv2d_up2_forward); the last output column / row is computed by the fp32 kernel's edge pass. */
PG_EXPORT int64_t pg_conv2d_up2x3_packed_size(int Cout, int Cin) {
    if (Cout <= 0 || Cin <= 0 || Cin % 16 != 0) return 0;
    return (int64_t)((Cout + 31) / 32) * (Cin / 16) * pgconv::UX_WW * 16;
}

PG_EXPORT int pg_conv2d_up2x3_pack_weight(const float* packed, void* packed_x3, int Cout, int Cin, void* stream) {
    if (!packed || !packed_x3 || Cout <= 0 || Cin <= 0) return PG_ERR_INVALID_ARG;
    if (Cin % 16 != 0) return PG_ERR_UNSUPPORTED;
    const int CoutP = (Cout + 31) / 32 * 32;
    const int64_t total = (int64_t)Cin * 9 * CoutP;
    int64_t blocks = (total + 255) / 256;
    if (blocks > (int64_t)pg::max_stream_blocks()) blocks = pg::max_stream_blocks();
    hipLaunchKernelGGL(pgconv::up2x3_pack_kernel, dim3((unsigned)blocks), dim3(256), 0, (hipStream_t)stream, packed, (unsigned short*)packed_x3, Cin, CoutP, CoutP);
    return pg::launch_status();
}

PG_EXPORT int pg_conv2d_up2x3_forward(const float* x, const float* packed, const void* packed_x3, float* y, int N, int Cin, int H, int W, int Cout,
                                      const int64_t ystride[4], const float* in_scale, const float* out_scale, float* edge_column, void* stream) {
    if (!x || !packed || !packed_x3 || !y || !ystride || N <= 0 || Cin <= 0 || H <= 0 || W <= 0 || Cout <= 0) return PG_ERR_INVALID_ARG;
    if ((((uintptr_t)packed) & 15) != 0 || (((uintptr_t)packed_x3) & 15) != 0) return PG_ERR_INVALID_ARG;
    if ((int64_t)Cin * H * W * 4 > 0x7fffffffLL) return PG_ERR_TOO_LARGE;
    pgconv::Up2Params p;
    p.x = x; p.wp = packed; p.y = y; p.in_scale = in_scale; p.out_scale = out_scale;
    p.N = N; p.Cin = Cin; p.H = H; p.W = W; p.Cout = Cout; p.CoutP = (Cout + 31) / 32 * 32;
    for (int i = 0; i < 4; i++) p.ys[i] = ystride[i];
    p.ksplit = 1; p.cpk = 0; p.ws_slice = 0;
    return pgconv::launch_up2x3(p, packed_x3, edge_column, (hipStream_t)stream);
}
