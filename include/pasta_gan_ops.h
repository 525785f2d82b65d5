/*
 * pasta_gan_ops.h -- C ABI of the MI355X (gfx950) kernels behind PASTA-GAN++'s
 * generator-synthesis operator API.
 *
 * Four shared libraries export these symbols (one per reference "plugin", plus the loader's patch routing):
 *   bias_act_plugin.so      pg_bias_act
 *   upfirdn2d_plugin.so     pg_upfirdn2d, pg_upfirdn2d_bias_act, pg_upfirdn2d_with_odd_samples
 *   conv2d_plugin.so        fp32: pg_conv2d_{packed_size,pack_weight,forward,splitk_plan,forward_splitk}, pg_conv2d_winograd_*,
 *                           pg_conv2d_up2_{forward,splitk_plan,forward_splitk}, pg_conv1x1_small, pg_conv3x3_cin1, pg_conv2d_wgrad{_plan,}, pg_split3_bf16_cl;
 *                           16-bit: pg_conv2d16_{packed_size,pack_weight,pack_weight_grouped,forward,splitk_plan,forward_splitk,up2_fused,wgrad,wgrad_plan,wgrad_x3}, pg_adam_flat_{chunk,step},
 *                           pg_conv1x1_small16;  glue: pg_modconv_{dcoefs,w2,prep}, pg_instance_norm_stats, pg_spade_*
 *   patch_routing_plugin.so pg_warp_perspective_u8, pg_patch_compose_u8
 * plus pg_<plugin>_abi_version() in each.  They are what the reference's L1
 * Python ops bind in place of its pybind plugins (see INTEGRATION.md for the
 * ctypes stub a maintainer adds to the reference tree).
 *
 * Conventions (all entry points)
 *   - plain pointers to DEVICE memory + sizes; no torch types; never throws.
 *   - asynchronous: work is enqueued on `stream` (a hipStream_t passed as void*;
 *     NULL = the null stream); no allocation, no synchronisation inside.
 *   - return 0 on success; a negative PG_ERR_* for rejected arguments (nothing
 *     was launched); a positive hipError_t if the launch itself failed.
 *   - strides are in ELEMENTS, not bytes.
 */
#ifndef PASTA_GAN_OPS_H
#define PASTA_GAN_OPS_H

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define PG_ABI_VERSION 13

enum pg_dtype { PG_F32 = 0, PG_F16 = 1, PG_BF16 = 2, PG_F64 = 3 };

enum pg_error {
    PG_OK = 0,
    PG_ERR_INVALID_ARG = -1,   /* NULL pointer, non-positive size, bad enum     */
    PG_ERR_UNSUPPORTED = -2,   /* valid request the kernels do not cover         */
    PG_ERR_TOO_LARGE = -3      /* index range exceeds the kernel's 32-bit maths  */
};

/* Activation indices: identical to the reference's cuda_idx (bias_act.py:23-33). */
enum pg_act {
    PG_ACT_LINEAR = 1, PG_ACT_RELU = 2, PG_ACT_LRELU = 3, PG_ACT_TANH = 4, PG_ACT_SIGMOID = 5,
    PG_ACT_ELU = 6, PG_ACT_SELU = 7, PG_ACT_SOFTPLUS = 8, PG_ACT_SWISH = 9
};

/* ------------------------------------------------------------------------
 * bias_act  -- replaces bias_act_plugin.bias_act (torch_utils/ops/bias_act.cpp:32-90,
 * kernel bias_act.cu:23-147).
 *
 *   grad == 0:  y = clamp(act(x + b[(i / stepB) % sizeB]) * gain)
 *   grad == 1:  x holds dy;          y = dy * act'(.) * gain, zeroed where |yref| >= clamp
 *   grad == 2:  x holds d_dx, `dy` the first-order upstream gradient; second derivative
 *   xref / yref: forward input / output as the activation's backward needs them
 *   (bias_act.py:23-33 column `ref`); NULL where the reference passes an empty tensor.
 *   All tensors share x's dense layout; `b` has sizeB contiguous elements of x's dtype.
 *   clamp < 0 disables clamping.
 */
int pg_bias_act(const void* x, const void* b, const void* xref, const void* yref, const void* dy, void* y,
                int dtype, int64_t sizeX, int sizeB, int64_t stepB,
                int grad, int act, float alpha, float gain, float clamp, void* stream);
int pg_bias_act_abi_version(void);

/* First-derivative form with the bias gradient gathered in the same pass (training route): dx as grad == 1 above (NULL: not stored),
 * db[c] = dx summed over every element of channel c, written in the tensors' dtype.  Replaces the pair "plugin call + dx.sum(...)" of
 * BiasActCuda.backward (torch_utils/ops/bias_act.py:176-186).  Covered: float32 / float16 / bfloat16; act linear, relu, lrelu (the
 * derivatives that need only yref); layouts [outer][sizeB][stepB] with stepB a multiple of 16 bytes, or stepB == 1 with sizeB / (16 bytes)
 * dividing 256.  `workspace`: at least pg_bias_act_grad_bias_workspace(...) bytes of device memory (0 = layout not covered); the sum is a
 * fixed-order fold of per-workgroup partials -- bit-identical from run to run.  PG_ERR_UNSUPPORTED: not covered / unaligned. */
int64_t pg_bias_act_grad_bias_workspace(int dtype, int64_t sizeX, int sizeB, int64_t stepB);
int pg_bias_act_grad_bias(const void* dy, const void* yref, void* dx, void* db, void* workspace, int64_t workspace_bytes,
                          int dtype, int64_t sizeX, int sizeB, int64_t stepB,
                          int act, float alpha, float gain, float clamp, void* stream);

/* ------------------------------------------------------------------------
 * upfirdn2d -- replaces upfirdn2d_plugin.upfirdn2d (torch_utils/ops/upfirdn2d.cpp:16-94,
 * kernels upfirdn2d.cu:29-200).
 *
 *   y[n,c,oy,ox] = gain * sum_{ky,kx} g[ky,kx] * xs[oy*downy + ky - pady0, ox*downx + kx - padx0]
 *   xs = x zero-stuffed by (upx, upy), zero outside; g = f flipped in both axes unless `flip`.
 *   outW = (inW*upx + padx0 + padx1 - fw + downx) / downx is computed by the CALLER
 *   (upfirdn2d.cpp:32-33) and passed in; x/y strides are (n, c, h, w) in elements, any layout;
 *   f is float32 [fh, fw] with its own strides.
 */
int pg_upfirdn2d(const void* x, const float* f, void* y, int dtype,
                 int N, int C, int inH, int inW, const int64_t xstride[4],
                 int fh, int fw, const int64_t fstride[2],
                 int outH, int outW, const int64_t ystride[4],
                 int upx, int upy, int downx, int downy, int padx0, int pady0,
                 int flip, float gain, void* stream);
/* Round 6 -- pg_upfirdn2d (no resampling) that also writes the odd rows and columns of its output as a dense tensor: y_odd[n, c, j, i] = y[n, c, 2 j + 1, 2 i + 1],
 * [N, C, (outH - 1) / 2, (outW - 1) / 2].  A ResBlock with down = 2 filters its input twice (conv2d_resample.py:119-122 with padding 2 in front of the strided 3x3
 * convolution, :107-110 with padding 1 and every second sample in front of the 1x1 skip convolution): the second result is exactly these samples of the first.
 * float32, dense NCHW, the tiled kernel's filter sizes; anything else PG_ERR_UNSUPPORTED (callers then make the two pg_upfirdn2d calls). */
int pg_upfirdn2d_with_odd_samples(const void* x, const float* f, void* y, float* y_odd, int dtype,
                                  int N, int C, int inH, int inW, const int64_t xstride[4],
                                  int fh, int fw, const int64_t fstride[2],
                                  int outH, int outW, const int64_t ystride[4],
                                  int padx0, int pady0, int flip, float gain, void* stream);
/* upfirdn2d followed, in the same pass, by the tail of a SynthesisLayer: v = fir(x) * gain + noise * noise_gain + bias[c];
 * y = clamp(act(v) * act_gain) with act in {linear, relu, lrelu}.  float32, dense NCHW, the filter sizes / factors of the
 * tiled kernel only (PG_ERR_UNSUPPORTED otherwise -- callers then run pg_upfirdn2d and pg_bias_act separately). */
typedef struct pg_fir_epilogue {
    const float* noise;         /* [outH, outW] (noise_batch_stride 0) or [N, outH, outW]; NULL = none */
    int64_t      noise_batch_stride;
    float        noise_gain;
    const float* bias;          /* [C]; NULL = none */
    int          act;           /* pg_act */
    float        alpha;
    float        act_gain;      /* 0 => 1 */
    float        clamp;         /* < 0 = off */
} pg_fir_epilogue;
int pg_upfirdn2d_bias_act(const void* x, const float* f, void* y, int dtype,
                          int N, int C, int inH, int inW, const int64_t xstride[4],
                          int fh, int fw, const int64_t fstride[2],
                          int outH, int outW, const int64_t ystride[4],
                          int upx, int upy, int downx, int downy, int padx0, int pady0,
                          int flip, float gain, const pg_fir_epilogue* epilogue, void* stream);
int pg_upfirdn2d_abi_version(void);

/* ------------------------------------------------------------------------
 * conv2d (new: the reference has no native conv; its arithmetic is cuDNN's behind
 * conv2d_gradfix.py:35-43 / conv2d_resample.py:29-54).  fp32 NCHW implicit GEMM on
 * v_mfma_f32_32x32x2_f32 with fused prologue/epilogue.
 *
 * Weight packing: OIHW [Cout, Cin, KH, KW] -> the kernel's [CinP][KH*KW][CoutP] layout
 * (CinP = Cin rounded up to 16, CoutP = Cout rounded up to 32, zero filled), scaled by
 * `scale` (the layers' runtime weight_gain, networks.py:171) and optionally flipped in
 * both spatial axes (flip_weight=False in conv2d_resample.py:34-35) or transposed
 * O<->I (the conv_transpose2d weight view, conv2d_resample.py:127).
 * pg_conv2d_packed_size returns the number of floats the packed buffer needs.
 */
int64_t pg_conv2d_packed_size(int Cout, int Cin, int KH, int KW);
int pg_conv2d_pack_weight(const float* w, float* packed, int Cout, int Cin, int KH, int KW,
                          float scale, int flip_hw, int transpose_oi, void* stream);

/* Optional fused stages of pg_conv2d_forward; every pointer may be NULL (= stage off). */
typedef struct pg_conv2d_fusion {
    /* prologue, applied to each in-image input element before the contraction
       (zero padding stays zero):  x' = clamp(in_act(x * in_scale[n,ci]) * in_gain); in_gain > 0, in_act in {linear, relu, lrelu} */
    const float* in_scale;      /* [N, Cin]  style modulation (networks.py:74)                  */
    const float* in_bias;       /* [Cin]     must be NULL: a pre-activation bias would turn the zero padding into act(b)
                                             (PG_ERR_UNSUPPORTED); none of the generator's SPADE convs has one          */
    int          in_act;        /* pg_act; 0 or PG_ACT_LINEAR = none                             */
    float        in_alpha;
    float        in_gain;       /* used only when in_act/in_bias/in_gain stage is on (0 => 1)    */
    float        in_clamp;      /* < 0 = off                                                     */
    /* epilogue:  v = acc * out_scale[n,co] + noise[...] * noise_gain + bias[co];
                  v = clamp(act(v) * gain);  y = v + residual * residual_gain                    */
    const float* out_scale;     /* [N, Cout] demodulation coefficients (networks.py:68,77)       */
    const float* noise;         /* [OH, OW] (noise_batch_stride = 0) or [N, OH, OW]              */
    int64_t      noise_batch_stride;
    float        noise_gain;
    const float* bias;          /* [Cout]                                                        */
    int          act;           /* pg_act                                                        */
    float        alpha;
    float        gain;          /* 0 => 1 */
    float        clamp;         /* < 0 = off */
    const float* residual;      /* same shape/strides as y: y += residual (ResBlock add, img.add_) */
    /* SPADE combine mode (all three set, Cout = 2*C packed as alternating blocks of 32 gamma rows / 32 beta rows of the
       same channels): y[n,c] = (spade_x[n,c] - spade_mean[n,c]) * spade_rstd[n,c] * (1 + gamma[n,c]) + beta[n,c]
       (Spade_Norm_Block, networks.py:1715-1722); y and spade_x are [N, C, OH, OW] with the strides passed for y;
       act / alpha / gain / clamp then apply to that result (the pre-activation of the Spade_Conv2dLayer consuming it,
       networks.py:1627-1633); out_scale / noise / bias / residual are ignored in this mode;
       the other epilogue stages are not applied. */
    const float* spade_x;
    const float* spade_mean;    /* [N, C] */
    const float* spade_rstd;    /* [N, C] */
    /* Second input (channel concatenation without the copy): input channels [cin_split, Cin) are read from x2
       ([N, Cin - cin_split, H, W], contiguous) instead of x ([N, cin_split, H, W]) -- the `torch.cat([x, feat], 1)` in
       front of merge_conv (networks.py:2179-2181).  cin_split must be a multiple of 16. */
    const float* x2;
    int          cin_split;
    /* Instance-norm statistics of the OUTPUT, gathered where it is produced (round 4): when set, every workgroup tile also writes the sum and M2
       (the sum of squared deviations from the tile's own mean; round 5, was the sum of squares) of its in-image outputs per output channel to stats_partial[((n * Cout + co) * T + t) * 2 + {0, 1}], T = the tile count of one
       image (pg_conv2d_winograd4_stats_tiles), t = the tile's index; pg_instance_norm_finish turns them into mean / rstd (fixed order: deterministic).
       The values are those written to y (after the epilogue).  Only pg_conv2d_winograd4_forward launches with the plain tail (no in_scale, noise,
       residual, SPADE) gather them; every other launch with this field set is declined (PG_ERR_UNSUPPORTED).  The SPADE res-blocks normalise the
       output of such a convolution (networks.py:1896-1904, 1715-1723): the separate statistics pass over it (pg_instance_norm_stats) is what this removes. */
    float*       stats_partial;
} pg_conv2d_fusion;

/*
 * y[n, co, oy*osy + ooy, ox*osx + oox] (+)= sum_{ci,ky,kx} wp[ci][ky*KW+kx][co] * x'[n, ci, oy*stride + ky - pad_y, ox*stride + kx - pad_x]
 * for oy in [0, OH), ox in [0, OW).  (osy, osx, ooy, oox) = (1,1,0,0) for ordinary convs; a stride-2
 * transposed conv is issued as up to four such calls, one per output phase (see conv2d_gradfix.py of
 * this package).  x: [N, Cin, H, W] contiguous; y: strides given in elements.
 */
int pg_conv2d_forward(const float* x, const float* packed_w, float* y,
                      int N, int Cin, int H, int W, int Cout, int KH, int KW,
                      int stride, int pad_y, int pad_x, int OH, int OW,
                      const int64_t ystride[4], int out_step_y, int out_step_x, int out_off_y, int out_off_x,
                      const pg_conv2d_fusion* fusion, void* stream);

/*
 * Split-K form of pg_conv2d_forward for launches with too few output tiles to occupy the chip (the 8x8 .. 32x32 layers of the
 * style branch: a workgroup's K loop is then a serial chain of 512 channels).  `ksplit` workgroups per tile each reduce
 * Cin / ksplit channels into their own slice of `workspace` (ksplit * N * Cout * OH * OW floats, caller-provided: this library
 * never allocates), then one elementwise pass sums the slices in fixed order and applies the fused epilogue.
 * pg_conv2d_splitk_plan returns the ksplit this library would choose (1 = use pg_conv2d_forward); not for spade_x / x2 launches.
 */
int pg_conv2d_splitk_plan(int N, int Cin, int OH, int OW, int Cout, int KH, int KW, int stride);
int pg_conv2d_forward_splitk(const float* x, const float* packed_w, float* y,
                             int N, int Cin, int H, int W, int Cout, int KH, int KW,
                             int stride, int pad_y, int pad_x, int OH, int OW,
                             const int64_t ystride[4], int out_step_y, int out_step_x, int out_off_y, int out_off_x,
                             const pg_conv2d_fusion* fusion, float* workspace, int ksplit, void* stream);

/*
 * Winograd F(2x2, 3x3) variant for KH = KW = 3, stride 1, out_step 1 (the bulk of the synthesis network): the same
 * result as pg_conv2d_forward up to fp32 summation order (~2e-6 of the output scale) with 2.25x fewer multiplies.
 * Weights are pre-transformed once (U = G g G^T, 16 * CinP * CoutP64 floats, CoutP64 = Cout rounded up to 64) by
 * pg_conv2d_winograd_pack_weight -- same scale / flip_hw / transpose_oi meaning as pg_conv2d_pack_weight.
 * Fusion: every stage of pg_conv2d_fusion except x2 (PG_ERR_UNSUPPORTED; use pg_conv2d_forward); SPADE mode uses
 * the same 32/32 gamma/beta row interleave as pg_conv2d_forward.  pad_x must be in [0, 4].
 */
int64_t pg_conv2d_winograd_packed_size(int Cout, int Cin);
int pg_conv2d_winograd_pack_weight(const float* w, float* packed, int Cout, int Cin,
                                   float scale, int flip_hw, int transpose_oi, void* stream);
int pg_conv2d_winograd_forward(const float* x, const float* packed_u, float* y,
                               int N, int Cin, int H, int W, int Cout, int pad_y, int pad_x, int OH, int OW,
                               const int64_t ystride[4], const pg_conv2d_fusion* fusion, void* stream);

/*
 * Winograd F(4x4, 3x3) variant (round 3; csrc/conv2d_wino4.h): 2.25 multiplies per output instead of 4 -- the same call
 * contract as the F(2x2) entry points above, its own weight layout (36 * CinP * CoutP64 floats, transformed in float64 and
 * rounded once).  Declines (PG_ERR_UNSUPPORTED; callers use pg_conv2d_winograd_forward) unless W % 4 == 0, x is 16-byte
 * aligned, pad_x in [0, 4], no x2 and no input pre-activation stage (in_scale is supported).  float32 error ~4x that of the
 * F(2x2) kernel: max-abs 3.7e-5 on the config-2 image against the direct float32 network (tools/f43_error_probe.py).
 */
int64_t pg_conv2d_winograd4_packed_size(int Cout, int Cin);
int pg_conv2d_winograd4_pack_weight(const float* w, float* packed, int Cout, int Cin,
                                    float scale, int flip_hw, int transpose_oi, void* stream);
int pg_conv2d_winograd4_forward(const float* x, const float* packed_u, float* y,
                                int N, int Cin, int H, int W, int Cout, int pad_y, int pad_x, int OH, int OW,
                                const int64_t ystride[4], const pg_conv2d_fusion* fusion, void* stream);

/*
 * The same F(4x4, 3x3) algorithm with TWO workgroups per CU (round 4; csrc/conv2d_wino4b.h, v_mfma_f32_16x16x4_f32, 64 couts x 16
 * tiles per workgroup): same call contract, acceptance rules and numerics as pg_conv2d_winograd4_forward, its own weight stream
 * order (pg_conv2d_winograd4_packed_size floats).  SPADE combine mode expects gamma / beta rows packed as ADJACENT cout pairs
 * (row 2c = gamma of channel c, row 2c + 1 = beta), not as blocks of 32.
 */
int pg_conv2d_winograd4b_pack_weight(const float* w, float* packed, int Cout, int Cin,
                                     float scale, int flip_hw, int transpose_oi, void* stream);
int pg_conv2d_winograd4b_forward(const float* x, const float* packed_u, float* y,
                                 int N, int Cin, int H, int W, int Cout, int pad_y, int pad_x, int OH, int OW,
                                 const int64_t ystride[4], const pg_conv2d_fusion* fusion, void* stream);

/*
 * The F(4x4, 3x3) one-workgroup kernel with its transform-domain GEMM on the bf16 matrix pipe (round 6; csrc/conv2d_wino4.h, X3 form): every float32 operand
 * of that GEMM is the exact sum of three bf16 values (truncation split, 8 + 8 + 8 significand bits) and the product is evaluated as the six largest of the nine
 * plane products on v_mfma_f32_32x32x16_bf16 with float32 accumulation -- float32-class results (the three dropped products are below 2^-24 of the product),
 * at 6/16 of the fp32 MFMA's issue time.  The transformed weights are the float32 values of pg_conv2d_winograd4_pack_weight, split once per weight version:
 * pg_conv2d_winograd4x3_packed_size float32 units (6 bytes per transformed weight).  Same call contract, acceptance rules, fused stages and statistics as
 * pg_conv2d_winograd4_forward.
 */
int64_t pg_conv2d_winograd4x3_packed_size(int Cout, int Cin);
int pg_conv2d_winograd4x3_pack_weight(const float* w, float* packed, int Cout, int Cin,
                                      float scale, int flip_hw, int transpose_oi, void* stream);
int pg_conv2d_winograd4x3_forward(const float* x, const float* packed_u, float* y,
                                  int N, int Cin, int H, int W, int Cout, int pad_y, int pad_x, int OH, int OW,
                                  const int64_t ystride[4], const pg_conv2d_fusion* fusion, void* stream);

/* Statistics gathered by pg_conv2d_winograd4_forward (pg_conv2d_fusion::stats_partial): tiles of one image for an OH x OW output, and the
 * reduction of the per-tile (sum, M2) pairs into mean[n*C + c] and rstd = 1 / sqrt(var + eps) (biased variance over OH*OW): Chan's pairwise merge in
 * float64, tiles in index order per lane, then a fixed-shape wave reduction -- deterministic, and not E[x^2] - E[x]^2.  T = pg_conv2d_winograd4_stats_tiles(OH, OW). */
int pg_conv2d_winograd4_stats_tiles(int OH, int OW);
int pg_instance_norm_finish(const float* stats_partial, float* mean, float* rstd, int NC, int T, int OH, int OW, float eps, void* stream);

/* Demodulation coefficients of modulated_conv2d (networks.py:64-68):
 *   dcoefs[n,o] = rsqrt(sum_{i,k} (w[o,i,k] * scale * styles[n,i])^2 + 1e-8);  w is OIHW. */
int pg_modconv_dcoefs(const float* w, const float* styles, float* dcoefs,
                      int N, int Cout, int Cin, int KHW, float scale, void* stream);

/* The same coefficients in two steps, for callers that keep weights across calls (eval): the tap energy
 *   w2[o,i] = scale^2 * sum_k w[o,i,k]^2        (pg_modconv_w2: weights only, once per weight version)
 * and the per-call style preparation (pg_modconv_prep, one launch):
 *   normalize != 0: smax[n] = max(max_i |styles[n,i]|, 1e-20), s' = styles / smax (the half-precision pre-normalisation of
 *                   networks.py:57-59); s_norm (float32 [N,Cin]) and s16 (bf16 / fp16 [N,Cin], half_dtype) receive s' when non-NULL
 *   normalize == 0: s' = styles
 *   out[n,o] = demodulate ? rsqrt(sum_i w2[o,i] * s'[n,i]^2 + 1e-8) : smax[n]. */
int pg_modconv_w2(const float* w, float* w2, int Cout, int Cin, int KHW, float scale, void* stream);
int pg_modconv_prep(const float* w2, const float* styles, float* out, float* s_norm, void* s16, int half_dtype,
                    int N, int Cout, int Cin, int normalize, int demodulate, void* stream);

/* pg_modconv_prep for every modulated convolution of a network in one launch (the per-layer launches are ~8 us each, 15 per step of the 1024^2 stack).
 * Job j = one pg_modconv_prep call: flags bit 0 = normalize, bit 1 = demodulate; `half_dtype` (PG_BF16 | PG_F16, or 0 when no job has an s16) is shared.
 * The table is passed BY VALUE to the kernel (a captured hipGraph keeps it; nothing is copied to the device per call). */
#define PG_MODCONV_PREP_MAX_JOBS 32
typedef struct {
    const float* w2[PG_MODCONV_PREP_MAX_JOBS];
    const float* styles[PG_MODCONV_PREP_MAX_JOBS];
    float*       out[PG_MODCONV_PREP_MAX_JOBS];
    float*       s_norm[PG_MODCONV_PREP_MAX_JOBS];
    void*        s16[PG_MODCONV_PREP_MAX_JOBS];
    int          cout[PG_MODCONV_PREP_MAX_JOBS];
    int          cin[PG_MODCONV_PREP_MAX_JOBS];
    int          flags[PG_MODCONV_PREP_MAX_JOBS];
    int          njobs;
    int          half_dtype;
} pg_modconv_prep_jobs;
int pg_modconv_prep_batched(const pg_modconv_prep_jobs* jobs, int N, void* stream);

/* Stride-2 transposed 3x3 convolution, all four output parities in one launch (csrc/conv2d_up2.h) -- the `up = 2` layers:
 * conv2d_gradfix.conv_transpose2d(stride=2, padding=0) behind conv2d_resample.py:125-142, with the modulation / demodulation of
 * networks.py:73-94 around it:
 *   y[n, co, 2 iy + ky, 2 ix + kx] += x[n, ci, iy, ix] * in_scale[n, ci] * w[co, ci, ky, kx];   y *= out_scale[n, co]
 * `packed` = pg_conv2d_pack_weight of the OIHW 3x3 kernel w.  y is [N, Cout, 2H+1, 2W+1] with strides `ystride` (elements; an even row
 * pitch gives 8-byte stores).  Every element of y is written (since ABI 6; before, the caller computed the last column ox = 2W itself).
 * in_scale / out_scale may be NULL. */
int pg_conv2d_up2_forward(const float* x, const float* packed, float* y, int N, int Cin, int H, int W, int Cout,
                          const int64_t ystride[4], const float* in_scale, const float* out_scale, void* stream);
/* Split-K form (ABI 11) for the low-resolution layers, whose few tiles would each walk all K chunks serially: `pg_conv2d_up2_splitk_plan` = the share
 * count the launch wants (1: call pg_conv2d_up2_forward).  Every tile then exists `ksplit` times, share z reduces its part of the input channels into slice z of
 * `workspace` (ksplit * N * ystride[0] floats, 16-byte aligned, laid out like y: y must be dense over n and N * ystride[0] % 4 == 0), and one pass adds the
 * slices into y in a fixed order. */
int pg_conv2d_up2_splitk_plan(int N, int Cin, int H, int W, int Cout);
int pg_conv2d_up2_forward_splitk(const float* x, const float* packed, float* y, int N, int Cin, int H, int W, int Cout,
                                 const int64_t ystride[4], const float* in_scale, const float* out_scale, float* workspace, int ksplit, void* stream);

/* Round 6 -- the same layer with its multiplies on the bf16 matrix pipe (csrc/conv2d_up2x3.h): every float32 operand the exact sum of three bf16 values
 * (truncation split), a float32 product = the six largest plane products on v_mfma_f32_32x32x16_bf16, float32 accumulation -- float32-class results at 6/16 of
 * the fp32 MFMA's issue time.  `packed` as above (its edge pass computes the last output column / row), `packed_x3` = pg_conv2d_up2x3_pack_weight(packed),
 * pg_conv2d_up2x3_packed_size(Cout, Cin) BYTES, once per weight version.  Serves W > 16, W % 4 == 0, Cin % 16 == 0, x 16-byte aligned; anything else
 * returns PG_ERR_UNSUPPORTED (callers then use pg_conv2d_up2_forward).  `edge_column`: N * Cin * H floats of scratch (the main kernel leaves input column W - 1
 * there for the kernel that makes the last output column / row, csrc/conv2d_up2_edges.h), or NULL (that kernel gathers the column from x: slower). */
int64_t pg_conv2d_up2x3_packed_size(int Cout, int Cin);
int pg_conv2d_up2x3_pack_weight(const float* packed, void* packed_x3, int Cout, int Cin, void* stream);
int pg_conv2d_up2x3_forward(const float* x, const float* packed, const void* packed_x3, float* y, int N, int Cin, int H, int W, int Cout,
                            const int64_t ystride[4], const float* in_scale, const float* out_scale, float* edge_column, void* stream);

/* Round 6 -- the 7x7 three-channel stem of the garment encoder (networks.py:2233-2238: Conv2dLayer(3, 64, kernel_size=7) = conv2d_resample.py:145-147 ->
 * conv2d_gradfix.conv2d, bias_act in the epilogue) on the bf16 matrix pipe with the three-term operand split (csrc/conv2d_stem7x3.h): float32-class results.
 * x [N, 3, H, W] contiguous, stride 1, padding 3; `packed` = pg_conv2d_stem7x3_pack_weight of the OIHW kernel w [Cout, 3, 7, 7] * scale (flip_hw as
 * pg_conv2d_pack_weight), pg_conv2d_stem7x3_packed_size(Cout) BYTES, once per weight version.  Fusion: bias, act in {linear, relu, lrelu}, gain, clamp;
 * any other stage set -> PG_ERR_UNSUPPORTED (callers then use pg_conv2d_forward). */
int64_t pg_conv2d_stem7x3_packed_size(int Cout);
int pg_conv2d_stem7x3_pack_weight(const float* w, void* packed, int Cout, float scale, int flip_hw, void* stream);
int pg_conv2d_stem7x3_forward(const float* x, const void* packed, float* y, int N, int H, int W, int Cout, const int64_t ystride[4],
                              const pg_conv2d_fusion* fusion, void* stream);

/* Streaming 1x1 convolution with few output channels (Cout <= 8; the ToRGB / parsing heads, networks.py:287-316,
 * modulated_conv2d with demodulate=False at networks.py:37-94): float32 NCHW, HW % 4 == 0, 16-byte aligned tensors:
 *   y[n,o,p] = clamp(sum_c x[n,c,p] * w[o,c] * scale * styles[n,c] + bias[o]) + skip[n,o,p]
 * styles / bias / skip may be NULL; clamp < 0 disables it.  PG_ERR_UNSUPPORTED otherwise (use pg_conv2d_forward). */
int pg_conv1x1_small(const float* x, const float* w, const float* styles, const float* bias, const float* skip, float* y,
                     int N, int Cin, int64_t HW, int Cout, float scale, float clamp, void* stream);

/* 3x3 convolution of a one-channel image (first layer of the SPADE blocks on the parsing / mask map, networks.py:1708-1712):
 * y = act(conv2d(x [N,1,H,W], w [Cout,1,3,3] * scale, padding=1)), cross-correlation as F.conv2d; act = PG_ACT_LINEAR | PG_ACT_RELU
 * (gain 1).  float32, W % 4 == 0, 16-byte aligned; PG_ERR_UNSUPPORTED otherwise (use pg_conv2d_forward). */
int pg_conv3x3_cin1(const float* x, const float* w, float* y, int N, int H, int W, int Cout, float scale, int act, void* stream);

/* Per-(n,c) mean and rsqrt(var + eps) over H*W (nn.InstanceNorm2d(affine=False), networks.py:1713),
 * and the SPADE combine out = (x - mean) * rstd * (1 + gamma) + beta (networks.py:1722). */
int pg_instance_norm_stats(const float* x, float* mean, float* rstd, int NC, int64_t HW, float eps, void* stream);
int pg_spade_norm(const float* x, const float* mean, const float* rstd, const float* gamma, const float* beta,
                  float* y, int NC, int64_t HW, void* stream);

/* The same combine on the TRAINING route, with its gradient (autograd of networks.py:1715-1723: instance norm, then x_hat (1 + gamma) + beta).
 * x, y, dy, dx: [N, C, HW] contiguous.  gamma / beta (and dgamma / dbeta) may be planes of a larger tensor -- the two 3x3 convolutions that
 * produce them run as one launch over the stacked weights, output [N, 2C, HW]: plane (n, c) starts at base + n * sample_stride + c * HW.
 *   forward:   y = (x - mean) rstd (1 + gamma) + beta                      (mean / rstd from pg_instance_norm_stats)
 *   backward:  dbeta = dy,  dgamma = dy x_hat,  g = dy (1 + gamma),  dx = rstd (g - mean_hw(g) - x_hat mean_hw(g x_hat))
 * `sums`: 2 * N * C floats of scratch (the two plane means; fixed-order workgroup reductions: deterministic).  dx or the (dgamma, dbeta)
 * pair may be NULL when not needed. */
int pg_spade_train_forward(const float* x, const float* mean, const float* rstd, const float* gamma, const float* beta, float* y,
                           int N, int C, int64_t HW, int64_t gamma_sample_stride, int64_t beta_sample_stride, void* stream);
int pg_spade_train_backward(const float* dy, const float* x, const float* mean, const float* rstd, const float* gamma, float* sums,
                            float* dx, float* dgamma, float* dbeta, int N, int C, int64_t HW,
                            int64_t gamma_sample_stride, int64_t dgamma_sample_stride, int64_t dbeta_sample_stride, void* stream);

/* Garment-feature assembly of the synthesis network (get_spade_feat + the merge, networks.py:2253-2276, 2311-2316) in two
 * launches instead of ~60 elementwise ones.  feat_*: [N,C,H,W]; masks: [N,1,2H,2W], sampled at the even pixels; with
 *   m = mask > 0.9,  v = m && denorm_mask > 0.9,  r = m - v,  count[n] = sum_p v,  sums[n,c] = sum_p feat * v:
 *   B   = feat * (1 - r) + sums / (count > 10 ? count : 256*256) * r
 *   out = B_upper * m_upper + B_lower * m_lower
 * pg_spade_masked_sums fills sums [N*C] and counts [N] of one branch. */
int pg_spade_masked_sums(const float* feat, const float* mask, const float* denorm_mask, float* sums, float* counts,
                         int N, int C, int H, int W, void* stream);
int pg_spade_feat_assemble(const float* feat_upper, const float* feat_lower, const float* mask_upper, const float* mask_lower,
                           const float* denorm_mask_upper, const float* denorm_mask_lower,
                           const float* sums_upper, const float* sums_lower, const float* counts_upper, const float* counts_lower,
                           float* out, int N, int C, int H, int W, void* stream);
int pg_conv2d_abi_version(void);
/* Multi-GPU training (round 5): keep `n` CUs free of this plugin's persistent grids (one workgroup per CU with most of its LDS: conv2d_wino4, conv2d_mfma16,
 * conv2d_wgrad, ...) so that RCCL's channel workgroups -- the gradient all-reduce launched from autograd hooks on a side stream, training/ddp.py -- find a CU
 * while the backward pass is still running.  0 = none.  Returns the CU count grids and split-K planners use from now on.  Process-wide, any device. */
int pg_conv2d_reserve_cus(int n);

/* Weight gradient of a float32 NCHW convolution (3x3 at stride 1 or 2, 1x1 at stride 1) -- what conv2d_gradfix.py:137-150 asks
 * aten::cudnn_convolution_backward_weight for:
 *   dw[co, ci, ky, kx] = sum_{n, oy, ox} dy[n, co, oy, ox] * x[n, ci, stride * oy + ky - pad_y, stride * ox + kx - pad_x]
 * A GEMM over pixels on the fp32 MFMA, split `splits` ways over the pixel axis with a fixed-order second pass (deterministic).
 * pg_conv2d_wgrad_plan returns the `splits` to use (0 = geometry not covered); `workspace` holds splits * KH*KW * Cout * Cin floats. */
int pg_conv2d_wgrad_plan(int N, int Cin, int OH, int OW, int Cout, int KH, int KW, int stride);
int pg_conv2d_wgrad(const float* x, const float* dy, float* dw, float* workspace,
                    int N, int Cin, int H, int W, int Cout, int KH, int KW, int stride, int pad_y, int pad_x, int OH, int OW,
                    int splits, void* stream);

/* The same gradient for 16-bit (PG_F16 | PG_BF16) CHANNELS-LAST operands -- x [N, H, W, Cin], dy [N, OH, OW, Cout] -- on
 * v_mfma_f32_32x32x16_{f16,bf16} with float32 accumulation; dw is float32 [Cout, Cin, KH, KW].  Replaces aten::convolution_backward
 * (MIOpen) behind the half-precision discriminator blocks (reference conv2d_gradfix.py:137-150).  Covered: 3x3 stride 1 | 2, 1x1 stride 1,
 * Cin and Cout multiples of 8 (callers zero-pad the image channels of `fromrgb`).  pg_conv2d16_wgrad_plan returns the `splits` to use
 * (0 = not covered); `workspace` holds splits * KH*KW * Cout * Cin floats; deterministic (fixed-order reduction of the K splits). */
int pg_conv2d16_wgrad_plan(int N, int Cin, int OH, int OW, int Cout, int KH, int KW, int stride);
int pg_conv2d16_wgrad(const void* x, const void* dy, float* dw, float* workspace, int dtype,
                      int N, int Cin, int H, int W, int Cout, int KH, int KW, int stride, int pad_y, int pad_x, int OH, int OW,
                      int splits, void* stream);

/* (ABI 11, exploratory: PG_WGRAD_BF16X3) The FLOAT32 weight gradient of pg_conv2d_wgrad on the bf16 matrix pipe by three-term operand splitting: every float32
 * value is the exact sum of three bf16 values (truncation), and of the nine partial products the six largest are accumulated in float32 -- float32-class results
 * (measured 1.3e-6 ... 2.8e-6 of max|dw| against float64; the fp32 MFMA kernel 1.0e-6 ... 1.7e-6), exact on integer data below 2^24 like the fp32 kernel.
 *   pg_split3_bf16_cl: float32 NCHW x [N, C, HW] -> out = three bf16 channels-last planes [3][N][HW][C], x == plane 0 + plane 1 + plane 2.  C % 8 == 0.
 *   pg_conv2d16_wgrad_x3: dw [Cout, Cin, KH, KW] from x3 = split(x), dy3 = split(dy): ONE launch of the 16-bit kernel over 6 N plane-mapped "images" (the six
 *   products share its float32 accumulation and fixed-order split reduction).  splits = pg_conv2d16_wgrad_plan(6 * N, ...); workspace as pg_conv2d16_wgrad. */
int pg_split3_bf16_cl(const float* x, void* out, int N, int C, int64_t HW, void* stream);
int pg_conv2d16_wgrad_x3(const void* x3, const void* dy3, float* dw, float* workspace,
                         int N, int Cin, int H, int W, int Cout, int KH, int KW, int stride, int pad_y, int pad_x, int OH, int OW,
                         int splits, void* stream);

/* ------------------------------------------------------------------------
 * 16-bit convolution (bf16 / fp16 storage, fp32 accumulation) on v_mfma_f32_32x32x16_{bf16,f16}: what cuDNN does behind
 * conv2d_gradfix.py:35-43 for the reference's half-precision blocks (discriminator `use_fp16`, networks.py:444-523;
 * the StyleGAN2 synthesis blocks with use_fp16, networks.py:2147-2194).  Activations are NHWC ("channels_last", the
 * layout the reference's --nhwc option selects): x is [N, H, W, Cin] dense with Cin % 16 == 0.
 *
 * Weights are packed once per parameter version -- and, for a modulated convolution, once per style batch -- by
 * pg_conv2d16_pack_weight into [ceil(Cin/32)*2][KH*KW][2][CoutP][8] (CoutP = Cout rounded up to 64; zero filled):
 *   packed[n][ci/16][tap][(ci/8)&1][co][ci&7] = T(w[co][ci][tap] * scale * styles[n][ci] * dcoefs[n][co])
 * with w float32 OIHW (IOHW when transpose_oi), optionally flipped in both spatial axes; styles [nsamples, Cin] /
 * dcoefs [nsamples, Cout] float32 or NULL (= 1).  This is the reference's fused modulated convolution
 * (networks.py:85-94: per-sample weights w * styles * dcoefs cast to the activation dtype) without the grouped-conv
 * reshape.  `taps_y` / `taps_x` (NULL = all) select kernel rows / columns in the given order: the gather-form
 * sub-kernels of a stride-2 transposed convolution.  pg_conv2d16_packed_size = 16-bit elements per sample.
 */
int64_t pg_conv2d16_packed_size(int Cout, int Cin, int KH, int KW);
int pg_conv2d16_pack_weight(const float* w, void* packed, int dtype, int Cout, int Cin, int KH, int KW,
                            const int* taps_y, int ntaps_y, const int* taps_x, int ntaps_x,
                            float scale, int flip_hw, int transpose_oi,
                            const float* styles, const float* dcoefs, int nsamples, void* stream);
/* The same for `ngroups` weight tensors of one shape in ONE launch (the four composite phase kernels of an up-by-2 modulated
 * convolution share styles / dcoefs): group g reads w + g * w_group_stride (floats) and writes packed[(g * nsamples + n)]. */
int pg_conv2d16_pack_weight_grouped(const float* w, void* packed, int dtype, int ngroups, int64_t w_group_stride,
                                    int Cout, int Cin, int KH, int KW, float scale, int flip_hw, int transpose_oi,
                                    const float* styles, const float* dcoefs, int nsamples, void* stream);

/* The per-sample packs (styles and / or dcoefs given) of SEVERAL 3x3 layers in one launch -- the ~7 us pg_conv2d16_pack_weight launch per modulated
 * convolution and step is what this removes.  Job j packs w[j] (OIHW, or IOHW when flags bit 1 = transpose_oi; bit 0 = flip_hw; all nine taps -- or, flags bit 2, the six taps of a 3 x 2 kernel: the stack of pg_conv2d16_up2_fused, packed size of (cout, cin, 3, 2)) into
 * packed[j] ([nsamples][pg_conv2d16_packed_size(cout, cin, 3, 3)]) scaled by scale[j] * styles[j][n, ci] * dcoefs[j][n, co % dcoefs_mod[j]] (either may
 * be NULL; dcoefs_mod < cout serves the four stacked phases of an up = 2 layer, which share one coefficient row).  The table is passed by value. */
#define PG_CONV2D16_PACK_MAX_JOBS 16
typedef struct {
    const float* w[PG_CONV2D16_PACK_MAX_JOBS];
    void*        packed[PG_CONV2D16_PACK_MAX_JOBS];
    const float* styles[PG_CONV2D16_PACK_MAX_JOBS];
    const float* dcoefs[PG_CONV2D16_PACK_MAX_JOBS];
    int          cout[PG_CONV2D16_PACK_MAX_JOBS];
    int          cin[PG_CONV2D16_PACK_MAX_JOBS];
    int          flags[PG_CONV2D16_PACK_MAX_JOBS];
    int          dcoefs_mod[PG_CONV2D16_PACK_MAX_JOBS];
    float        scale[PG_CONV2D16_PACK_MAX_JOBS];
    int          njobs;
    int          nsamples;
    int          dtype;
} pg_conv2d16_pack_jobs;
int pg_conv2d16_pack_weight_batched(const pg_conv2d16_pack_jobs* jobs, void* stream);

/* Fused epilogue of pg_conv2d16_forward:  v = acc * out_scale[n,co] + noise[n?,oy,ox] * noise_gain + bias[co];
 * v = clamp(act(v) * gain);  y = T(v + residual).  Every pointer may be NULL; all vectors are float32. */
typedef struct pg_conv2d16_fusion {
    const float* out_scale;     /* [N, Cout] */
    const float* noise;         /* [OH, OW] (noise_batch_stride 0) or [N, OH, OW] */
    int64_t      noise_batch_stride;
    float        noise_gain;
    const float* bias;          /* [Cout] */
    int          act;           /* pg_act: linear / relu / lrelu */
    float        alpha;
    float        gain;          /* 0 => 1 */
    float        clamp;         /* < 0 = off */
    const void*  residual;      /* dtype / strides of y */
    /* Four output phases in one launch (the composite kernels of an up-by-2 layer, conv2d_resample.py:125-142 with the FIR folded in):
     * phase_cout > 0 => the Cout = 4 * phase_cout output channels are four blocks; block ph = 2a + b is written as channels
     * 0 .. phase_cout-1 of output pixel (oy * out_step_y + a, ox * out_step_x + b) (out_off_* ignored); out_scale / bias have
     * phase_cout entries per row; noise is [N?, 4, OH, OW] with noise_phase_stride elements between phases.  phase_cout must be 32
     * or a multiple of 64 (PG_ERR_UNSUPPORTED otherwise); no residual. */
    int          phase_cout;
    int64_t      noise_phase_stride;
} pg_conv2d16_fusion;

/*
 * y[n, co, oy*osy + ooy, ox*osx + oox] = epilogue(sum_{ci,ky,kx} packed[n * w_sample_stride ...][ci][ky,kx][co] *
 *                                                  x[n, oy*stride + ky - pad_y, ox*stride + kx - pad_x, ci])
 * (KH, KW, stride) in {(3,3,1), (1,1,1), (2,2,1), (2,1,1), (1,2,1), (3,3,2)}.  w_sample_stride = 0 (shared weights) or
 * pg_conv2d16_packed_size (per-sample weights).  y has dtype `out_dtype` -- the activation dtype or PG_F32 -- and strides
 * ystride[4] = (n, c, y, x) in elements: channels_last 16-bit outputs with Cout % 8 == 0 are written as 16-byte vectors,
 * anything else (a float32 NCHW image from a ToRGB layer, say) element by element.
 */
int pg_conv2d16_forward(const void* x, const void* packed, void* y, int dtype, int out_dtype,
                        int N, int Cin, int H, int W, int Cout, int KH, int KW,
                        int stride, int pad_y, int pad_x, int OH, int OW, int64_t w_sample_stride,
                        const int64_t ystride[4], int out_step_y, int out_step_x, int out_off_y, int out_off_x,
                        const pg_conv2d16_fusion* fusion, void* stream);

/* Round 5 -- the reference's up-by-2 modulated 3x3 layer (conv2d_resample.py:125-142: conv_transpose2d stride 2, then upfirdn2d with the
 * separable 4-tap filter, padding 1, gain 4; then noise / bias_act, networks.py:73-94) in ONE launch without the (2H+1)^2 intermediate and with
 * half the tap-products of the composite four-phase launch above: the y half of the filter is folded into the weights on the host, the x half
 * is applied to the accumulators in the epilogue (csrc/conv2d_up2f16.h).
 *   packed: pg_conv2d16_pack_weight(transpose_oi) of the float32 stack [Cin, 4 * Cout, 3, 2], channel block p = 2a + b of it holding, for output
 *           row parity a and intermediate column parity b, tap (ty, tx) = Ky_a[ty][kx(b, tx)] with Ky = (2 fy) (*)_y w (rows 4+a, 2+a, a of the
 *           6-row result) and kx(0, .) = (2, 0), kx(1, .) = (none, 1): the plain transposed convolution along x
 *           (training/networks.py `_up2_fused_weights` builds it);  w_sample_stride = 0 or pg_conv2d16_packed_size(4 * Cout, Cin, 3, 2).
 *   fir_x:  2 * fx, applied as out[o] = sum_X fir_x[o + 2 - X] * z[X] over the 2W+1 intermediate columns.
 *   y:      [N, Cout, 2H, 2W] 16-bit with ystride[1] == 1 (channels-last), written whole.
 *   fusion: out_scale / bias [.., Cout], noise [N?, 4, H, W] phase-major (noise_phase_stride between the planes of output pixel parities
 *           (a, b) = 2a + b), act / gain / clamp as pg_conv2d16_forward; no residual; phase_cout is ignored.
 * Cin % 16 == 0, Cin >= 32, Cout % 32 == 0. */
int pg_conv2d16_up2_fused(const void* x, const void* packed, void* y, int dtype, int N, int Cin, int H, int W, int Cout,
                          int64_t w_sample_stride, const int64_t ystride[4], const float fir_x[4],
                          const pg_conv2d16_fusion* fusion, void* stream);

/* Round 5 -- the optimizer side of a training phase in one launch (csrc/optim.hip): torch.nan_to_num(grad, nan, posinf, neginf) followed by torch.optim.Adam's
 * step (training_loop_fullbody.py:632-639) over the phase's flat buffers.  p / g / m / v share one flat layout (every parameter starts on a 16-byte boundary);
 * chunks[nchunks][4] = (element offset, length <= pg_adam_flat_chunk(), parameter index, 1 for the parameter's first chunk), one workgroup each.  A parameter
 * with alive[i] <= 0 (no rank produced a gradient: `grad is None` in the reference) is left untouched and keeps its step count; steps_in / steps_out are the
 * per-parameter step counts before / after (two buffers: the chunks of a parameter read the count while its first chunk writes it).  g is cleaned in place. */
int pg_adam_flat_chunk(void);
int pg_adam_flat_step(float* p, float* g, float* m, float* v, const int* chunks, int nchunks, const float* alive, const float* steps_in, float* steps_out,
                      float lr, float beta1, float beta2, float eps, float nan_value, float posinf_value, float neginf_value, void* stream);

/* Split-K form for launches with fewer output tiles than CUs: `ksplit` launches-worth of workgroups each reduce Cin/ksplit
 * channels into float32 [ksplit][N][OH][OW][Cout] `workspace`; one pass sums the slices in fixed order, applies the
 * epilogue and writes y.  pg_conv2d16_splitk_plan returns the ksplit this library would choose (1 = do not split). */
int pg_conv2d16_splitk_plan(int N, int Cin, int OH, int OW, int Cout, int KH, int KW, int stride);
int pg_conv2d16_forward_splitk(const void* x, const void* packed, void* y, int dtype, int out_dtype,
                               int N, int Cin, int H, int W, int Cout, int KH, int KW,
                               int stride, int pad_y, int pad_x, int OH, int OW, int64_t w_sample_stride,
                               const int64_t ystride[4], int out_step_y, int out_step_x, int out_off_y, int out_off_x,
                               const pg_conv2d16_fusion* fusion, float* workspace, int ksplit, void* stream);

/* 1x1 modulated convolution to a handful of output channels (ToRGB / parsing heads, networks.py:1957-1967) as a streaming
 * dot product: y[n, o, p] = clamp(sum_c x[n, p, c] * w[o, c] * styles[n, c] + bias[o]) + skip[n, o, p], x 16-bit NHWC,
 * w float32 [Cout, Cin] (already scaled by weight_gain), y / skip float32 NCHW; Cout <= 8, Cin % 8 == 0. */
int pg_conv1x1_small16(const void* x, const float* w, const float* styles, const float* bias, const float* skip, float* y,
                       int dtype, int N, int Cin, int64_t HW, int Cout, float clamp, int skip_up2_width, void* stream);
/* skip_up2_width = 0: `skip` has y's shape.  skip_up2_width = W (the width of y, even; H = HW / W even): `skip` is the HALF-resolution image
 * [N, Cout, H/2, W/2] and is up-sampled on the fly exactly as upfirdn2d.upsample2d does with the [1, 3, 3, 1] filter (networks.py:2165-2167 with
 * upfirdn2d.py:308-342: zero insertion, padding (2, 1), gain 4) -- the skip image's own FIR launch per block disappears. */

/* ------------------------------------------------------------------------
 * patch_routing_plugin.so -- the perspective warps of the data loader's patch routing (training/dataset.py:2555-2700; there:
 * cv2.warpPerspective / cv2.erode on the DataLoader thread).  8-bit interleaved (HWC) images.
 *
 * pg_warp_perspective_u8: `njobs` independent warps in one launch.  Each job maps a destination pixel (x, y) through `minv`
 * (the INVERSE of the matrix cv2.warpPerspective is given, row major, double) with OpenCV's arithmetic: block-wise coordinate
 * evaluation (block_w = the WarpPerspectiveInvoker block width for this destination size), 5 fractional bits, 15-bit bilinear
 * weights, taps outside the source read 0 (INTER_LINEAR, BORDER_CONSTANT 0).  `jobs_device` lives in device memory.
 * pg_patch_compose_u8: canvas[p] = erode8x8(mask[..., 0])[p] == 255 ? patch[p] : canvas[p] (and the same into canvas2 if given):
 * the paste step of the de-normalisation (dataset.py:2624-2633); patch / canvas are HxWx3, mask HxWxmask_channels.
 */
typedef struct {
    const unsigned char* src;
    unsigned char*       dst;
    int    src_h, src_w, dst_h, dst_w, channels, block_w;
    double minv[9];
} pg_warp_job;
int pg_warp_perspective_u8(const pg_warp_job* jobs_device, int njobs, int max_dst_pixels, void* stream);
int pg_patch_compose_u8(const unsigned char* patch, const unsigned char* mask, unsigned char* canvas, unsigned char* canvas2,
                        int h, int w, int mask_channels, void* stream);
/* Round 6 -- the whole paste sequence of `njobs` canvases in one launch (a batch's de-normalised garments: dataset.py:2620-2633 run per part and per sample
 * in the reference): job = one h x w x 3 canvas, its parts in paste order (patch, mask; later parts overwrite earlier ones), and optionally a second canvas that
 * receives only the parts with to_canvas2 != 0.  Every pixel of the canvases is written (0 where no part's eroded mask is set).  `jobs_device` lives in device memory. */
#define PG_COMPOSE_MAX_PARTS 10
typedef struct pg_compose_job {
    unsigned char* canvas;
    unsigned char* canvas2;                              /* may be NULL */
    const unsigned char* patch[PG_COMPOSE_MAX_PARTS];    /* h x w x 3 */
    const unsigned char* mask[PG_COMPOSE_MAX_PARTS];     /* h x w x mask_channels, channel 0 is used */
    int nparts;
    int to_canvas2[PG_COMPOSE_MAX_PARTS];
    int pad_;
} pg_compose_job;
int pg_patch_compose_ordered_u8(const pg_compose_job* jobs_device, int njobs, int h, int w, int mask_channels, void* stream);
int pg_patch_routing_abi_version(void);

#ifdef __cplusplus
}
#endif
#endif /* PASTA_GAN_OPS_H */
