// bias_act for gfx950: one streaming elementwise kernel, wavefront-64, 16-byte vector
// loads/stores, grid-stride.  Drop-in for the reference's bias_act plugin
// (torch_utils/ops/bias_act.cpp:32-90, bias_act.cu:23-147): forward, first- and
// second-derivative forms of 9 activations with optional bias, gain and clamp.
//
// Roofline: pure HBM streaming. Algorithmic bytes per element = sizeof(T) * (1 read + 1 write)
// for grad 0 (plus one read per reference tensor for grad 1/2); target is the 6.3 TB/s
// achievable HBM rate (MI355X_MICROARCH.md, HBM).
#include "pg_common.h"

namespace {

using namespace pg;

struct Params {
    const void* x; const void* b; const void* xref; const void* yref; const void* dy; void* y;
    int64_t sizeX; int sizeB; int64_t stepB;
    float alpha, gain, clamp;
};

template <typename S> __device__ __forceinline__ S exp_(S v);
template <> __device__ __forceinline__ float exp_<float>(float v) { return expf(v); }
template <> __device__ __forceinline__ double exp_<double>(double v) { return exp(v); }
template <typename S> __device__ __forceinline__ S log1p_(S v);
template <> __device__ __forceinline__ float log1p_<float>(float v) { return log1pf(v); }
template <> __device__ __forceinline__ double log1p_<double>(double v) { return log1p(v); }
template <typename S> __device__ __forceinline__ S expm1_(S v);
template <> __device__ __forceinline__ float expm1_<float>(float v) { return expm1f(v); }
template <> __device__ __forceinline__ double expm1_<double>(double v) { return expm1(v); }
template <typename S> __device__ __forceinline__ S tanh_(S v);
template <> __device__ __forceinline__ float tanh_<float>(float v) { return tanhf(v); }
template <> __device__ __forceinline__ double tanh_<double>(double v) { return tanh(v); }

// One element.  G == 0: `x` is the activation input (bias already added by the caller for
// G == 0; for G >= 1 the bias is added to xref).  Returns the value before gain/clamp.
template <typename S, int A, int G>
__device__ __forceinline__ S act_core(S x, S xr, S yy, S alpha) {
    const S one = (S)1, two = (S)2;
    const S selu_s = (S)1.0507009873554804934193349852946;
    const S selu_sa = (S)(1.0507009873554804934193349852946 * 1.6732632423543772848170429916717);
    if (A == PG_ACT_LINEAR) return G == 2 ? (S)0 : x;
    if (A == PG_ACT_RELU) {
        if (G == 0) return x > 0 ? x : (S)0;
        if (G == 1) return yy > 0 ? x : (S)0;
        return (S)0;
    }
    if (A == PG_ACT_LRELU) {
        if (G == 0) return x > 0 ? x : x * alpha;
        if (G == 1) return yy > 0 ? x : x * alpha;
        return (S)0;
    }
    if (A == PG_ACT_TANH) {
        if (G == 0) return tanh_<S>(x);
        if (G == 1) return x * (one - yy * yy);
        return x * (one - yy * yy) * (-two * yy);
    }
    if (A == PG_ACT_SIGMOID) {
        if (G == 0) return x >= 0 ? one / (one + exp_<S>(-x)) : exp_<S>(x) / (one + exp_<S>(x));
        if (G == 1) return x * yy * (one - yy);
        return x * yy * (one - yy) * (one - two * yy);
    }
    if (A == PG_ACT_ELU) {
        if (G == 0) return x >= 0 ? x : expm1_<S>(x);
        if (G == 1) return yy >= 0 ? x : x * (yy + one);
        return yy >= 0 ? (S)0 : x * (yy + one);
    }
    if (A == PG_ACT_SELU) {
        if (G == 0) return x >= 0 ? selu_s * x : selu_sa * expm1_<S>(x);
        if (G == 1) return yy >= 0 ? x * selu_s : x * (yy + selu_sa);
        return yy >= 0 ? (S)0 : x * (yy + selu_sa);
    }
    if (A == PG_ACT_SOFTPLUS) {
        if (G == 0) return x > (S)20 ? x : log1p_<S>(exp_<S>(x));
        if (G == 1) return x * (one - exp_<S>(-yy));
        const S c = exp_<S>(-yy);
        return x * c * (one - c);
    }
    if (A == PG_ACT_SWISH) {
        if (G == 0) return x >= 0 ? x / (one + exp_<S>(-x)) : x * exp_<S>(x) / (one + exp_<S>(x));
        // derivatives in terms of the saved input xr; sg = sigmoid(xr)
        const S sg = xr >= 0 ? one / (one + exp_<S>(-xr)) : exp_<S>(xr) / (one + exp_<S>(xr));
        if (G == 1) return x * (sg + xr * sg * (one - sg));
        return x * sg * (one - sg) * (two + xr * (one - two * sg));
    }
    return (S)0;
}

template <typename S, int A>
__device__ __forceinline__ S swish_fwd(S xr) {
    const S one = (S)1;
    return xr >= 0 ? xr / (one + exp_<S>(-xr)) : xr * exp_<S>(xr) / (one + exp_<S>(xr));
}

template <typename T, int A, int G>
__device__ __forceinline__ T bias_act_one(T xv, T bv, T xrv, T yrv, T dyv, float alpha_f, float gain_f, float clamp_f) {
    typedef typename acc_of<T>::type S;
    const S alpha = (S)alpha_f, gain = (S)gain_f, clamp = (S)clamp_f;
    S x = (S)xv, b = (S)bv, xr = (S)xrv, yr = (S)yrv;
    if (G == 0) x += b; else xr += b;
    const S yy = (gain != (S)0) ? yr / gain : (S)0;
    S y = act_core<S, A, G>(x, xr, yy, alpha);
    y *= gain;
    if (G == 2) y *= (S)dyv;
    if (clamp >= (S)0) {
        if (G == 0) {
            y = y > clamp ? clamp : (y < -clamp ? -clamp : y);
        } else {
            if (A == PG_ACT_SWISH) yr = swish_fwd<S, A>(xr) * gain;   // swish saves x, not y (bias_act.py:32)
            y = (yr > -clamp && yr < clamp) ? y : (S)0;
        }
    }
    return (T)y;
}

// VEC elements per lane per step (16 bytes).  BVEC: every vector lies inside one bias
// run (stepB % VEC == 0), so one bias load serves the vector.
template <typename T, int A, int G, int VEC, bool BVEC>
__global__ __launch_bounds__(256) void bias_act_kernel(Params p) {
    typedef vec_t<T, VEC> V;
    const T* __restrict__ xp = (const T*)p.x;
    const T* __restrict__ bp = (const T*)p.b;
    const T* __restrict__ xrp = (const T*)p.xref;
    const T* __restrict__ yrp = (const T*)p.yref;
    const T* __restrict__ dyp = (const T*)p.dy;
    T* __restrict__ yp = (T*)p.y;
    const int64_t nvec = p.sizeX / VEC;
    const int64_t stride = (int64_t)gridDim.x * blockDim.x;
    const int64_t stepV = BVEC ? p.stepB / VEC : 1;

    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < nvec; i += stride) {
        V xv = ((const V*)xp)[i];
        V xrv, yrv, dyv, out;
        if (G > 0 && xrp) xrv = ((const V*)xrp)[i];
        if (G > 0 && yrp) yrv = ((const V*)yrp)[i];
        if (G == 2 && dyp) dyv = ((const V*)dyp)[i];
        T bs = (T)0;
        if (BVEC && bp) bs = bp[(i / stepV) % p.sizeB];
#pragma unroll
        for (int k = 0; k < VEC; k++) {
            T bv = bs;
            if (!BVEC && bp) bv = bp[((i * VEC + k) / p.stepB) % p.sizeB];
            out.v[k] = bias_act_one<T, A, G>(xv.v[k], bv, (G > 0 && xrp) ? xrv.v[k] : (T)0, (G > 0 && yrp) ? yrv.v[k] : (T)0,
                                             (G == 2 && dyp) ? dyv.v[k] : (T)1, p.alpha, p.gain, p.clamp);
        }
        ((V*)yp)[i] = out;
    }
    // tail (sizeX % VEC elements), handled by the first lanes of block 0
    const int64_t tail0 = nvec * VEC;
    if (blockIdx.x == 0 && tail0 + threadIdx.x < p.sizeX) {
        const int64_t e = tail0 + threadIdx.x;
        T bv = bp ? bp[(e / p.stepB) % p.sizeB] : (T)0;
        yp[e] = bias_act_one<T, A, G>(xp[e], bv, (G > 0 && xrp) ? xrp[e] : (T)0, (G > 0 && yrp) ? yrp[e] : (T)0,
                                      (G == 2 && dyp) ? dyp[e] : (T)1, p.alpha, p.gain, p.clamp);
    }
}

template <typename T, int A, int G>
int launch(const Params& p, hipStream_t stream) {
    constexpr int VEC = 16 / sizeof(T);
    const bool vec_ok = aligned16(p.x) && aligned16(p.y) && (!p.xref || aligned16(p.xref)) &&
                        (!p.yref || aligned16(p.yref)) && (!p.dy || aligned16(p.dy));
    const int block = 256;
    if (vec_ok) {
        const int64_t nvec = p.sizeX / VEC;
        int64_t blocks = (nvec + block - 1) / block;
        int grid = (int)(blocks < 1 ? 1 : (blocks > max_stream_blocks() ? max_stream_blocks() : blocks));
        if (!p.b || p.stepB % VEC == 0)
            hipLaunchKernelGGL((bias_act_kernel<T, A, G, VEC, true>), dim3(grid), dim3(block), 0, stream, p);
        else
            hipLaunchKernelGGL((bias_act_kernel<T, A, G, VEC, false>), dim3(grid), dim3(block), 0, stream, p);
    } else {
        int64_t blocks = (p.sizeX + block - 1) / block;
        int grid = (int)(blocks > max_stream_blocks() ? max_stream_blocks() : blocks);
        hipLaunchKernelGGL((bias_act_kernel<T, A, G, 1, true>), dim3(grid), dim3(block), 0, stream, p);
    }
    return launch_status();
}

template <typename T, int A>
int dispatch_grad(const Params& p, int grad, hipStream_t s) {
    if (grad == 0) return launch<T, A, 0>(p, s);
    if (grad == 1) return launch<T, A, 1>(p, s);
    return launch<T, A, 2>(p, s);
}

template <typename T>
int dispatch_act(const Params& p, int act, int grad, hipStream_t s) {
    switch (act) {
        case PG_ACT_LINEAR: return dispatch_grad<T, PG_ACT_LINEAR>(p, grad, s);
        case PG_ACT_RELU: return dispatch_grad<T, PG_ACT_RELU>(p, grad, s);
        case PG_ACT_LRELU: return dispatch_grad<T, PG_ACT_LRELU>(p, grad, s);
        case PG_ACT_TANH: return dispatch_grad<T, PG_ACT_TANH>(p, grad, s);
        case PG_ACT_SIGMOID: return dispatch_grad<T, PG_ACT_SIGMOID>(p, grad, s);
        case PG_ACT_ELU: return dispatch_grad<T, PG_ACT_ELU>(p, grad, s);
        case PG_ACT_SELU: return dispatch_grad<T, PG_ACT_SELU>(p, grad, s);
        case PG_ACT_SOFTPLUS: return dispatch_grad<T, PG_ACT_SOFTPLUS>(p, grad, s);
        case PG_ACT_SWISH: return dispatch_grad<T, PG_ACT_SWISH>(p, grad, s);
    }
    return PG_ERR_INVALID_ARG;
}


// ---------------------------------------------------------------------------------------------------------------------
// First-derivative form with the bias gradient in the same pass (training route): dx = dy * act'(.) * gain (clamp-masked)
// AND db[c] = sum over everything but the channel axis of dx -- the reference composes these as the plugin call plus
// `dx.sum(...)` (bias_act.py:176-186), i.e. a second read of dx.  Two layouts:
//   planar       (stepB = plane size, NCHW): one workgroup per (plane, chunk of SUM_CHUNK vectors) -> partial[(n*C + c)*K + k]
//   interleaved  (stepB = 1, channels-last / [N, C]): grid-stride; with C/VEC dividing the block, a lane keeps ONE channel
//                vector for its whole walk -> per-lane sums, one LDS fold per block -> partial[block*C + c]
// then a fixed-order fold of the partials (one wavefront per channel, fp64) -> deterministic, unlike an atomic sum.
// Sums are taken over the values as stored (rounded to T), what `dx.sum()` sees.  HBM-bound: sizeof(T) * (2 reads + 1 write).

constexpr int SUM_ITERS = 4;          // 16-byte vectors per lane in a planar chunk
constexpr int SUM_MAX_BLOCKS = 1024;  // interleaved grid cap (the workspace is sized from it without asking the device)

struct SumPlan { int mode; int64_t planes; int K; int64_t chunk; int grid; int64_t floats; };   // mode 0 = not covered, 1 planar, 2 interleaved

static SumPlan sum_plan(int dtype, int64_t sizeX, int sizeB, int64_t stepB) {
    SumPlan pl = {0, 0, 0, 0, 0, 0};
    const int esz = dtype == PG_F32 ? 4 : (dtype == PG_F16 || dtype == PG_BF16) ? 2 : 0;
    if (!esz || sizeX <= 0 || sizeB <= 0 || stepB <= 0) return pl;
    const int VEC = 16 / esz;
    if (stepB == 1) {
        if (sizeB % VEC || 256 % (sizeB / VEC) || sizeX % sizeB) return pl;
        const int64_t nvec = sizeX / VEC;
        int64_t blocks = (nvec + 255) / 256;
        pl.mode = 2; pl.grid = (int)(blocks > SUM_MAX_BLOCKS ? SUM_MAX_BLOCKS : blocks); pl.floats = (int64_t)pl.grid * sizeB;
        return pl;
    }
    if (stepB % VEC || sizeX % (stepB * sizeB)) return pl;
    pl.chunk = (int64_t)VEC * 256 * SUM_ITERS;
    const int64_t K = (stepB + pl.chunk - 1) / pl.chunk;
    pl.planes = sizeX / stepB;
    if (K > (1 << 20) || pl.planes * K > 0x7fffffffLL) return pl;
    pl.mode = 1; pl.K = (int)K; pl.floats = pl.planes * K;
    return pl;
}

struct SumParams {
    const void* dy; const void* yref; void* dx; float* partial;
    int64_t sizeX; int64_t stepB; int sizeB; int K; int64_t chunk;
    float alpha, gain, clamp;
};

__device__ __forceinline__ float block_sum_256(float s, float* sm) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) s += __shfl_down(s, o, 64);
    if ((threadIdx.x & 63) == 0) sm[threadIdx.x >> 6] = s;
    __syncthreads();
    return sm[0] + sm[1] + sm[2] + sm[3];       // every lane returns the same, fixed-order total
}

template <typename T, int A, bool WRITE>
__global__ __launch_bounds__(256) void bias_act_grad_sum_planar(SumParams p) {
    constexpr int VEC = 16 / sizeof(T);
    typedef vec_t<T, VEC> V;
    __shared__ float sm[4];
    const int64_t q = blockIdx.x / p.K;
    const int64_t lo = (int64_t)(blockIdx.x % p.K) * p.chunk;
    const int64_t hi = lo + p.chunk < p.stepB ? lo + p.chunk : p.stepB;
    const T* __restrict__ dyp = (const T*)p.dy + q * p.stepB;
    const T* __restrict__ yrp = p.yref ? (const T*)p.yref + q * p.stepB : nullptr;
    T* __restrict__ dxp = WRITE ? (T*)p.dx + q * p.stepB : nullptr;
    float s = 0.f;
    for (int64_t e = lo + (int64_t)threadIdx.x * VEC; e < hi; e += 256 * VEC) {
        const V dv = *(const V*)(dyp + e);
        V yv, out;
        if (yrp) yv = *(const V*)(yrp + e);
#pragma unroll
        for (int k = 0; k < VEC; k++) {
            out.v[k] = bias_act_one<T, A, 1>(dv.v[k], (T)0, (T)0, yrp ? yv.v[k] : (T)0, (T)1, p.alpha, p.gain, p.clamp);
            s += (float)out.v[k];
        }
        if (WRITE) *(V*)(dxp + e) = out;
    }
    s = block_sum_256(s, sm);
    if (threadIdx.x == 0) p.partial[blockIdx.x] = s;
}

template <typename T, int A, bool WRITE>
__global__ __launch_bounds__(256) void bias_act_grad_sum_interleaved(SumParams p) {
    constexpr int VEC = 16 / sizeof(T);
    typedef vec_t<T, VEC> V;
    __shared__ float sm[256 * VEC];
    const T* __restrict__ dyp = (const T*)p.dy;
    const T* __restrict__ yrp = (const T*)p.yref;
    T* __restrict__ dxp = (T*)p.dx;
    const int64_t nvec = p.sizeX / VEC;
    const int64_t stride = (int64_t)gridDim.x * 256;       // a multiple of C / VEC: the lane's channel vector never changes
    float s[VEC];
#pragma unroll
    for (int k = 0; k < VEC; k++) s[k] = 0.f;
    for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < nvec; i += stride) {
        const V dv = ((const V*)dyp)[i];
        V yv, out;
        if (yrp) yv = ((const V*)yrp)[i];
#pragma unroll
        for (int k = 0; k < VEC; k++) {
            out.v[k] = bias_act_one<T, A, 1>(dv.v[k], (T)0, (T)0, yrp ? yv.v[k] : (T)0, (T)1, p.alpha, p.gain, p.clamp);
            s[k] += (float)out.v[k];
        }
        if (WRITE) ((V*)dxp)[i] = out;
    }
#pragma unroll
    for (int k = 0; k < VEC; k++) sm[threadIdx.x * VEC + k] = s[k];
    __syncthreads();
    const int cv = p.sizeB / VEC;
    if ((int)threadIdx.x < cv) {
#pragma unroll
        for (int k = 0; k < VEC; k++) {
            float a = 0.f;
            for (int r = 0; r < 256 / cv; r++) a += sm[(r * cv + threadIdx.x) * VEC + k];
            p.partial[(int64_t)blockIdx.x * p.sizeB + threadIdx.x * VEC + k] = a;
        }
    }
}

// out[c] = sum_{j < J} sum_{i < I} partial[j*sj + c*sc + i]; one wavefront per channel, lanes take (j, i) pairs round-robin, fixed-order fold.
template <typename T>
__global__ __launch_bounds__(64) void channel_sum_finish(const float* __restrict__ partial, T* __restrict__ out, int64_t J, int64_t sj, int64_t sc, int I) {
    const int c = blockIdx.x;
    double a = 0.0;
    const int64_t total = J * I;
    for (int64_t t = threadIdx.x; t < total; t += 64) a += (double)partial[(t / I) * sj + (int64_t)c * sc + (t % I)];
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) a += __shfl_down(a, o, 64);
    if (threadIdx.x == 0) out[c] = (T)a;
}

template <typename T, int A>
int launch_grad_sum(const SumPlan& pl, SumParams p, void* db, hipStream_t stream) {
    const bool write = p.dx != nullptr;
    if (pl.mode == 1) {
        p.K = pl.K; p.chunk = pl.chunk;
        const dim3 grid((unsigned)(pl.planes * pl.K));
        if (write) hipLaunchKernelGGL((bias_act_grad_sum_planar<T, A, true>), grid, dim3(256), 0, stream, p);
        else hipLaunchKernelGGL((bias_act_grad_sum_planar<T, A, false>), grid, dim3(256), 0, stream, p);
        const int64_t n = pl.planes / p.sizeB;
        hipLaunchKernelGGL((channel_sum_finish<T>), dim3(p.sizeB), dim3(64), 0, stream, (const float*)p.partial, (T*)db, n, (int64_t)p.sizeB * pl.K, (int64_t)pl.K, pl.K);
    } else {
        if (write) hipLaunchKernelGGL((bias_act_grad_sum_interleaved<T, A, true>), dim3(pl.grid), dim3(256), 0, stream, p);
        else hipLaunchKernelGGL((bias_act_grad_sum_interleaved<T, A, false>), dim3(pl.grid), dim3(256), 0, stream, p);
        hipLaunchKernelGGL((channel_sum_finish<T>), dim3(p.sizeB), dim3(64), 0, stream, (const float*)p.partial, (T*)db, (int64_t)pl.grid, (int64_t)p.sizeB, (int64_t)1, 1);
    }
    return launch_status();
}

template <typename T>
int dispatch_grad_sum(const SumPlan& pl, const SumParams& p, void* db, int act, hipStream_t s) {
    switch (act) {      // the activations of the networks' layers; the others keep the two-pass composition
        case PG_ACT_LINEAR: return launch_grad_sum<T, PG_ACT_LINEAR>(pl, p, db, s);
        case PG_ACT_RELU: return launch_grad_sum<T, PG_ACT_RELU>(pl, p, db, s);
        case PG_ACT_LRELU: return launch_grad_sum<T, PG_ACT_LRELU>(pl, p, db, s);
    }
    return PG_ERR_UNSUPPORTED;
}

}  // namespace

PG_EXPORT int pg_bias_act_abi_version(void) { return PG_ABI_VERSION; }

PG_EXPORT int pg_bias_act(const void* x, const void* b, const void* xref, const void* yref, const void* dy, void* y,
                          int dtype, int64_t sizeX, int sizeB, int64_t stepB,
                          int grad, int act, float alpha, float gain, float clamp, void* stream) {
    if (sizeX == 0) return PG_OK;
    if (!x || !y || sizeX < 0 || grad < 0 || grad > 2) return PG_ERR_INVALID_ARG;
    if (b && (sizeB <= 0 || stepB <= 0)) return PG_ERR_INVALID_ARG;
    if (act < PG_ACT_LINEAR || act > PG_ACT_SWISH) return PG_ERR_INVALID_ARG;
    Params p;
    p.x = x; p.b = b; p.xref = xref; p.yref = yref; p.dy = dy; p.y = y;
    p.sizeX = sizeX; p.sizeB = b ? sizeB : 1; p.stepB = b ? stepB : 1;
    p.alpha = alpha; p.gain = gain; p.clamp = clamp;
    hipStream_t s = (hipStream_t)stream;
    switch (dtype) {
        case PG_F32: return dispatch_act<float>(p, act, grad, s);
        case PG_F16: return dispatch_act<pg::f16_t>(p, act, grad, s);
        case PG_BF16: return dispatch_act<pg::bf16_t>(p, act, grad, s);
        case PG_F64: return dispatch_act<double>(p, act, grad, s);
    }
    return PG_ERR_INVALID_ARG;
}

PG_EXPORT int64_t pg_bias_act_grad_bias_workspace(int dtype, int64_t sizeX, int sizeB, int64_t stepB) {
    return sum_plan(dtype, sizeX, sizeB, stepB).floats * (int64_t)sizeof(float);
}

PG_EXPORT int pg_bias_act_grad_bias(const void* dy, const void* yref, void* dx, void* db, void* workspace, int64_t workspace_bytes,
                                    int dtype, int64_t sizeX, int sizeB, int64_t stepB,
                                    int act, float alpha, float gain, float clamp, void* stream) {
    if (!dy || !db || !workspace || sizeX <= 0 || sizeB <= 0 || stepB <= 0) return PG_ERR_INVALID_ARG;
    const SumPlan pl = sum_plan(dtype, sizeX, sizeB, stepB);
    if (!pl.mode) return PG_ERR_UNSUPPORTED;
    if (workspace_bytes < pl.floats * (int64_t)sizeof(float)) return PG_ERR_INVALID_ARG;
    if (!aligned16(dy) || (yref && !aligned16(yref)) || (dx && !aligned16(dx))) return PG_ERR_UNSUPPORTED;
    if ((act == PG_ACT_RELU || act == PG_ACT_LRELU || clamp >= 0.f) && !yref) return PG_ERR_INVALID_ARG;   // their derivative reads the forward output
    SumParams p;
    p.dy = dy; p.yref = yref; p.dx = dx; p.partial = (float*)workspace;
    p.sizeX = sizeX; p.stepB = stepB; p.sizeB = sizeB; p.K = 0; p.chunk = 0;
    p.alpha = alpha; p.gain = gain; p.clamp = clamp;
    hipStream_t s = (hipStream_t)stream;
    switch (dtype) {
        case PG_F32: return dispatch_grad_sum<float>(pl, p, db, act, s);
        case PG_F16: return dispatch_grad_sum<pg::f16_t>(pl, p, db, act, s);
        case PG_BF16: return dispatch_grad_sum<pg::bf16_t>(pl, p, db, act, s);
    }
    return PG_ERR_UNSUPPORTED;
}
