// The optimizer side of the training step as ONE launch per phase (round 5): nan_to_num + Adam over the phase's flat parameter / gradient / moment buffers.
//
// What it replaces in the reference: training_loop_fullbody.py:632-639 -- per parameter `torch.nan_to_num(param.grad, nan=0, posinf=1e5, neginf=-1e5,
// out=param.grad)`, then `opt.step()` (torch.optim.Adam, betas (0, 0.99), eps 1e-8, no weight decay) -- which PyTorch-ROCm runs as ~10 multi-tensor
// passes over ~200 tensors per phase plus one nan_to_num pass: ~120 launches and ~11 reads / writes of every parameter per phase.  Here the phase's
// parameters, gradients (training/ddp.py GradBucket: the .grad tensors are views of one flat fp32 bucket) and both Adam moments share ONE flat layout, a
// chunk table maps workgroups to (parameter, range), and one pass reads p, g, m, v and writes g (cleaned), m, v, p: 28 bytes per parameter, HBM-bound.
// Parameters no rank produced a gradient for (a per-parameter `alive` flag on the DEVICE: no read-back, no host decision) are skipped entirely -- their
// moments and step count stay untouched, which is what Adam does with `grad is None`.  The step count is per parameter, like torch's state['step'].
#include "pg_common.h"
#include <math.h>

namespace {

struct AdamParams {
    float* p; float* g; float* m; float* v;
    const int* chunks;          // [nchunks][4]: element offset, length, parameter index, 1 if the parameter's first chunk
    const float* alive;         // [nparams] > 0: some rank produced a gradient
    const float* steps_in;      // [nparams] steps taken so far
    float* steps_out;           // [nparams] after this call
    float lr, beta1, beta2, eps, nan_v, posinf_v, neginf_v;
};

constexpr int ADAM_THREADS = 256, ADAM_CHUNK = 2048;

__device__ __forceinline__ float clean(float g, float nan_v, float posinf_v, float neginf_v) {
    if (g != g) return nan_v;
    if (g == __builtin_inff()) return posinf_v;
    if (g == -__builtin_inff()) return neginf_v;
    return g;
}

// p, m, v exactly as torch.optim.Adam's single-tensor formulas in float32 (exp_avg.lerp_(g, 1 - b1); exp_avg_sq.mul_(b2).addcmul_(g, g, value = 1 - b2);
// denom = exp_avg_sq.sqrt() / sqrt(bias_correction2) + eps; p.addcdiv_(exp_avg, denom, value = -lr / bias_correction1)), bias corrections in float64
__global__ __launch_bounds__(ADAM_THREADS) void adam_flat_kernel(AdamParams a) {
    const int* c = a.chunks + 4 * (int64_t)blockIdx.x;
    const int off = c[0], len = c[1], pi = c[2], first = c[3];
    const bool alive = a.alive[pi] > 0.f;
    const float t = a.steps_in[pi] + (alive ? 1.f : 0.f);
    if (first && threadIdx.x == 0) a.steps_out[pi] = t;
    if (!alive) return;
    const double bc1 = 1.0 - pow((double)a.beta1, (double)t), bc2 = 1.0 - pow((double)a.beta2, (double)t);
    const float step_size = (float)((double)a.lr / bc1), bc2_sqrt = (float)sqrt(bc2);
    const float w = 1.f - a.beta1, om2 = 1.f - a.beta2;
    auto one = [&](float& p, float& g, float& m, float& v) {
        g = clean(g, a.nan_v, a.posinf_v, a.neginf_v);
        m = w < 0.5f ? m + w * (g - m) : g - (g - m) * (1.f - w);          // at::lerp
        v = v * a.beta2 + om2 * g * g;
        const float denom = sqrtf(v) / bc2_sqrt + a.eps;
        p = p - step_size * (m / denom);
    };
    typedef float f4 __attribute__((ext_vector_type(4)));
    float* P = a.p + off; float* G = a.g + off; float* M = a.m + off; float* V = a.v + off;
    const int n4 = len >> 2;                                               // (parameter starts are 16-byte aligned in the flat layout)
    for (int i = threadIdx.x; i < n4; i += ADAM_THREADS) {
        f4 p = ((f4*)P)[i], g = ((f4*)G)[i], m = ((f4*)M)[i], v = ((f4*)V)[i];
#pragma unroll
        for (int e = 0; e < 4; e++) { float pe = p[e], ge = g[e], me = m[e], ve = v[e]; one(pe, ge, me, ve); p[e] = pe; g[e] = ge; m[e] = me; v[e] = ve; }
        ((f4*)P)[i] = p; ((f4*)G)[i] = g; ((f4*)M)[i] = m; ((f4*)V)[i] = v;
    }
    for (int i = 4 * n4 + threadIdx.x; i < len; i += ADAM_THREADS) one(P[i], G[i], M[i], V[i]);
}

}  // namespace

PG_EXPORT int pg_adam_flat_chunk(void) { return ADAM_CHUNK; }

PG_EXPORT int pg_adam_flat_step(float* p, float* g, float* m, float* v, const int* chunks, int nchunks, const float* alive, const float* steps_in, float* steps_out,
                                float lr, float beta1, float beta2, float eps, float nan_value, float posinf_value, float neginf_value, void* stream) {
    if (!p || !g || !m || !v || !chunks || !alive || !steps_in || !steps_out || nchunks <= 0) return PG_ERR_INVALID_ARG;
    if ((((uintptr_t)p) | ((uintptr_t)g) | ((uintptr_t)m) | ((uintptr_t)v)) & 15) return PG_ERR_UNSUPPORTED;
    if (!(lr >= 0.f) || !(beta1 >= 0.f && beta1 < 1.f) || !(beta2 >= 0.f && beta2 < 1.f) || !(eps >= 0.f)) return PG_ERR_INVALID_ARG;
    AdamParams a{p, g, m, v, chunks, alive, steps_in, steps_out, lr, beta1, beta2, eps, nan_value, posinf_value, neginf_value};
    hipLaunchKernelGGL(adam_flat_kernel, dim3((unsigned)nchunks), dim3(ADAM_THREADS), 0, (hipStream_t)stream, a);
    return pg::launch_status();
}
