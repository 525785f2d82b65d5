// Forward activation table shared by the conv prologue/epilogue (runtime-selected, wave-uniform).
// Same definitions as bias_act.hip's templated forms (reference: bias_act.py:23-33).
#pragma once
#include "pg_common.h"

namespace pg {

__device__ __forceinline__ float act_forward(int act, float x, float alpha) {
    switch (act) {
        case PG_ACT_RELU: return x > 0.f ? x : 0.f;
        case PG_ACT_LRELU: return x > 0.f ? x : x * alpha;
        case PG_ACT_TANH: return tanhf(x);
        case PG_ACT_SIGMOID: return x >= 0.f ? 1.f / (1.f + expf(-x)) : expf(x) / (1.f + expf(x));
        case PG_ACT_ELU: return x >= 0.f ? x : expm1f(x);
        case PG_ACT_SELU: return x >= 0.f ? 1.0507009873554804934193349852946f * x
                                         : (1.0507009873554804934193349852946f * 1.6732632423543772848170429916717f) * expm1f(x);
        case PG_ACT_SOFTPLUS: return x > 20.f ? x : log1pf(expf(x));
        case PG_ACT_SWISH: return x >= 0.f ? x / (1.f + expf(-x)) : x * expf(x) / (1.f + expf(x));
        default: return x;
    }
}

__device__ __forceinline__ float clampf(float v, float c) { return v > c ? c : (v < -c ? -c : v); }

}  // namespace pg
