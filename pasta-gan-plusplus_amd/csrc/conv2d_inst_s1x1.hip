// Streaming 1x1 convolution instantiations (own translation unit: see conv2d_kernel.h on build time).
#include <cstdlib>
#include "conv2d_s1x1.h"

namespace pgconv {
int launch_s1x1(const ConvParams& p, hipStream_t s) {
    static const bool on = [] { const char* e = getenv("PG_S1X1"); return e ? atoi(e) != 0 : true; }();      // A/B switch
    if (!on) return PG_ERR_UNSUPPORTED;
    static const int ring = [] { const char* e = getenv("PG_S1X1_RING"); return e ? atoi(e) : 2; }();      // A/B switch: 0 off, 1 Cout = 64 only, 2 every multiple of 64
    if (ring && s1x1_ring_ok(p) && (ring > 1 || p.Cout == 64)) return launch_s1x1_ring(p, s);
    if (!s1x1_ok(p)) return PG_ERR_UNSUPPORTED;
    return launch_s1x1_t<2, 4>(p, s);
}
}  // namespace pgconv
