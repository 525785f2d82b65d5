// Streaming 1x1 convolution instantiations (own translation unit: see conv2d_kernel.h on build time).
#include <cstdlib>
#include "conv2d_s1x1.h"

namespace pgconv {
int launch_s1x1(const ConvParams& p, hipStream_t s) {
    static const bool on = [] { const char* e = getenv("PG_S1X1"); return e ? atoi(e) != 0 : true; }();      // A/B switch
    if (!on || !s1x1_ok(p)) return PG_ERR_UNSUPPORTED;
    static const bool ring = [] { const char* e = getenv("PG_S1X1_RING"); return e ? atoi(e) != 0 : true; }();    // A/B switch
    if (ring && s1x1_ring_ok(p)) return launch_s1x1_ring(p, s);
    return launch_s1x1_t<2, 4>(p, s);
}
}  // namespace pgconv
