// conv2d_mfma16<T, KH, KW, STRIDE, ...> instantiations for geometry k3s1 (see conv2d_kernel16.h): <KH, KW, stride, N tiles per wave,
// channels per chunk[, staging buffers]> x {bf16, fp16} x {32, 64}-cout blocks.
#include "conv2d_kernel16.h"
namespace pgconv16 {
int launch16_k3s1_kc32(const Conv16Params& p, int dtype, hipStream_t s);      // conv2d16_inst_k3s1b.hip
int launch16_k3s1(const Conv16Params& p, int dtype, hipStream_t s) {
    // 32-channel K chunks (two-role form only, two staging buffers): half the chunk barriers for the layers with >= 64 input channels
    static const bool kc32 = [] { const char* e = getenv("PG_CONV16_KC32"); const char* sp = getenv("PG_CONV16_SPLIT"); return (e ? atoi(e) != 0 : true) && (sp ? atoi(sp) != 0 : true); }();
    const int cin_loop = p.ksplit > 1 ? p.kpart : p.Cin;
    static const int kc32_min = [] { const char* e = getenv("PG_CONV16_KC32_MIN"); return e ? atoi(e) : 64; }();      // dev A/B: 32 = also the 32-channel layer (one chunk per tile)
    if (kc32 && cin_loop % 32 == 0 && cin_loop >= kc32_min && small_tile16(3, 3, 1, p.OH, p.OW, p.Cout, p.f.phase_cout) != 1)
        return launch16_k3s1_kc32(p, dtype, s);
    return launch16_dt<3, 3, 1, 2, 16>(p, dtype, s);
}
}
