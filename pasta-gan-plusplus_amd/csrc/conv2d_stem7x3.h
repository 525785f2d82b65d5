// The 7x7, three-channel stem convolution of the garment encoder (networks.py:2233-2238 `spade_encoder[0]`: Conv2dLayer(3, 64, kernel_size=7), stride 1, pad 3)
// with its multiplies on the bf16 matrix pipe (round 6): every float32 operand the exact sum of three bf16 values (truncation split), a float32 product = the six
// largest plane products on v_mfma_f32_32x32x16_bf16 with float32 accumulation -- float32-class results (conv2d_up2x3.h, conv2d_wino4.h X3 form).
// The layer is 39.5 GFLOP at N = 8, 512^2: 251 us at the fp32 matrix peak, 452 us measured on conv2d_mfma<7,7,1,64,2,0>; its output is 537 MB (~110 us of HBM).
//
//   * K order = (kx, ci, ky): 21 groups of 8 (seven rows ky + one zero weight), 168 -> 11 MFMA K steps (the last half step all zero).  A lane's B operand of a
//     group is then 8 VERTICALLY consecutive samples of one input column and channel -- a "column record": 16 bytes, the same for every pixel of an output row
//     that uses the column, 16-byte aligned whatever the pixel's x (a horizontal window would start at any 2-byte offset).
//   * workgroup = 8 waves = a vertical strip of 256 output columns x 32 rows x 64 couts, walked row by row.  LDS holds the records of the current row
//     [plane 3][channel 3][262 columns] x 16 B, double buffered.  While the row is multiplied, the threads make the next row's records (786 (channel, column) tasks, at most two per thread): read the
//     task's three plane records, shift them down one sample (v_alignbit), insert the split of the sample five rows below (requested a row ahead) -- one barrier per row.
//     Every input sample is split once per strip and row: ~60 VALU per lane and row beside 132 MFMAs per wave and row.
//   * the weights (64 couts x 176 x three planes) never leave registers: wave = (m-tile of 32 couts, two of the eight 32-pixel blocks of the row), its 33 A
//     fragments (132 registers) are loaded once per launch.  Two accumulators per wave; two waves per SIMD.
//   * epilogue = pg_conv2d_forward's for the stages this layer uses: bias, linear | relu | lrelu, gain, clamp.  Anything else is declined.
#pragma once
#include "conv2d_kernel.h"

#ifndef S7_EXP
#define S7_EXP 0          // dev ablations (results wrong by design; tools/stem7_variants.py): 1 no record updates, 2 no output stores, 4 no MFMAs, 8 no B-operand reads
#endif

namespace pgconv {

constexpr int S7_TW = 256, S7_NCOL = S7_TW + 6, S7_ROWS = 32, S7_KS = 11, S7_GROUPS = 21;      // S7_ROWS: rows of a segment at most (the launcher halves it while the grid does not fill the chip)
constexpr int S7_REC_B = 9 * S7_NCOL * 16;                   // bytes of one buffer of records

struct Stem7Params {
    const float* x; float* y; const unsigned char* wx3; const float* bias;
    int N, H, W, Cout, mpairs, strips, segs, seg_rows;
    int64_t ys[4];
    int act; float alpha, gain, clamp;
};

typedef __bf16 s7_bf16x8 __attribute__((ext_vector_type(8)));
typedef unsigned s7_u32x4 __attribute__((ext_vector_type(4)));

__device__ __forceinline__ void s7_split(float v, unsigned& b0, unsigned& b1, unsigned& b2) {      // three bf16 terms in the low halves
    const unsigned h0 = __float_as_uint(v) & 0xffff0000u;
    const float r = v - __uint_as_float(h0);
    const unsigned h1 = __float_as_uint(r) & 0xffff0000u;
    const float r2 = r - __uint_as_float(h1);
    b0 = h0 >> 16; b1 = h1 >> 16; b2 = __float_as_uint(r2) >> 16;
}

__global__ __launch_bounds__(512, 1) void conv2d_stem7x3(Stem7Params p) {
    extern __shared__ __attribute__((aligned(16))) unsigned char s7_smem[];
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int half = lane >> 5, l31 = lane & 31;
    const int mt = wave & 1, bq = wave >> 1;

    int L = blockIdx.x;
    const int seg = L % p.segs; L /= p.segs;
    const int strip = L % p.strips; L /= p.strips;
    const int mp = L % p.mpairs;
    const int n = L / p.mpairs;
    const int x0 = strip * S7_TW, y0 = seg * p.seg_rows;
    const int y1 = y0 + p.seg_rows < p.H ? y0 + p.seg_rows : p.H;
    const int HW = p.H * p.W;
    const float* xin = p.x + (int64_t)n * 3 * HW;

    // ---- this wave's weights: [m-tile][K step][plane][lane] x 16 B
    s7_u32x4 A[S7_KS][3];
    {
        const unsigned char* wb = p.wx3 + ((int64_t)(mp * 2 + mt) * S7_KS * 3 * 64 + lane) * 16;
#pragma unroll
        for (int ks = 0; ks < S7_KS; ks++)
#pragma unroll
            for (int pl = 0; pl < 3; pl++) A[ks][pl] = *(const s7_u32x4*)(wb + (size_t)(ks * 3 + pl) * 64 * 16);
    }

    // ---- records of row y0: rows y0 - 3 .. y0 + 4.  A record task = (channel, column): 3 x 262 = 786 of them, at most two per thread (task = tid, tid + 512;
    // consecutive threads take consecutive columns: coalesced sample loads)
    int rec_off[2];                                           // byte offset of the task's plane-0 record in a buffer, < 0: no task
    unsigned voff[2];                                         // byte offset of the task's column in row 0 of its channel (0x80000000: outside the image -> zero)
#pragma unroll
    for (int q = 0; q < 2; q++) {
        const int task = tid + 512 * q;
        const int ci = task / S7_NCOL, col = task - ci * S7_NCOL;
        const int cg = x0 - 3 + col;
        const bool in = task < 3 * S7_NCOL && cg >= 0 && cg < p.W;
        rec_off[q] = task < 3 * S7_NCOL ? (ci * S7_NCOL + col) * 16 : -1;
        voff[q] = in ? (unsigned)(ci * HW + cg) * 4u : 0x80000000u;
    }
#pragma unroll
    for (int q = 0; q < 2; q++) {
        if (rec_off[q] >= 0) {
            s7_u32x4 r[3] = {{0u, 0u, 0u, 0u}, {0u, 0u, 0u, 0u}, {0u, 0u, 0u, 0u}};
#pragma unroll
            for (int j = 0; j < 8; j++) {
                const int row = y0 - 3 + j;
                const float v = (voff[q] != 0x80000000u && row >= 0 && row < p.H) ? xin[(voff[q] >> 2) + row * p.W] : 0.f;
                unsigned b0, b1, b2;
                s7_split(v, b0, b1, b2);
                r[0][j >> 1] |= b0 << (16 * (j & 1));
                r[1][j >> 1] |= b1 << (16 * (j & 1));
                r[2][j >> 1] |= b2 << (16 * (j & 1));
            }
#pragma unroll
            for (int pl = 0; pl < 3; pl++) *(s7_u32x4*)(s7_smem + (size_t)(pl * 3 * S7_NCOL * 16 + rec_off[q])) = r[pl];
        }
    }
    float* smp_s = (float*)(s7_smem + 2 * S7_REC_B);          // [2 tasks][512 threads] samples of the row being inserted (LDS-DMA)
    float* bias_s = smp_s + 2 * 512;                          // [64]
    if (tid < 64) {
        const int co = mp * 64 + tid;
        bias_s[tid] = (p.bias && co < p.Cout) ? p.bias[co] : 0.f;
    }
    __syncthreads();
    // the weights are home before the row loop: a load still pending at the loop's first use would put `s_waitcnt vmcnt(n)` into the loop body, where it then
    // waits on the previous row's output stores every time round (found in the first build's assembly: -120 us with the stores taken out)
#pragma unroll
    for (int ks = 0; ks < S7_KS; ks++)
#pragma unroll
        for (int pl = 0; pl < 3; pl++) asm volatile("" : "+v"(A[ks][pl]));

    i32x4 xrsrc;                                              // the input image through a buffer descriptor (range check = zero fill)
    {
        const uint64_t base = (uint64_t)(uintptr_t)xin;
        xrsrc[0] = __builtin_amdgcn_readfirstlane((int)(unsigned)base);
        xrsrc[1] = __builtin_amdgcn_readfirstlane((int)(unsigned)(base >> 32) & 0xffff);
        xrsrc[2] = 3 * HW * 4;
        xrsrc[3] = 0x00020000;
    }
    const unsigned smp_b = __builtin_amdgcn_readfirstlane(lds_offset(smp_s)) + (unsigned)wave * 256u;
    // output stores through a buffer descriptor: one 32-bit offset register per store instead of a 64-bit address pair
    i32x4 yrsrc;
    {
        const uint64_t base = (uint64_t)(uintptr_t)(p.y + (int64_t)n * p.ys[0]);
        yrsrc[0] = __builtin_amdgcn_readfirstlane((int)(unsigned)base);
        yrsrc[1] = __builtin_amdgcn_readfirstlane((int)(unsigned)(base >> 32) & 0xffff);
        yrsrc[2] = 0x7ffffffc;
        yrsrc[3] = 0x00020000;
    }

    const float gain = p.gain;
    const float cl = p.clamp >= 0.f ? p.clamp : __builtin_inff();
    const float slope = act_slope(p.act, p.alpha);

    for (int yy = y0; yy < y1; yy++) {
        const int buf = (yy - y0) & 1;
        const unsigned char* cur = s7_smem + (size_t)buf * S7_REC_B;
        unsigned char* nb = s7_smem + (size_t)(buf ^ 1) * S7_REC_B;
        // ---- the next row's records (rows yy - 2 .. yy + 5): the samples of row yy + 5 go straight to LDS (LDS-DMA from inline asm, waited for by hand after
        // K step 7).  A load the compiler can see would make it wait -- `s_waitcnt vmcnt(0)` at the top of the loop -- for the previous row's output stores as well.
        const bool upd = yy + 1 < y1 && !(S7_EXP & 1);
        if (upd) {
            const bool row_ok = yy + 5 < p.H;                 // (yy + 5 >= 0 always)
            const int soff = row_ok ? (yy + 5) * p.W * 4 : 0;
#pragma unroll
            for (int q = 0; q < 2; q++) dma_dword(xrsrc, smp_b + 2048u * q, row_ok ? voff[q] : 0x80000000u, soff);
        }
        auto update_records = [&]() __attribute__((always_inline)) {
            if (!upd) return;
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");  // the two sample requests (and, long since, the previous row's stores)
#pragma unroll
            for (int q = 0; q < 2; q++) {
                if (rec_off[q] >= 0) {
                    unsigned b[3];
                    s7_split(smp_s[512 * q + tid], b[0], b[1], b[2]);
#pragma unroll
                    for (int pl = 0; pl < 3; pl++) {
                        const size_t ro = (size_t)(pl * 3 * S7_NCOL * 16 + rec_off[q]);
                        const s7_u32x4 r = *(const s7_u32x4*)(cur + ro);
                        s7_u32x4 o;
                        o[0] = __builtin_amdgcn_alignbit(r[1], r[0], 16);
                        o[1] = __builtin_amdgcn_alignbit(r[2], r[1], 16);
                        o[2] = __builtin_amdgcn_alignbit(r[3], r[2], 16);
                        o[3] = __builtin_amdgcn_alignbit(b[pl], r[3], 16);
                        *(s7_u32x4*)(nb + ro) = o;
                    }
                }
            }
        };
        // ---- 11 K steps x 6 plane products x 2 pixel blocks
        f32x16 acc[2];
#pragma unroll
        for (int bi = 0; bi < 2; bi++)
#pragma unroll
            for (int k = 0; k < 16; k++) acc[bi][k] = 0.f;
        // B operands: plane 0 of a K step is requested one step ahead and its three products go first, so planes 1 and 2 -- requested at the top of the step --
        // arrive behind 6 MFMAs (two waves per SIMD and 132 registers of weights leave no room for a whole step of look-ahead: 134 us of exposed LDS latency)
        auto b_addr = [&](int ks) __attribute__((always_inline)) {
            const int g0 = 2 * ks, g1 = 2 * ks + 1 < S7_GROUPS ? 2 * ks + 1 : S7_GROUPS - 1;      // group = 3 kx + ci (the 22nd half step has zero weights: any record)
            const int off0 = ((g0 % 3) * S7_NCOL + g0 / 3) * 16, off1 = ((g1 % 3) * S7_NCOL + g1 / 3) * 16;
            return cur + (half ? off1 : off0) + (size_t)(32 * bq + l31) * 16;
        };
        auto b_load = [&](const unsigned char* bb, int pl, int bi) __attribute__((always_inline)) {
            return *(const s7_u32x4*)(bb + (size_t)(pl * 3 * S7_NCOL + 128 * bi) * 16);
        };
        s7_u32x4 B0[2], B0n[2], B1[2], B2[2];
        {
            const unsigned char* bb = b_addr(0);
            B0[0] = b_load(bb, 0, 0); B0[1] = b_load(bb, 0, 1);
        }
#pragma unroll
        for (int ks = 0; ks < S7_KS; ks++) {
            const unsigned char* bb = b_addr(ks);
#pragma unroll
            for (int bi = 0; bi < 2; bi++) { B1[bi] = b_load(bb, 1, bi); B2[bi] = b_load(bb, 2, bi); }
            if (ks + 1 < S7_KS) {
                const unsigned char* bn = b_addr(ks + 1);
                B0n[0] = b_load(bn, 0, 0); B0n[1] = b_load(bn, 0, 1);
            }
            __builtin_amdgcn_sched_barrier(0);
            constexpr int PA6[6] = {2, 1, 0, 1, 0, 0}, PB6[6] = {0, 0, 0, 1, 1, 2};
#pragma unroll
            for (int pr = 0; pr < 6; pr++)
#pragma unroll
                for (int bi = 0; bi < 2; bi++) {
                    const s7_u32x4 bv = (S7_EXP & 8) ? A[ks][PB6[pr]] : (PB6[pr] == 0 ? B0[bi] : (PB6[pr] == 1 ? B1[bi] : B2[bi]));
                    if (!(S7_EXP & 4) || bv[0] == 0x12345678u)
                        acc[bi] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(s7_bf16x8, A[ks][PA6[pr]]), __builtin_bit_cast(s7_bf16x8, bv), acc[bi], 0, 0, 0);
                }
            __builtin_amdgcn_sched_barrier(0);
            B0[0] = B0n[0]; B0[1] = B0n[1];
            if (ks == 7) { update_records(); __builtin_amdgcn_sched_barrier(0); }
        }
        // ---- epilogue: D column = lane & 31 = pixel, row = (k & 3) + 8 (k >> 2) + 4 half = cout of the m-tile
        const int row_soff = yy * (int)p.ys[2] * 4;
#pragma unroll
        for (int bi = 0; bi < 2; bi++) {
            const int px = x0 + 32 * (bq + 4 * bi) + l31;
            const int co0 = mp * 64 + mt * 32 + 4 * half;
            const unsigned o0 = (unsigned)(px * (int)p.ys[3] + co0 * (int)p.ys[1]) * 4u;
#pragma unroll
            for (int k = 0; k < 16; k++) {
                const int rowc = (k & 3) + 8 * (k >> 2);
                float v = acc[bi][k] + bias_s[mt * 32 + 4 * half + rowc];
                v = v > 0.f ? v : v * slope;
                v = fminf(fmaxf(v * gain, -cl), cl);
                const unsigned off = (px < p.W && co0 + rowc < p.Cout && (!(S7_EXP & 2) || v == 12345.678f)) ? o0 + (unsigned)(rowc * (int)p.ys[1]) * 4u : 0x80000000u;      // out of range: dropped
                asm volatile("buffer_store_dword %0, %1, %2, %3 offen" :: "v"(v), "v"(off), "s"(yrsrc), "s"(row_soff));
            }
        }
        __syncthreads();                                      // the next row's records are complete; this row's have been read
    }
}

// w: OIHW float32 [Cout][3][7][7] (contiguous) -> [m-tile][K step 11][plane 3][lane 64][8] bf16; value = w * scale rounded to float32 once, then split
__global__ __launch_bounds__(256) void stem7x3_pack_kernel(const float* __restrict__ w, unsigned short* __restrict__ out, int Cout, int mtiles, float scale, int flip) {
    const int total = mtiles * S7_KS * 64 * 8;
    for (int i = blockIdx.x * 256 + threadIdx.x; i < total; i += gridDim.x * 256) {
        const int j = i & 7, lane = (i >> 3) & 63, ks = (i >> 9) % S7_KS, mtile = i / (512 * S7_KS);
        const int g = 2 * ks + (lane >> 5), co = mtile * 32 + (lane & 31);
        float v = 0.f;
        if (g < S7_GROUPS && j < 7 && co < Cout) {
            const int kx = g / 3, ci = g % 3, ky = j;
            v = w[((co * 3 + ci) * 7 + (flip ? 6 - ky : ky)) * 7 + (flip ? 6 - kx : kx)] * scale;
        }
        const unsigned u0 = __float_as_uint(v) & 0xffff0000u;
        const float r = v - __uint_as_float(u0);
        const unsigned u1 = __float_as_uint(r) & 0xffff0000u;
        const unsigned u2 = __float_as_uint(r - __uint_as_float(u1)) & 0xffff0000u;
        const int64_t dst = ((((int64_t)mtile * S7_KS + ks) * 3) * 64 + lane) * 8 + j;      // plane 0; planes are 64 * 8 words apart
        out[dst] = (unsigned short)(u0 >> 16);
        out[dst + 512] = (unsigned short)(u1 >> 16);
        out[dst + 1024] = (unsigned short)(u2 >> 16);
    }
}

inline int64_t stem7x3_packed_bytes(int Cout) { return (int64_t)((Cout + 63) / 64 * 2) * S7_KS * 3 * 64 * 16; }

inline int launch_stem7x3(Stem7Params p, hipStream_t s) {
    p.mpairs = (p.Cout + 63) / 64;
    p.strips = (p.W + S7_TW - 1) / S7_TW;
    p.seg_rows = S7_ROWS;
    while (p.seg_rows > 8 && (int64_t)p.N * p.mpairs * p.strips * ((p.H + p.seg_rows - 1) / p.seg_rows) < (int64_t)num_cu()) p.seg_rows /= 2;      // (a segment's first row costs 8 rows of samples)
    p.segs = (p.H + p.seg_rows - 1) / p.seg_rows;
    const int64_t blocks = (int64_t)p.N * p.mpairs * p.strips * p.segs;
    if (blocks > 0x7fffffffLL || (int64_t)3 * p.H * p.W >= 0x7fffffffLL) return PG_ERR_TOO_LARGE;
    if (((int64_t)(p.mpairs * 64) * p.ys[1] + (int64_t)p.H * p.ys[2] + (int64_t)p.W * p.ys[3]) * 4 >= 0x7fffffffLL) return PG_ERR_UNSUPPORTED;      // 32-bit byte offsets inside one image of y
    const size_t lds = (size_t)2 * S7_REC_B + (2 * 512 + 64) * sizeof(float);
    static PerDeviceOnce once;
    const hipError_t e = once.run([] { return hipFuncSetAttribute((const void*)conv2d_stem7x3, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024); });
    if (e != hipSuccess) return (int)e;
    hipLaunchKernelGGL(conv2d_stem7x3, dim3((unsigned)blocks), dim3(512), lds, s, p);
    return launch_status();
}

}  // namespace pgconv
