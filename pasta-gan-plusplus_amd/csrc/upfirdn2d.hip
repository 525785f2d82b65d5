// upfirdn2d for gfx950: pad -> zero-stuff -> 2-D FIR -> decimate in one pass.
// Drop-in for the reference's upfirdn2d plugin (torch_utils/ops/upfirdn2d.cpp:16-94,
// upfirdn2d.cu:29-200).
//
// Two kernels:
//   * upfirdn2d_tiled  -- NCHW-dense tensors, small filters.  A 256-thread workgroup owns a
//     64 x 32 output tile of one (n, c) plane: the input footprint is staged once into LDS,
//     one image row per wave-iteration (coalesced, zero filled outside the image); each lane
//     produces a 1 x 8 column strip.  Without up-sampling every LDS offset is a compile-time
//     constant and the taps sit in registers, so an input row is read once for all the outputs
//     it feeds (~5.5 LDS reads + 16 FMAs per output for the 4x4 filter).  Polyphase for up > 1:
//     only the taps that hit non-zero samples are visited (FW/up x FH/up per output).  Wave
//     lanes run along x: LDS reads are conflict-free and stores are 256-byte row segments.
//   * upfirdn2d_generic -- any strides / dtype / factors / filter size: one output per
//     thread straight from global memory, visiting valid taps only.
//
//   * upfirdn2d_cl     -- channels-last (NHWC) dense tensors, no up-sampling, filters up to 4 x 4 (the blur behind every
//     transposed convolution and in front of every strided one, and the 2x decimation of the discriminator's skip path,
//     for the half-precision blocks whose convolutions run channels-last: conv2d_kernel16.h).  A thread owns one
//     16-byte channel vector of one output column and walks RPT output rows: consecutive threads = consecutive channel
//     vectors, then pixels, so every load and store is a coalesced 16-byte access; an input row is fetched once for all
//     the output rows it feeds; fp32 accumulation; the SynthesisLayer tail (noise, bias, activation, gain, clamp) rides
//     in the same pass.
//
// Roofline: HBM streaming; algorithmic bytes per call = sizeof(T) * (numel(x) + numel(y))
// (SURVEY.md section 8d).
#include <cstdlib>
#include "pg_common.h"

namespace {

using namespace pg;

struct Params {
    const void* x; const float* f; void* y;
    int N, C, inH, inW, outH, outW, fh, fw;
    int64_t xs[4], ys[4], fs[2];
    int upx, upy, dnx, dny, padx0, pady0, flip;
    float gain;
    // optional fused tail (tiled float32 path only)
    int has_ep;
    const float* noise; int64_t noise_bs; float noise_gain;
    const float* bias; float slope, act_gain, clamp;
    // optional second output (tiled float32 path, no resampling): the odd rows and columns of y, dense -- y_odd[n, c, j, i] = y[n, c, 2 j + 1, 2 i + 1]
    // (pg_upfirdn2d_with_odd_samples: what the down = 2 FIR of a ResBlock's skip path computes from the same input)
    float* y2; int y2H, y2W;
};

__device__ __forceinline__ int floor_div(int a, int b) {   // b > 0
    int q = a / b;
    return (a % b != 0 && a < 0) ? q - 1 : q;
}
__device__ __forceinline__ int pos_mod(int a, int b) {     // b > 0, result in [0, b)
    int r = a % b;
    return r < 0 ? r + b : r;
}

// ---------------------------------------------------------------- generic
template <typename T>
__global__ __launch_bounds__(256) void upfirdn2d_generic(Params p) {
    typedef typename acc_of<T>::type S;
    const T* __restrict__ x = (const T*)p.x;
    T* __restrict__ y = (T*)p.y;
    const int64_t total = (int64_t)p.N * p.C * p.outH * p.outW;
    const int64_t stride = (int64_t)gridDim.x * blockDim.x;
    for (int64_t idx = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; idx < total; idx += stride) {
        int ox = (int)(idx % p.outW);
        int64_t r = idx / p.outW;
        int oy = (int)(r % p.outH); r /= p.outH;
        int c = (int)(r % p.C);
        int n = (int)(r / p.C);
        const int uy0 = oy * p.dny - p.pady0;      // upsampled-grid coordinate of tap ky = 0
        const int ux0 = ox * p.dnx - p.padx0;
        const int ky0 = pos_mod(-uy0, p.upy);      // first tap that lands on a real sample
        const int kx0 = pos_mod(-ux0, p.upx);
        const T* xb = x + n * p.xs[0] + c * p.xs[1];
        S acc = (S)0;
        for (int ky = ky0; ky < p.fh; ky += p.upy) {
            const int iy = (uy0 + ky) / p.upy;     // exact
            if (iy < 0 || iy >= p.inH) continue;
            const int fy = p.flip ? ky : p.fh - 1 - ky;
            for (int kx = kx0; kx < p.fw; kx += p.upx) {
                const int ix = (ux0 + kx) / p.upx;
                if (ix < 0 || ix >= p.inW) continue;
                const int fx = p.flip ? kx : p.fw - 1 - kx;
                acc += (S)xb[iy * p.xs[2] + ix * p.xs[3]] * (S)p.f[fy * p.fs[0] + fx * p.fs[1]];
            }
        }
        y[n * p.ys[0] + c * p.ys[1] + oy * p.ys[2] + ox * p.ys[3]] = (T)(acc * (S)p.gain);
    }
}

// ---------------------------------------------------------------- tiled
template <int UPX, int UPY, int DNX, int DNY, int FW, int FH>
struct Tile {
    static constexpr int TOW = 64, TOH = 32, RPT = 8;                       // outputs per workgroup (w x h), rows per lane
    static constexpr int TIW = ((TOW - 1) * DNX + FW - 1) / UPX + 2;        // staged input footprint
    static constexpr int TIH = ((TOH - 1) * DNY + FH - 1) / UPY + 2;
    static constexpr int TIWP = TIW | 1;                                    // odd row pitch
};

template <typename T, int UPX, int UPY, int DNX, int DNY, int FW, int FH>
__global__ __launch_bounds__(256) void upfirdn2d_tiled(Params p, int tilesX, int tilesY) {
    typedef Tile<UPX, UPY, DNX, DNY, FW, FH> G;
    static_assert(FW % UPX == 0 && FH % UPY == 0, "padded filter must be a multiple of the up factor");
    __shared__ float sf[FH][FW];
    __shared__ float sx[G::TIH][G::TIWP];

    int bid = blockIdx.x;
    const int tx = bid % tilesX; bid /= tilesX;
    const int ty = bid % tilesY;
    const int64_t plane = bid / tilesY;
    const int t = threadIdx.x, lx = t & 63, wave = t >> 6;    // lanes of a wave run along x

    // taps: flipped (true convolution) unless p.flip, zero padded up to FH x FW
    for (int i = t; i < FH * FW; i += 256) {
        const int ky = i / FW, kx = i % FW;
        float v = 0.f;
        if (ky < p.fh && kx < p.fw) {
            const int fy = p.flip ? ky : p.fh - 1 - ky;
            const int fx = p.flip ? kx : p.fw - 1 - kx;
            v = p.f[fy * p.fs[0] + fx * p.fs[1]];
        }
        sf[ky][kx] = v;
    }

    // input footprint -> LDS, one image row per wave-iteration: 64 coalesced lanes + a short tail, no div/mod per element
    const int ox0 = tx * G::TOW, oy0 = ty * G::TOH;
    const int ix0 = floor_div(ox0 * DNX - p.padx0, UPX);
    const int iy0 = floor_div(oy0 * DNY - p.pady0, UPY);
    const T* __restrict__ xp = (const T*)p.x + plane * p.xs[1];           // rows may be pitched (xs[2] >= inW): a producer that pads its rows for aligned stores
    const int pitch = (int)p.xs[2];
    // The footprint is walked as one flat element range (256 consecutive elements per step, rows break mid-wave): every
    // wave-instruction is full -- a per-row walk spends half of its load instructions on the 3-column halo tail, and vector
    // memory instructions are what this kernel runs out of.  Two phases so that all loads are in flight together (a
    // load -> wait -> ds_write loop is latency-bound).
    constexpr int NE = G::TIH * G::TIW, NL = (NE + 255) / 256;
    float stage[NL];
#pragma unroll
    for (int i = 0; i < NL; i++) {
        const int e = t + 256 * i;
        const int r = e / G::TIW, c = e - r * G::TIW;
        const int gy = iy0 + r, gx = ix0 + c;
        const bool ok = e < NE && gy >= 0 && gy < p.inH && gx >= 0 && gx < p.inW;
        const float v = (float)xp[ok ? (int64_t)gy * pitch + gx : 0];       // unconditional load from a valid address
        stage[i] = ok ? v : 0.f;
    }
#pragma unroll
    for (int i = 0; i < NL; i++) {
        const int e = t + 256 * i;
        const int r = e / G::TIW, c = e - r * G::TIW;
        if (NE % 256 == 0 || e < NE) sx[r][c] = stage[i];
    }
    __syncthreads();

    const int ly0 = wave * G::RPT;            // each wave owns RPT adjacent output rows
    const int ox = ox0 + lx;
    T* __restrict__ yp = (T*)p.y + plane * (int64_t)p.outH * p.outW;
    float acc[G::RPT];
#pragma unroll
    for (int r = 0; r < G::RPT; r++) acc[r] = 0.f;

    if constexpr (UPX == 1 && UPY == 1) {
        // No zero-stuffing: output (ly, lx) reads sx[ly*DNY + jy][lx*DNX + jx] -- every LDS offset is a compile-time
        // constant from one base, so vertically adjacent outputs share their reads; the taps sit in registers.
        float tap[FH][FW];
#pragma unroll
        for (int jy = 0; jy < FH; jy++)
#pragma unroll
            for (int jx = 0; jx < FW; jx++) tap[jy][jx] = sf[jy][jx];
        const float* base = &sx[ly0 * DNY][lx * DNX];
#pragma unroll
        for (int rr = 0; rr < (G::RPT - 1) * DNY + FH; rr++) {             // walk the needed input rows once
            float v[FW];
#pragma unroll
            for (int jx = 0; jx < FW; jx++) v[jx] = base[rr * G::TIWP + jx];
#pragma unroll
            for (int r = 0; r < G::RPT; r++) {
                const int jy = rr - r * DNY;                               // compile-time after unrolling
                if (jy >= 0 && jy < FH) {
#pragma unroll
                    for (int jx = 0; jx < FW; jx++) acc[r] += v[jx] * tap[jy][jx];
                }
            }
        }
    } else {
        // polyphase: only the taps that land on real samples (FW/UPX x FH/UPY per output)
        const int ux0 = ox * DNX - p.padx0;
        const int kx0 = UPX == 1 ? 0 : pos_mod(-ux0, UPX);
        const int cx = floor_div(ux0 + kx0, UPX) - ix0;
#pragma unroll
        for (int r = 0; r < G::RPT; r++) {
            const int uy0 = (oy0 + ly0 + r) * DNY - p.pady0;
            const int ky0 = UPY == 1 ? 0 : pos_mod(-uy0, UPY);
            const int cy = floor_div(uy0 + ky0, UPY) - iy0;
#pragma unroll
            for (int jy = 0; jy < FH / UPY; jy++)
#pragma unroll
                for (int jx = 0; jx < FW / UPX; jx++) acc[r] += sx[cy + jy][cx + jx] * sf[ky0 + jy * UPY][kx0 + jx * UPX];
        }
    }

    const int c = (int)(plane % p.C), n = (int)(plane / p.C);
    const float bias = (p.has_ep && p.bias) ? p.bias[c] : 0.f;
#pragma unroll
    for (int r = 0; r < G::RPT; r++) {
        const int oy = oy0 + ly0 + r;
        if (ox < p.outW && oy < p.outH) {
            float v = acc[r] * p.gain;
            if (p.has_ep) {       // SynthesisLayer tail: + noise, + bias, linear/relu/lrelu, gain, clamp
                if (p.noise) v += p.noise[n * p.noise_bs + (int64_t)oy * p.outW + ox] * p.noise_gain;
                v += bias;
                v = (v > 0.f ? v : v * p.slope) * p.act_gain;
                v = fminf(fmaxf(v, -p.clamp), p.clamp);
            }
            yp[(int64_t)oy * p.outW + ox] = (T)v;
            if constexpr (sizeof(T) == 4 && UPX == 1 && UPY == 1 && DNX == 1 && DNY == 1) {
                if (p.y2 && (oy & 1) && (ox & 1) && (oy >> 1) < p.y2H && (ox >> 1) < p.y2W) p.y2[(plane * p.y2H + (oy >> 1)) * p.y2W + (ox >> 1)] = (float)v;      // (an even extent has one odd sample more than the decimated pass keeps)
            }
        }
    }
}

// ---------------------------------------------------------------- blur (no resampling), float32: separable, 16-byte accesses
// The FIR call that carries the bytes of the synthesis forward is the 4x4 blur after every up-sampling convolution.  Measured on the
// generic tiled kernel above: it is bound by VALU work, not by memory -- 16 multiply-adds per output (its 2x decimation variant, with a
// quarter of the outputs per input byte, runs at 5 TB/s; the blur at 3.9; adding the fused tail's arithmetic slows it further; wider
// tiles, 16-byte accesses and a prefetching persistent loop each changed nothing).  So this kernel cuts the arithmetic:
//   * the filter is an outer product fy (x) fx (setup_filter builds it that way; verified on the taps by every workgroup, with the
//     full 2-D evaluation as the fallback): a horizontal 4-tap pass per footprint row, then a vertical 4-tap pass -- with 8 output
//     rows per thread 9.5 multiply-adds per output instead of 16;
//   * a thread owns 4 adjacent output columns x 8 rows of a 64 x 128 tile; per footprint row one ds_read_b128 + one b64 (+ b32): the
//     7 samples its 4 columns touch (D = the tile-constant misalignment of the footprint, a template parameter so the reads stay aligned);
//   * every global access is 16 bytes: the footprint is staged as ALIGNED float4 words (the producer pads its rows to a multiple of 4
//     floats, conv2d_up2.h; words straddling the image edge fall back to guarded scalars), stores and noise loads are float4;
//   * persistent workgroups; the next tile's footprint is in flight (in registers) while this one is filtered.
typedef float f32x4f __attribute__((ext_vector_type(4)));
typedef float f32x2f __attribute__((ext_vector_type(2)));

template <int D, int TOW>
__global__ __launch_bounds__(256) void upfirdn2d_blur4(Params p, int tilesX, int tilesY, int total) {
    constexpr int TOH = 8192 / TOW, RPT = 8, IH = TOH + 3, WORDS = TOW / 4 + 2, PITCH = 4 * WORDS, CQ = TOW / 4;
    extern __shared__ __attribute__((aligned(16))) float sxf[];            // [IH][PITCH]
    const int t = threadIdx.x;
    const int pitch = (int)p.xs[2];
    constexpr int NL = (IH * WORDS + 255) / 256;
    // taps: flipped (true convolution) unless p.flip, zero padded to 4 x 4 (uniform: scalar registers)
    float tap[4][4];
#pragma unroll
    for (int ky = 0; ky < 4; ky++)
#pragma unroll
        for (int kx = 0; kx < 4; kx++) {
            const int fy = p.flip ? ky : p.fh - 1 - ky, fx = p.flip ? kx : p.fw - 1 - kx;
            tap[ky][kx] = (ky < p.fh && kx < p.fw) ? p.f[fy * p.fs[0] + fx * p.fs[1]] : 0.f;
        }
    // outer-product factors through the largest tap; `sep` = the 16 taps really are fyv[ky] * fxv[kx]
    int pk = 0, pj = 0;
    float pv = 0.f;
#pragma unroll
    for (int ky = 0; ky < 4; ky++)
#pragma unroll
        for (int kx = 0; kx < 4; kx++)
            if (fabsf(tap[ky][kx]) > fabsf(pv)) { pv = tap[ky][kx]; pk = ky; pj = kx; }
    float fxv[4], fyv[4];
    bool sep = pv != 0.f;
#pragma unroll
    for (int k = 0; k < 4; k++) {
        fxv[k] = pk == 0 ? tap[0][k] : (pk == 1 ? tap[1][k] : (pk == 2 ? tap[2][k] : tap[3][k]));
        const float col = pj == 0 ? tap[k][0] : (pj == 1 ? tap[k][1] : (pj == 2 ? tap[k][2] : tap[k][3]));
        fyv[k] = sep ? col / pv : 0.f;
    }
#pragma unroll
    for (int ky = 0; ky < 4; ky++)
#pragma unroll
        for (int kx = 0; kx < 4; kx++) sep = sep && fabsf(tap[ky][kx] - fyv[ky] * fxv[kx]) <= 1e-6f * fabsf(pv);

    auto fetch = [&](int tile, f32x4f (&stage)[NL]) __attribute__((always_inline)) {
        const int tx = tile % tilesX, ty = (tile / tilesX) % tilesY;
        const int64_t plane = tile / (tilesX * tilesY);
        const int iy0 = ty * TOH - p.pady0, a0 = tx * TOW - p.padx0 - D;             // a0: a multiple of 4 by the choice of D
        const float* __restrict__ xp = (const float*)p.x + plane * p.xs[1];
#pragma unroll
        for (int i = 0; i < NL; i++) {
            const int e = t + 256 * i;
            const int r = e / WORDS, wd = e - r * WORDS;
            const int gy = iy0 + r, gx = a0 + 4 * wd;
            const bool row_ok = e < IH * WORDS && gy >= 0 && gy < p.inH;
            f32x4f v = {0.f, 0.f, 0.f, 0.f};
            if (row_ok && gx >= 0 && gx + 4 <= p.inW) {
                v = *(const f32x4f*)(xp + (int64_t)gy * pitch + gx);
            } else if (row_ok && gx + 4 > 0 && gx < p.inW) {               // word straddles the left / right edge
#pragma unroll
                for (int k = 0; k < 4; k++)
                    if (gx + k >= 0 && gx + k < p.inW) v[k] = xp[(int64_t)gy * pitch + gx + k];
            }
            stage[i] = v;
        }
    };
    const int cx = t % CQ, ry = t / CQ;                                   // columns 4cx .. 4cx + 3, rows RPT * ry .. of the tile
    f32x4f stage[NL];
    int tile = blockIdx.x;
    if (tile < total) fetch(tile, stage);
    for (; tile < total; tile += gridDim.x) {
#pragma unroll
        for (int i = 0; i < NL; i++) {
            const int e = t + 256 * i;
            const int r = e / WORDS, wd = e - r * WORDS;
            if (e < IH * WORDS) *(f32x4f*)&sxf[r * PITCH + 4 * wd] = stage[i];
        }
        __syncthreads();
        if (tile + (int)gridDim.x < total) fetch(tile + gridDim.x, stage);
        const int tx = tile % tilesX, ty = (tile / tilesX) % tilesY;
        const int64_t plane = tile / (tilesX * tilesY);
        const int ox0 = tx * TOW, oy0 = ty * TOH;
        float acc[RPT][4];
#pragma unroll
        for (int r = 0; r < RPT; r++)
#pragma unroll
            for (int c = 0; c < 4; c++) acc[r][c] = 0.f;
#pragma unroll
        for (int rr = 0; rr < RPT + 3; rr++) {                            // footprint rows of this thread's strip
            // output column c, tap kx: staged column 4cx + c + kx + D, i.e. the samples v[D .. D + 6]
            float v[12];
            const float* row = &sxf[(RPT * ry + rr) * PITCH + 4 * cx];
            *(f32x4f*)&v[0] = *(const f32x4f*)(row);
            *(f32x4f*)&v[4] = *(const f32x4f*)(row + 4);
            if (D + 6 >= 8) { *(f32x2f*)&v[8] = *(const f32x2f*)(row + 8); }
            if (sep) {
                float h[4];
#pragma unroll
                for (int c = 0; c < 4; c++) {
                    h[c] = v[D + c] * fxv[0];
#pragma unroll
                    for (int kx = 1; kx < 4; kx++) h[c] = fmaf(v[D + c + kx], fxv[kx], h[c]);
                }
#pragma unroll
                for (int r = 0; r < RPT; r++) {
                    const int ky = rr - r;
                    if (ky >= 0 && ky < 4) {
#pragma unroll
                        for (int c = 0; c < 4; c++) acc[r][c] = fmaf(h[c], fyv[ky], acc[r][c]);
                    }
                }
            } else {
#pragma unroll
                for (int r = 0; r < RPT; r++) {
                    const int ky = rr - r;
                    if (ky >= 0 && ky < 4) {
#pragma unroll
                        for (int c = 0; c < 4; c++)
#pragma unroll
                            for (int kx = 0; kx < 4; kx++) acc[r][c] = fmaf(v[D + c + kx], tap[ky][kx], acc[r][c]);
                    }
                }
            }
        }
        const int ch = (int)(plane % p.C), n = (int)(plane / p.C);
        const float bias = (p.has_ep && p.bias) ? p.bias[ch] : 0.f;
        float* __restrict__ yp = (float*)p.y + plane * (int64_t)p.outH * p.outW;
        const int ox = ox0 + 4 * cx;
#pragma unroll
        for (int r = 0; r < RPT; r++) {
            const int oy = oy0 + RPT * ry + r;
            if (oy >= p.outH || ox >= p.outW) continue;                    // outW % 4 == 0 (host): a thread's 4 columns are in or out together
            f32x4f o = {acc[r][0] * p.gain, acc[r][1] * p.gain, acc[r][2] * p.gain, acc[r][3] * p.gain};
            if (p.has_ep) {       // SynthesisLayer tail: + noise, + bias, linear/relu/lrelu, gain, clamp
                if (p.noise) o += *(const f32x4f*)(p.noise + n * p.noise_bs + (int64_t)oy * p.outW + ox) * p.noise_gain;
#pragma unroll
                for (int c = 0; c < 4; c++) {
                    float w = o[c] + bias;
                    w = (w > 0.f ? w : w * p.slope) * p.act_gain;
                    o[c] = fminf(fmaxf(w, -p.clamp), p.clamp);
                }
            }
            *(f32x4f*)(yp + (int64_t)oy * p.outW + ox) = o;
        }
        __syncthreads();                                                   // every thread is done with the tile before the next one overwrites it
    }
}

// Returns 1 if the 16-byte blur kernel took the call (status in *st).
inline bool try_blur4(const Params& p, hipStream_t s, int* st) {
    static const bool on = [] { const char* e = getenv("PG_FIR_BLUR4"); return e ? atoi(e) != 0 : true; }();      // A/B switch
    if (!on || p.upx != 1 || p.upy != 1 || p.dnx != 1 || p.dny != 1 || p.fw > 4 || p.fh > 4) return false;
    if ((p.xs[2] & 3) != 0 || (p.xs[1] & 3) != 0 || (p.outW & 3) != 0 || !aligned16(p.x) || !aligned16(p.y)) return false;
    if (p.noise && (!aligned16(p.noise) || (p.noise_bs & 3) != 0)) return false;
    const int d = ((-p.padx0) % 4 + 4) % 4;                               // ox0 - padx0 - d is a multiple of 4 (ox0 is one of 64)
    if (d + 6 > 11) return false;
    static const int shape = [] { const char* e = getenv("PG_FIR_BLUR4_TILE"); return e ? atoi(e) : 128; }();     // tile width 64 | 128 | 256 (dev switch; 128: 263 us, 64: 272 us, 256: 261 / slower with the tail)
    const int TOW = shape == 128 ? 128 : (shape == 256 ? 256 : 64), TOH = 8192 / TOW;
    if (p.outW % TOW != 0 || p.outH % TOH != 0) return false;             // large planes only: on the 128^2 and smaller layers the big tiles are mostly empty (measured slower)
    const int tilesX = (p.outW + TOW - 1) / TOW, tilesY = (p.outH + TOH - 1) / TOH;
    const int64_t tiles = (int64_t)tilesX * tilesY * p.N * p.C;
    if (tiles > 0x7fffffffLL) { *st = PG_ERR_TOO_LARGE; return true; }
    const int total = (int)tiles;
    const size_t lds = (size_t)(TOH + 3) * (TOW + 8) * sizeof(float);     // ~37 KB: 4 workgroups per CU
    const int64_t blocks = tiles < (int64_t)num_cu() * 4 ? tiles : (int64_t)num_cu() * 4;
#define PG_B4(DD) \
    if (TOW == 64) hipLaunchKernelGGL((upfirdn2d_blur4<DD, 64>), dim3((unsigned)blocks), dim3(256), lds, s, p, tilesX, tilesY, total); \
    else if (TOW == 128) hipLaunchKernelGGL((upfirdn2d_blur4<DD, 128>), dim3((unsigned)blocks), dim3(256), lds, s, p, tilesX, tilesY, total); \
    else hipLaunchKernelGGL((upfirdn2d_blur4<DD, 256>), dim3((unsigned)blocks), dim3(256), lds, s, p, tilesX, tilesY, total);
    switch (d) {
        case 0: PG_B4(0) break;
        case 1: PG_B4(1) break;
        case 2: PG_B4(2) break;
        default: PG_B4(3) break;
    }
#undef PG_B4
    *st = launch_status();
    return true;
}

template <typename T, int UPX, int UPY, int DNX, int DNY, int FW, int FH>
int launch_tiled(const Params& p, hipStream_t s) {
    typedef Tile<UPX, UPY, DNX, DNY, FW, FH> G;
    const int tilesX = (p.outW + G::TOW - 1) / G::TOW, tilesY = (p.outH + G::TOH - 1) / G::TOH;
    const int64_t blocks = (int64_t)tilesX * tilesY * p.N * p.C;
    if (blocks > 0x7fffffffLL) return PG_ERR_TOO_LARGE;
    hipLaunchKernelGGL((upfirdn2d_tiled<T, UPX, UPY, DNX, DNY, FW, FH>), dim3((unsigned)blocks), dim3(256), 0, s, p, tilesX, tilesY);
    return launch_status();
}

// Returns 1 if a tiled specialisation took the call (status in *st).
template <typename T>
bool try_tiled(const Params& p, hipStream_t s, int* st) {
#define PG_SPEC(UX, UY, DX, DY, W, H)                                                               \
    if (p.upx == UX && p.upy == UY && p.dnx == DX && p.dny == DY && p.fw <= W && p.fh <= H) {        \
        *st = launch_tiled<T, UX, UY, DX, DY, W, H>(p, s);                                           \
        return true;                                                                                 \
    }
    // 2-D filters (the generator's [1,3,3,1] x [1,3,3,1] and anything up to 8 x 8)
    PG_SPEC(1, 1, 1, 1, 4, 4) PG_SPEC(2, 2, 1, 1, 4, 4) PG_SPEC(1, 1, 2, 2, 4, 4)
    PG_SPEC(1, 1, 1, 1, 8, 8) PG_SPEC(2, 2, 1, 1, 8, 8) PG_SPEC(1, 1, 2, 2, 8, 8)
    // separable passes (one axis at a time; upfirdn2d.py:239-240), up to 16 taps
    PG_SPEC(1, 1, 1, 1, 16, 1) PG_SPEC(1, 1, 1, 1, 1, 16)
    PG_SPEC(2, 1, 1, 1, 16, 1) PG_SPEC(1, 2, 1, 1, 1, 16)
    PG_SPEC(1, 1, 2, 1, 16, 1) PG_SPEC(1, 1, 1, 2, 1, 16)
#undef PG_SPEC
    return false;
}

// ---------------------------------------------------------------- channels-last
// 16 bytes of channels as a register vector (plain ext-vector registers: element-wise writes to a struct of 16-bit values
// would send it to scratch), widened to / narrowed from fp32 with bit operations
typedef unsigned u32x4 __attribute__((ext_vector_type(4)));
typedef float f32x2v __attribute__((ext_vector_type(2)));
typedef __bf16 bf16x2v __attribute__((ext_vector_type(2)));
typedef _Float16 f16x2v __attribute__((ext_vector_type(2)));
template <typename T> struct Vec16;
template <> struct Vec16<float> {
    static constexpr int N = 4;
    static __device__ __forceinline__ void widen(u32x4 r, float (&o)[4]) {
#pragma unroll
        for (int i = 0; i < 4; i++) { const unsigned e = r[i]; o[i] = __builtin_bit_cast(float, e); }   // (bit_cast of `r[i]` itself reads element 0: a vector element is not an addressable object)
    }
    static __device__ __forceinline__ u32x4 narrow(const float (&v)[4]) {
        return u32x4{__builtin_bit_cast(unsigned, v[0]), __builtin_bit_cast(unsigned, v[1]), __builtin_bit_cast(unsigned, v[2]), __builtin_bit_cast(unsigned, v[3])};
    }
};
template <> struct Vec16<bf16_t> {
    static constexpr int N = 8;
    static __device__ __forceinline__ void widen(u32x4 r, float (&o)[8]) {
#pragma unroll
        for (int i = 0; i < 4; i++) { const unsigned e = r[i]; const bf16x2v h = __builtin_bit_cast(bf16x2v, e); o[2 * i] = (float)h[0]; o[2 * i + 1] = (float)h[1]; }
    }
    static __device__ __forceinline__ u32x4 narrow(const float (&v)[8]) {
        u32x4 r;
#pragma unroll
        for (int i = 0; i < 4; i++) { const f32x2v t = {v[2 * i], v[2 * i + 1]}; r[i] = __builtin_bit_cast(unsigned, __builtin_convertvector(t, bf16x2v)); }
        return r;
    }
};
template <> struct Vec16<f16_t> {
    static constexpr int N = 8;
    static __device__ __forceinline__ void widen(u32x4 r, float (&o)[8]) {
#pragma unroll
        for (int i = 0; i < 4; i++) { const unsigned e = r[i]; const f16x2v h = __builtin_bit_cast(f16x2v, e); o[2 * i] = (float)h[0]; o[2 * i + 1] = (float)h[1]; }
    }
    static __device__ __forceinline__ u32x4 narrow(const float (&v)[8]) {
        u32x4 r;
#pragma unroll
        for (int i = 0; i < 4; i++) { const f32x2v t = {v[2 * i], v[2 * i + 1]}; r[i] = __builtin_bit_cast(unsigned, __builtin_convertvector(t, f16x2v)); }
        return r;
    }
};

template <typename T, int FH, int FW, int DN, int RPT>
__global__ __launch_bounds__(256, 2) void upfirdn2d_cl(Params p) {
    constexpr int V = Vec16<T>::N;
    const int CV = p.C / V;
    const int64_t idx = (int64_t)blockIdx.x * 256 + threadIdx.x;
    const int cv = (int)(idx % CV);
    const int ox = (int)(idx / CV);
    if (ox >= p.outW) return;
    const int n = blockIdx.z, oy0 = blockIdx.y * RPT;
    const int ix0 = ox * DN - p.padx0, iy0 = oy0 * DN - p.pady0;
    float tap[FH][FW];                                   // flipped (true convolution) unless p.flip, zero padded to FH x FW
#pragma unroll
    for (int ky = 0; ky < FH; ky++)
#pragma unroll
        for (int kx = 0; kx < FW; kx++) {
            const int fy = p.flip ? ky : p.fh - 1 - ky, fx = p.flip ? kx : p.fw - 1 - kx;
            tap[ky][kx] = (ky < p.fh && kx < p.fw) ? p.f[fy * p.fs[0] + fx * p.fs[1]] : 0.f;
        }
    const T* __restrict__ xn = (const T*)p.x + (int64_t)n * p.xs[0] + cv * V;      // rows / images may be pitched (xs[2] >= inW * C): a view into a padded producer buffer
    const int64_t xrow = p.xs[2];
    float acc[RPT][V];
#pragma unroll
    for (int r = 0; r < RPT; r++)
#pragma unroll
        for (int c = 0; c < V; c++) acc[r][c] = 0.f;
    auto load_row = [&](int rr, u32x4 (&raw)[FW]) __attribute__((always_inline)) {
        const int iy = iy0 + rr;
        const bool row_ok = iy >= 0 && iy < p.inH;
#pragma unroll
        for (int kx = 0; kx < FW; kx++) {
            const int ix = ix0 + kx;
            const bool ok = row_ok && ix >= 0 && ix < p.inW;
            u32x4 v = *(const u32x4*)(xn + (int64_t)(ok ? iy : 0) * xrow + (int64_t)(ok ? ix : 0) * p.C);    // always a valid address
#pragma unroll
            for (int e = 0; e < 4; e++) v[e] = ok ? v[e] : 0u;
            raw[kx] = v;
        }
    };
    constexpr int NROWS = (RPT - 1) * DN + FH;            // input rows of the strip's footprint: each is fetched once,
    u32x4 cur[FW], nxt[FW];                               // one row ahead of the one being accumulated
    load_row(0, cur);
#pragma unroll
    for (int rr = 0; rr < NROWS; rr++) {
        if (rr + 1 < NROWS) load_row(rr + 1, nxt);
#pragma unroll
        for (int kx = 0; kx < FW; kx++) {
            float xv[V];
            Vec16<T>::widen(cur[kx], xv);
#pragma unroll
            for (int r = 0; r < RPT; r++) {
                const int ky = rr - r * DN;               // compile-time after unrolling
                if (ky >= 0 && ky < FH) {
#pragma unroll
                    for (int c = 0; c < V; c++) acc[r][c] = fmaf(xv[c], tap[ky][kx], acc[r][c]);
                }
            }
        }
#pragma unroll
        for (int kx = 0; kx < FW; kx++) cur[kx] = nxt[kx];
        __builtin_amdgcn_sched_barrier(0);                // keep the scheduler from hoisting every row's loads to the top (spills)
    }
    float bias[V];
#pragma unroll
    for (int c = 0; c < V; c++) bias[c] = (p.has_ep && p.bias) ? p.bias[cv * V + c] : 0.f;
    T* __restrict__ yn = (T*)p.y + (int64_t)n * p.outH * p.outW * p.C + cv * V;
#pragma unroll
    for (int r = 0; r < RPT; r++) {
        const int oy = oy0 + r;
        if (oy >= p.outH) break;
        const float nz = (p.has_ep && p.noise) ? p.noise[n * p.noise_bs + (int64_t)oy * p.outW + ox] * p.noise_gain : 0.f;
        float out[V];
#pragma unroll
        for (int c = 0; c < V; c++) {
            float v = acc[r][c] * p.gain;
            if (p.has_ep) {
                v += nz + bias[c];
                v = (v > 0.f ? v : v * p.slope) * p.act_gain;
                v = fminf(fmaxf(v, -p.clamp), p.clamp);
            }
            out[c] = v;
        }
        *(u32x4*)(yn + ((int64_t)oy * p.outW + ox) * p.C) = Vec16<T>::narrow(out);
    }
}

// Channels-last zero-insertion up-sampling by 2 (no decimation): the gradient of the discriminator's FIR down-sampling
// (Upfirdn2dCuda.backward swaps up and down: upfirdn2d.py:245-264) and the skip-image up-sampling of 16-bit blocks.  Polyphase: an
// output pixel touches only the taps that land on real samples -- (FH/2) x (FW/2) of them -- so a thread (one output pixel x one
// 16-byte channel vector) reads 2 x 2 input vectors for a 4-tap filter.
template <typename T, int FH, int FW>
__global__ __launch_bounds__(256) void upfirdn2d_cl_up2(Params p) {
    constexpr int V = Vec16<T>::N;
    const int CV = p.C / V;
    const int64_t idx = (int64_t)blockIdx.x * 256 + threadIdx.x;
    const int cv = (int)(idx % CV);
    const int ox = (int)(idx / CV);
    if (ox >= p.outW) return;
    const int n = blockIdx.z, oy = blockIdx.y;
    const int ux0 = ox - p.padx0, uy0 = oy - p.pady0;        // up-sampled grid coordinate of tap 0
    const int kx0 = ux0 & 1, ky0 = uy0 & 1;                  // first tap on a real sample: (u0 + k) even  (two's complement: works for negatives)
    const int ix0 = (ux0 + kx0) >> 1, iy0 = (uy0 + ky0) >> 1;
    const T* __restrict__ xn = (const T*)p.x + (int64_t)n * p.inH * p.inW * p.C + cv * V;
    float acc[V];
#pragma unroll
    for (int c = 0; c < V; c++) acc[c] = 0.f;
#pragma unroll
    for (int jy = 0; jy < FH / 2; jy++) {
        const int ky = ky0 + 2 * jy, iy = iy0 + jy;
        const int fy = p.flip ? ky : p.fh - 1 - ky;
#pragma unroll
        for (int jx = 0; jx < FW / 2; jx++) {
            const int kx = kx0 + 2 * jx, ix = ix0 + jx;
            const int fx = p.flip ? kx : p.fw - 1 - kx;
            const bool ok = ky < p.fh && kx < p.fw && iy >= 0 && iy < p.inH && ix >= 0 && ix < p.inW;
            const float tap = ok ? p.f[fy * p.fs[0] + fx * p.fs[1]] : 0.f;
            const u32x4 v = *(const u32x4*)(xn + ((int64_t)(ok ? iy : 0) * p.inW + (ok ? ix : 0)) * p.C);    // always a valid address
            float xv[V];
            Vec16<T>::widen(v, xv);
#pragma unroll
            for (int c = 0; c < V; c++) acc[c] = fmaf(xv[c], tap, acc[c]);
        }
    }
    float out[V];
#pragma unroll
    for (int c = 0; c < V; c++) out[c] = acc[c] * p.gain;
    T* __restrict__ yn = (T*)p.y + (int64_t)n * p.outH * p.outW * p.C + cv * V;
    *(u32x4*)(yn + ((int64_t)oy * p.outW + ox) * p.C) = Vec16<T>::narrow(out);
}

template <typename T>
bool try_channels_last(const Params& p, hipStream_t s, int* st) {
    constexpr int V = Vec16<T>::N;
    const bool x_dense = p.xs[2] == (int64_t)p.inW * p.C && p.xs[0] == (int64_t)p.inH * p.inW * p.C;
    // the blur / decimation kernel also takes a PITCHED channels-last input (a [:, :, :H, :W] view of a larger NHWC buffer: the merged transposed-conv phases
    // of the 16-bit up-sampling layers write a (2H+2) x (2W+2) buffer whose first 2H+1 rows / columns are the transposed convolution's result)
    const bool x_pitched = p.xs[2] >= (int64_t)p.inW * p.C && p.xs[0] >= (int64_t)p.inH * p.xs[2] && p.xs[2] % V == 0 && p.xs[0] % V == 0;
    const bool dense_cl = p.xs[1] == 1 && p.xs[3] == p.C && (x_dense || x_pitched) &&
                          p.ys[1] == 1 && p.ys[3] == p.C && p.ys[2] == (int64_t)p.outW * p.C && p.ys[0] == (int64_t)p.outH * p.outW * p.C;
    if (!dense_cl || p.C % V != 0 || p.fw > 4 || p.fh > 4 || !aligned16(p.x) || !aligned16(p.y) || p.N > 65535) return false;
    if (p.upx == 2 && p.upy == 2 && p.dnx == 1 && p.dny == 1 && !p.has_ep) {
        if (!x_dense) return false;
        const int64_t bx2 = ((int64_t)p.outW * (p.C / V) + 255) / 256;
        if (bx2 > 0x7fffffffLL || p.outH > 65535) return false;
        hipLaunchKernelGGL((upfirdn2d_cl_up2<T, 4, 4>), dim3((unsigned)bx2, (unsigned)p.outH, (unsigned)p.N), dim3(256), 0, s, p);
        *st = launch_status();
        return true;
    }
    if (p.upx != 1 || p.upy != 1 || p.dnx != p.dny || p.dnx > 2) return false;
    constexpr int RPT = 4;                                 // output rows per thread: (RPT*DN + 3) x 4 16-byte loads in flight
    const int64_t bx = ((int64_t)p.outW * (p.C / V) + 255) / 256;
    const int by = (p.outH + RPT - 1) / RPT;
    if (bx > 0x7fffffffLL || by > 65535) return false;
    const dim3 grid((unsigned)bx, (unsigned)by, (unsigned)p.N);
    if (p.dnx == 1) hipLaunchKernelGGL((upfirdn2d_cl<T, 4, 4, 1, RPT>), grid, dim3(256), 0, s, p);
    else hipLaunchKernelGGL((upfirdn2d_cl<T, 4, 4, 2, RPT>), grid, dim3(256), 0, s, p);
    *st = launch_status();
    return true;
}

template <typename T>
int run(const Params& p, hipStream_t s, bool allow_tiled) {
    // planes contiguous in (n, c) order, rows contiguous or pitched, output dense
    const bool dense_nchw = p.xs[3] == 1 && p.xs[2] >= p.inW && p.xs[2] <= 0x7fffffffLL && p.xs[1] == (int64_t)p.inH * p.xs[2] &&
                            p.xs[0] == (int64_t)p.C * p.xs[1] && p.ys[3] == 1 && p.ys[2] == p.outW &&
                            p.ys[1] == (int64_t)p.outH * p.outW && p.ys[0] == (int64_t)p.C * p.outH * p.outW;
    int st = PG_OK;
    if (p.y2) {                                          // the second output exists on the tiled float32 kernel without resampling only
        if constexpr (sizeof(T) == 4) {
            if (allow_tiled && dense_nchw && !p.has_ep && p.upx == 1 && p.upy == 1 && p.dnx == 1 && p.dny == 1 && try_tiled<T>(p, s, &st)) return st;
        }
        return PG_ERR_UNSUPPORTED;
    }
    if constexpr (sizeof(T) == 4) { if (allow_tiled && dense_nchw && try_blur4(p, s, &st)) return st; }
    if (allow_tiled && dense_nchw && (!p.has_ep || sizeof(T) == 4) && try_tiled<T>(p, s, &st)) return st;
    if constexpr (sizeof(T) <= 4) { if (allow_tiled && try_channels_last<T>(p, s, &st)) return st; }
    if (p.has_ep) return PG_ERR_UNSUPPORTED;
    const int64_t total = (int64_t)p.N * p.C * p.outH * p.outW;
    int64_t blocks = (total + 255) / 256;
    if (blocks > max_stream_blocks() * 4) blocks = max_stream_blocks() * 4;
    hipLaunchKernelGGL((upfirdn2d_generic<T>), dim3((unsigned)blocks), dim3(256), 0, s, p);
    return launch_status();
}

}  // namespace

PG_EXPORT int pg_upfirdn2d_abi_version(void) { return PG_ABI_VERSION; }

static int upfirdn2d_impl(const void* x, const float* f, void* y, int dtype,
                          int N, int C, int inH, int inW, const int64_t xstride[4],
                          int fh, int fw, const int64_t fstride[2],
                          int outH, int outW, const int64_t ystride[4],
                          int upx, int upy, int downx, int downy, int padx0, int pady0,
                          int flip, float gain, const pg_fir_epilogue* ep, void* stream, float* y_odd = nullptr) {
    if (!x || !f || !y || !xstride || !fstride || !ystride) return PG_ERR_INVALID_ARG;
    if (N <= 0 || C <= 0 || inH <= 0 || inW <= 0 || outH <= 0 || outW <= 0 || fh <= 0 || fw <= 0) return PG_ERR_INVALID_ARG;
    if (upx < 1 || upy < 1 || downx < 1 || downy < 1) return PG_ERR_INVALID_ARG;
    // 32-bit coordinate maths inside the kernels (the reference has the same INT_MAX limits, upfirdn2d.cpp:22-23,36)
    if ((int64_t)inW * upx + 2LL * (fw + 1) + (padx0 < 0 ? -(int64_t)padx0 : padx0) > 0x7fffffffLL ||
        (int64_t)inH * upy + 2LL * (fh + 1) + (pady0 < 0 ? -(int64_t)pady0 : pady0) > 0x7fffffffLL)
        return PG_ERR_TOO_LARGE;
    Params p;
    p.x = x; p.f = f; p.y = y;
    p.N = N; p.C = C; p.inH = inH; p.inW = inW; p.outH = outH; p.outW = outW; p.fh = fh; p.fw = fw;
    for (int i = 0; i < 4; i++) { p.xs[i] = xstride[i]; p.ys[i] = ystride[i]; }
    p.fs[0] = fstride[0]; p.fs[1] = fstride[1];
    p.upx = upx; p.upy = upy; p.dnx = downx; p.dny = downy; p.padx0 = padx0; p.pady0 = pady0;
    p.flip = flip ? 1 : 0; p.gain = gain;
    p.y2 = y_odd; p.y2H = (outH - 1) / 2; p.y2W = (outW - 1) / 2;
    if (y_odd && (dtype != PG_F32 || p.y2H <= 0 || p.y2W <= 0)) return PG_ERR_UNSUPPORTED;
    p.has_ep = 0; p.noise = nullptr; p.bias = nullptr; p.noise_bs = 0; p.noise_gain = 0.f; p.slope = 1.f; p.act_gain = 1.f; p.clamp = __builtin_inff();
    if (ep) {
        if (dtype == PG_F64) return PG_ERR_UNSUPPORTED;
        if (ep->act != 0 && (ep->act < PG_ACT_LINEAR || ep->act > PG_ACT_LRELU)) return PG_ERR_UNSUPPORTED;
        p.has_ep = 1; p.noise = ep->noise; p.noise_bs = ep->noise_batch_stride; p.noise_gain = ep->noise_gain; p.bias = ep->bias;
        p.slope = (ep->act == PG_ACT_RELU) ? 0.f : (ep->act == PG_ACT_LRELU ? ep->alpha : 1.f);
        p.act_gain = ep->act_gain == 0.f ? 1.f : ep->act_gain;
        p.clamp = ep->clamp >= 0.f ? ep->clamp : __builtin_inff();
    }
    hipStream_t s = (hipStream_t)stream;
    switch (dtype) {
        case PG_F32: return run<float>(p, s, true);
        case PG_F16: return run<pg::f16_t>(p, s, true);
        case PG_BF16: return run<pg::bf16_t>(p, s, true);
        case PG_F64: return run<double>(p, s, false);
    }
    return PG_ERR_INVALID_ARG;
}

PG_EXPORT int pg_upfirdn2d(const void* x, const float* f, void* y, int dtype,
                           int N, int C, int inH, int inW, const int64_t xstride[4],
                           int fh, int fw, const int64_t fstride[2],
                           int outH, int outW, const int64_t ystride[4],
                           int upx, int upy, int downx, int downy, int padx0, int pady0,
                           int flip, float gain, void* stream) {
    return upfirdn2d_impl(x, f, y, dtype, N, C, inH, inW, xstride, fh, fw, fstride, outH, outW, ystride,
                          upx, upy, downx, downy, padx0, pady0, flip, gain, nullptr, stream);
}

PG_EXPORT int pg_upfirdn2d_bias_act(const void* x, const float* f, void* y, int dtype,
                                    int N, int C, int inH, int inW, const int64_t xstride[4],
                                    int fh, int fw, const int64_t fstride[2],
                                    int outH, int outW, const int64_t ystride[4],
                                    int upx, int upy, int downx, int downy, int padx0, int pady0,
                                    int flip, float gain, const pg_fir_epilogue* epilogue, void* stream) {
    if (!epilogue) return PG_ERR_INVALID_ARG;
    return upfirdn2d_impl(x, f, y, dtype, N, C, inH, inW, xstride, fh, fw, fstride, outH, outW, ystride,
                          upx, upy, downx, downy, padx0, pady0, flip, gain, epilogue, stream);
}

/* Round 6 -- pg_upfirdn2d that also writes the odd rows and columns of its output as a dense tensor: y_odd[n, c, j, i] = y[n, c, 2 j + 1, 2 i + 1],
 * [N, C, (outH - 1) / 2, (outW - 1) / 2].  A ResBlock with down = 2 (networks.py:1586-1621) filters its input twice -- padding 2 in front of the strided 3x3
 * convolution (conv2d_resample.py:119-122), padding 1 and every second sample in front of the 1x1 skip convolution (conv2d_resample.py:107-110) -- and the second
 * result is exactly these samples of the first.  float32, no resampling, the tiled kernel's filter sizes, dense NCHW: anything else PG_ERR_UNSUPPORTED. */
PG_EXPORT int pg_upfirdn2d_with_odd_samples(const void* x, const float* f, void* y, float* y_odd, int dtype,
                                            int N, int C, int inH, int inW, const int64_t xstride[4],
                                            int fh, int fw, const int64_t fstride[2],
                                            int outH, int outW, const int64_t ystride[4],
                                            int padx0, int pady0, int flip, float gain, void* stream) {
    if (!y_odd) return PG_ERR_INVALID_ARG;
    return upfirdn2d_impl(x, f, y, dtype, N, C, inH, inW, xstride, fh, fw, fstride, outH, outW, ystride,
                          1, 1, 1, 1, padx0, pady0, flip, gain, nullptr, stream, y_odd);
}
