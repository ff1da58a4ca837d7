// backbone.hip — channel-last helper kernels around the torch-ROCm backbones (A8/A9 callers).
//
// The profile of the first end-to-end build (profiles/r01_bench_step_kernels_v1.csv) showed that
// MIOpen runs every fp32 depthwise 3x3 (MiT Mix-FFN `dwconv`, smp SeparableConv2d) through its
// naive reference kernel (12.9 ms / step) and that eval-mode BatchNorm + ReLU + residual add are
// three separate full passes (≈20 ms / step).  Both are HBM-bound elementwise / stencil work:
//   * awseg_dwconv3x3_nhwc : depthwise 3x3 (stride 1, dilation d, zero pad d) on [B,H,W,C] with a
//                            fused per-channel bias and activation (none / ReLU / exact GELU);
//   * awseg_bias_act_nhwc  : y = act(x + bias[c] (+ residual)) in place — the epilogue of a
//                            convolution whose BatchNorm scale has been folded into its weights.
// float4 over the channel dimension (C % 4 == 0): 16 B per lane, 64 lanes = one 1 KiB line.
#include "awseg_common.h"

namespace {

constexpr int kThreads = 256;

// erf to 1.5e-7 absolute (Abramowitz & Stegun 7.1.26: 1 - (a1 t + ... + a5 t^5) exp(-x^2), t = 1 / (1 + p |x|)) with the hardware
// reciprocal and exp2: 14 vector instructions, two of them transcendental.  The library erff is ~40 (two range branches, both
// evaluated and selected): the depthwise 3x3 + GELU kernel spent 160 of its ~200 instructions per output quad there and was
// bound by them, not by memory.  GELU error <= 0.5 |x| x 3e-7: two orders inside the 1e-4 gate (AWSEG_GELU_EXACT=1: erff).
__device__ __forceinline__ float erf_as(float x)
{
    const float ax = __builtin_fabsf(x);
    const float t = __builtin_amdgcn_rcpf(fmaf(0.3275911f, ax, 1.0f));
    float p = fmaf(1.061405429f, t, -1.453152027f);
    p = fmaf(p, t, 1.421413741f);
    p = fmaf(p, t, -0.284496736f);
    p = fmaf(p, t, 0.254829592f);
    p = p * t;
    const float e = __builtin_amdgcn_exp2f(-1.4426950408889634f * ax * ax);
    const float r = fmaf(-p, e, 1.0f);
    return __builtin_copysignf(r, x);
}
__device__ __forceinline__ float act1(float v, int act)
{
    if (act == 1) return v > 0.f ? v : 0.f;
    if (act == 2) return 0.5f * v * (1.0f + erf_as(v * 0.70710678118654752440f));   // torch GELU (erf form), fast erf
    if (act == 3) return 0.5f * v * (1.0f + erff(v * 0.70710678118654752440f));     // the same with the library erff
    return v;
}
__device__ __forceinline__ float4 act4(float4 v, int act)
{
    return make_float4(act1(v.x, act), act1(v.y, act), act1(v.z, act), act1(v.w, act));
}

__global__ __launch_bounds__(kThreads)
void dwconv3x3_nhwc_kernel(const float* __restrict__ x, int64_t batch, int H, int W, int C, int dil,
                           const float* __restrict__ w9, const float* __restrict__ bias, int act,
                           float* __restrict__ out)
{
    const int c4n = C / 4;
    const int64_t total = batch * H * W * c4n;
    for (int64_t i = (int64_t)blockIdx.x * kThreads + threadIdx.x; i < total; i += (int64_t)gridDim.x * kThreads) {
        const int c4 = (int)(i % c4n);
        const int64_t p = i / c4n;
        const int xx = (int)(p % W);
        const int64_t t = p / W;
        const int yy = (int)(t % H);
        const int64_t b = t / H;
        const float* xb = x + b * (int64_t)H * W * C;
        float4 acc = bias ? *reinterpret_cast<const float4*>(bias + c4 * 4) : make_float4(0.f, 0.f, 0.f, 0.f);
#pragma unroll
        for (int ky = 0; ky < 3; ++ky) {
            const int sy = yy + (ky - 1) * dil;
            if (sy < 0 || sy >= H) continue;
#pragma unroll
            for (int kx = 0; kx < 3; ++kx) {
                const int sx = xx + (kx - 1) * dil;
                if (sx < 0 || sx >= W) continue;
                const float4 v = *reinterpret_cast<const float4*>(xb + ((int64_t)sy * W + sx) * C + c4 * 4);
                const float4 k = *reinterpret_cast<const float4*>(w9 + (ky * 3 + kx) * C + c4 * 4);
                acc.x = fmaf(v.x, k.x, acc.x); acc.y = fmaf(v.y, k.y, acc.y);
                acc.z = fmaf(v.z, k.z, acc.z); acc.w = fmaf(v.w, k.w, acc.w);
            }
        }
        *reinterpret_cast<float4*>(out + p * C + c4 * 4) = act4(acc, act);
    }
}

// dilation-1 fast path: one lane walks a strip of SX consecutive pixels of one row for its four
// channels, keeping the 3x3 weights (9 float4) and a rolling 3-column window (9 float4) in
// registers: 3 new 16-byte loads per output instead of 9 + 9 (the per-CU L1 was the limiter).
template <int SX>
__global__ __launch_bounds__(kThreads)
void dwconv3x3_nhwc_strip_kernel(const float* __restrict__ x, int64_t batch, int H, int W, int C,
                                 const float* __restrict__ w9, const float* __restrict__ bias, int act,
                                 float* __restrict__ out)
{
    const int c4n = C / 4;
    const int nsx = (W + SX - 1) / SX;
    const int64_t total = batch * H * nsx * c4n;
    for (int64_t i = (int64_t)blockIdx.x * kThreads + threadIdx.x; i < total; i += (int64_t)gridDim.x * kThreads) {
        const int c4 = (int)(i % c4n);
        int64_t t = i / c4n;
        const int xs = (int)(t % nsx); t /= nsx;
        const int yy = (int)(t % H);
        const int64_t b = t / H;
        const float* xb = x + b * (int64_t)H * W * C + c4 * 4;
        float4 k[9];
#pragma unroll
        for (int j = 0; j < 9; ++j) k[j] = *reinterpret_cast<const float4*>(w9 + j * C + c4 * 4);
        const float4 bz = bias ? *reinterpret_cast<const float4*>(bias + c4 * 4) : make_float4(0.f, 0.f, 0.f, 0.f);
        const float4 zero = make_float4(0.f, 0.f, 0.f, 0.f);
        auto col = [&](int sx, float4* v) {
#pragma unroll
            for (int ky = 0; ky < 3; ++ky) {
                const int sy = yy + ky - 1;
                v[ky] = (sx >= 0 && sx < W && sy >= 0 && sy < H) ? *reinterpret_cast<const float4*>(xb + ((int64_t)sy * W + sx) * C) : zero;
            }
        };
        const int x0 = xs * SX;
        float4 c0[3], c1[3], c2[3];
        col(x0 - 1, c0); col(x0, c1);
#pragma unroll
        for (int u = 0; u < SX; ++u) {
            const int xx = x0 + u;
            col(xx + 1, c2);
            if (xx < W) {
                float4 acc = bz;
#pragma unroll
                for (int ky = 0; ky < 3; ++ky) {
                    const float4 a = c0[ky], m = c1[ky], d = c2[ky];
                    const float4 ka = k[ky * 3], km = k[ky * 3 + 1], kd = k[ky * 3 + 2];
                    acc.x = fmaf(a.x, ka.x, acc.x); acc.y = fmaf(a.y, ka.y, acc.y); acc.z = fmaf(a.z, ka.z, acc.z); acc.w = fmaf(a.w, ka.w, acc.w);
                    acc.x = fmaf(m.x, km.x, acc.x); acc.y = fmaf(m.y, km.y, acc.y); acc.z = fmaf(m.z, km.z, acc.z); acc.w = fmaf(m.w, km.w, acc.w);
                    acc.x = fmaf(d.x, kd.x, acc.x); acc.y = fmaf(d.y, kd.y, acc.y); acc.z = fmaf(d.z, kd.z, acc.z); acc.w = fmaf(d.w, kd.w, acc.w);
                }
                *reinterpret_cast<float4*>(out + ((b * H + yy) * (int64_t)W + xx) * C + c4 * 4) = act4(acc, act);
            }
#pragma unroll
            for (int ky = 0; ky < 3; ++ky) { c0[ky] = c1[ky]; c1[ky] = c2[ky]; }
        }
    }
}

// (Measured and dropped in round 3: the input rows staged through an LDS ring by a (image, 32-channel slice, column tile, row band)
// block, the way aspp_dw3_lds_kernel does it — 0.31 ms with 128-column tiles, 0.275 with 64, against 0.274 for the kernel below.)
// Two output rows per lane: the rolling window holds 4 input rows x 3 columns, a new column costs 4 loads and
// finishes 2 outputs — 2.5 loads per output over a strip of 8 instead of 3.75 (the one-row form is bound by the
// vector-memory pipe, not by HBM: every input row is fetched by the lanes of three output rows).
template <int SX>
__global__ __launch_bounds__(kThreads)
void dwconv3x3_nhwc_strip2_kernel(const float* __restrict__ x, int64_t batch, int H, int W, int C,
                                  const float* __restrict__ w9, const float* __restrict__ bias, int act,
                                  float* __restrict__ out)
{
    const int c4n = C / 4;
    const int nsx = (W + SX - 1) / SX, nry = (H + 1) / 2;
    const int64_t total = batch * nry * nsx * c4n;
    for (int64_t i = (int64_t)blockIdx.x * kThreads + threadIdx.x; i < total; i += (int64_t)gridDim.x * kThreads) {
        const int c4 = (int)(i % c4n);
        int64_t t = i / c4n;
        const int xs = (int)(t % nsx); t /= nsx;
        const int yy = (int)(t % nry) * 2;
        const int64_t b = t / nry;
        const float* xb = x + b * (int64_t)H * W * C + c4 * 4;
        float4 k[9];
#pragma unroll
        for (int j = 0; j < 9; ++j) k[j] = *reinterpret_cast<const float4*>(w9 + j * C + c4 * 4);
        const float4 bz = bias ? *reinterpret_cast<const float4*>(bias + c4 * 4) : make_float4(0.f, 0.f, 0.f, 0.f);
        const float4 zero = make_float4(0.f, 0.f, 0.f, 0.f);
        auto col = [&](int sx, float4* v) {
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const int sy = yy + r - 1;
                v[r] = (sx >= 0 && sx < W && sy >= 0 && sy < H) ? *reinterpret_cast<const float4*>(xb + ((int64_t)sy * W + sx) * C) : zero;
            }
        };
        const int x0 = xs * SX;
        float4 c0[4], c1[4], c2[4];
        col(x0 - 1, c0); col(x0, c1);
#pragma unroll
        for (int u = 0; u < SX; ++u) {
            const int xx = x0 + u;
            col(xx + 1, c2);
            if (xx < W) {
#pragma unroll
                for (int o = 0; o < 2; ++o) {
                    if (yy + o >= H) break;
                    float4 acc = bz;
#pragma unroll
                    for (int ky = 0; ky < 3; ++ky) {
                        const float4 a = c0[ky + o], m = c1[ky + o], d = c2[ky + o];
                        const float4 ka = k[ky * 3], km = k[ky * 3 + 1], kd = k[ky * 3 + 2];
                        acc.x = fmaf(a.x, ka.x, acc.x); acc.y = fmaf(a.y, ka.y, acc.y); acc.z = fmaf(a.z, ka.z, acc.z); acc.w = fmaf(a.w, ka.w, acc.w);
                        acc.x = fmaf(m.x, km.x, acc.x); acc.y = fmaf(m.y, km.y, acc.y); acc.z = fmaf(m.z, km.z, acc.z); acc.w = fmaf(m.w, km.w, acc.w);
                        acc.x = fmaf(d.x, kd.x, acc.x); acc.y = fmaf(d.y, kd.y, acc.y); acc.z = fmaf(d.z, kd.z, acc.z); acc.w = fmaf(d.w, kd.w, acc.w);
                    }
                    *reinterpret_cast<float4*>(out + ((b * H + yy + o) * (int64_t)W + xx) * C + c4 * 4) = act4(acc, act);
                }
            }
#pragma unroll
            for (int r = 0; r < 4; ++r) { c0[r] = c1[r]; c1[r] = c2[r]; }
        }
    }
}

// DeepLabV3+ decoder (smp DeepLabV3PlusDecoder.forward: up(aspp) -> cat with the 48-channel skip -> SeparableConv2d):
// the depthwise 3x3 of block2 applied to cat(UpsamplingBilinear2d(x4, align_corners=True)(a), hi) WITHOUT
// materialising the upsampled map or the concatenation (1.07 + 1.27 GB per batch at 1024x2048).  Same strip
// scheme as above; a column of the first Ca channels is sampled from the stride-16 map with torch's
// upsample_bilinear2d source indices and weights (source index = dst * (in-1)/(out-1)).
template <int SX>
__global__ __launch_bounds__(kThreads)
void dwconv3x3_upcat_strip_kernel(const float* __restrict__ a, int h, int w, int Ca, const float* __restrict__ hi, int Ch,
                                  int64_t batch, int H, int W, float ry, float rx, const float* __restrict__ w9,
                                  float* __restrict__ out, int c4_lo)
{
    const int C = Ca + Ch, c4n = C / 4, ca4 = Ca / 4;
    const int nsx = (W + SX - 1) / SX;
    const int c4w = c4n - c4_lo;                                  // this launch covers channel quads [c4_lo, c4n)
    const int64_t total = batch * H * nsx * c4w;
    for (int64_t i = (int64_t)blockIdx.x * kThreads + threadIdx.x; i < total; i += (int64_t)gridDim.x * kThreads) {
        const int c4 = c4_lo + (int)(i % c4w);
        int64_t t = i / c4w;
        const int xs = (int)(t % nsx); t /= nsx;
        const int yy = (int)(t % H);
        const int64_t b = t / H;
        float4 k[9];
#pragma unroll
        for (int j = 0; j < 9; ++j) k[j] = *reinterpret_cast<const float4*>(w9 + j * C + c4 * 4);
        const float4 zero = make_float4(0.f, 0.f, 0.f, 0.f);
        const bool up = c4 < ca4;
        const float* ab = a + b * (int64_t)h * w * Ca + c4 * 4;
        const float* hb = hi + b * (int64_t)H * W * Ch + (c4 - ca4) * 4;
        // per-row sampling constants of the three tap rows
        int r0[3], r1[3]; float l1[3], l0[3]; bool rok[3];
#pragma unroll
        for (int ky = 0; ky < 3; ++ky) {
            const int sy = yy + ky - 1;
            rok[ky] = sy >= 0 && sy < H;
            const float f = ry * (float)(rok[ky] ? sy : 0);
            const int i0 = (int)f;
            r0[ky] = i0; r1[ky] = i0 + (i0 < h - 1 ? 1 : 0);
            l1[ky] = f - (float)i0; l0[ky] = 1.0f - l1[ky];
        }
        // the four corner columns of the current source cell, kept across output columns: at x4 upsampling the
        // source column index changes every ~4 output columns, so most columns reuse all 12 loads
        float4 vj0[3], vj1[3];                              // per tap row: the two source columns, already blended vertically
        int cur_j0 = -1;
        auto col = [&](int sx, float4* v) {
            const bool cok = sx >= 0 && sx < W;
            if (up) {
                const float f = rx * (float)(cok ? sx : 0);
                const int j0 = (int)f, j1 = j0 + (j0 < w - 1 ? 1 : 0);
                const float m1 = f - (float)j0, m0 = 1.0f - m1;
                if (cok && j0 != cur_j0) {
                    cur_j0 = j0;
#pragma unroll
                    for (int ky = 0; ky < 3; ++ky) {
                        const float4 v00 = *reinterpret_cast<const float4*>(ab + ((int64_t)r0[ky] * w + j0) * Ca);
                        const float4 v01 = *reinterpret_cast<const float4*>(ab + ((int64_t)r0[ky] * w + j1) * Ca);
                        const float4 v10 = *reinterpret_cast<const float4*>(ab + ((int64_t)r1[ky] * w + j0) * Ca);
                        const float4 v11 = *reinterpret_cast<const float4*>(ab + ((int64_t)r1[ky] * w + j1) * Ca);
                        // vertical blend once per source cell (torch blends horizontally first: same value up to the
                        // last bit of a convex combination, far inside the 1e-4 budget)
                        vj0[ky] = make_float4(l0[ky] * v00.x + l1[ky] * v10.x, l0[ky] * v00.y + l1[ky] * v10.y,
                                              l0[ky] * v00.z + l1[ky] * v10.z, l0[ky] * v00.w + l1[ky] * v10.w);
                        vj1[ky] = make_float4(l0[ky] * v01.x + l1[ky] * v11.x, l0[ky] * v01.y + l1[ky] * v11.y,
                                              l0[ky] * v01.z + l1[ky] * v11.z, l0[ky] * v01.w + l1[ky] * v11.w);
                    }
                }
#pragma unroll
                for (int ky = 0; ky < 3; ++ky) {
                    if (cok && rok[ky]) {
                        const float4 a0 = vj0[ky], a1 = vj1[ky];
                        v[ky] = make_float4(m0 * a0.x + m1 * a1.x, m0 * a0.y + m1 * a1.y, m0 * a0.z + m1 * a1.z, m0 * a0.w + m1 * a1.w);
                    } else {
                        v[ky] = zero;
                    }
                }
            } else {
#pragma unroll
                for (int ky = 0; ky < 3; ++ky) {
                    const int sy = yy + ky - 1;
                    v[ky] = (cok && rok[ky]) ? *reinterpret_cast<const float4*>(hb + ((int64_t)sy * W + sx) * Ch) : zero;
                }
            }
        };
        const int x0 = xs * SX;
        float4 c0[3], c1[3], c2[3];
        col(x0 - 1, c0); col(x0, c1);
#pragma unroll
        for (int u = 0; u < SX; ++u) {
            const int xx = x0 + u;
            col(xx + 1, c2);
            if (xx < W) {
                float4 acc = zero;
#pragma unroll
                for (int ky = 0; ky < 3; ++ky) {
                    const float4 p = c0[ky], m = c1[ky], d = c2[ky];
                    const float4 ka = k[ky * 3], km = k[ky * 3 + 1], kd = k[ky * 3 + 2];
                    acc.x = fmaf(p.x, ka.x, acc.x); acc.y = fmaf(p.y, ka.y, acc.y); acc.z = fmaf(p.z, ka.z, acc.z); acc.w = fmaf(p.w, ka.w, acc.w);
                    acc.x = fmaf(m.x, km.x, acc.x); acc.y = fmaf(m.y, km.y, acc.y); acc.z = fmaf(m.z, km.z, acc.z); acc.w = fmaf(m.w, km.w, acc.w);
                    acc.x = fmaf(d.x, kd.x, acc.x); acc.y = fmaf(d.y, kd.y, acc.y); acc.z = fmaf(d.z, kd.z, acc.z); acc.w = fmaf(d.w, kd.w, acc.w);
                }
                *reinterpret_cast<float4*>(out + ((b * H + yy) * (int64_t)W + xx) * C + c4 * 4) = acc;
            }
#pragma unroll
            for (int ky = 0; ky < 3; ++ky) { c0[ky] = c1[ky]; c1[ky] = c2[ky]; }
        }
    }
}

// The upsampled channels of the same operation with the source cells staged in LDS.  Block = one output row x 32 columns
// x 64 channel quads (4 strips of 8 columns).  The <= n_src source columns the 34 tap columns touch are blended
// vertically ONCE per (source column, tap row) and stored as [source column][tap row][quad] float4; a tap column is then
// two 16-byte LDS reads per tap row at a computed address (lane stride 16 B: conflict-free) and one horizontal blend.
// (The strip kernel above reloads its 12 corner values behind a data-dependent branch inside the serial column loop —
// nothing can be prefetched across it: 18 % of the HBM roof.)  Same expressions, same order: bit-identical results.
// (Measured and dropped in round 3: the staging loop in batches of five cells per thread, all ten loads in flight before the
// first blend — 118 registers, 0.646 -> 0.689 ms: the staging round trips are not what bounds a block.)
constexpr int kUpTile = 32, kUpQuads = 64;
template <int SX>
__global__ __launch_bounds__(kThreads)
void dwconv3x3_upcat_lds_kernel(const float* __restrict__ a, int h, int w, int Ca, int Ch, int H, int W, float ry, float rx,
                                const float* __restrict__ w9, float* __restrict__ out, int n_src)
{
    static_assert(SX * (kThreads / kUpQuads) == kUpTile, "4 strips of 8 columns");
    extern __shared__ float4 s_cell[];                             // [n_src][3][kUpQuads]
    const int C = Ca + Ch, qg = Ca / 4 / kUpQuads;
    const int b = blockIdx.z / qg, g = blockIdx.z - b * qg;
    const int yy = blockIdx.y, x0 = blockIdx.x * kUpTile;
    const int q = threadIdx.x % kUpQuads, strip = threadIdx.x / kUpQuads;
    const int c = (g * kUpQuads + q) * 4;
    int r0[3], r1[3]; float l1[3], l0[3]; bool rok[3];
#pragma unroll
    for (int ky = 0; ky < 3; ++ky) {
        const int sy = yy + ky - 1;
        rok[ky] = sy >= 0 && sy < H;
        const float f = ry * (float)(rok[ky] ? sy : 0);
        const int i0 = (int)f;
        r0[ky] = i0; r1[ky] = i0 + (i0 < h - 1 ? 1 : 0);
        l1[ky] = f - (float)i0; l0[ky] = 1.0f - l1[ky];
    }
    const int jbase = (int)(rx * (float)(x0 > 0 ? x0 - 1 : 0));
    const float4 zero = make_float4(0.f, 0.f, 0.f, 0.f);
    for (int i = threadIdx.x; i < n_src * 3 * kUpQuads; i += kThreads) {
        const int iq = i % kUpQuads, t = i / kUpQuads;
        const int ky = t % 3;
        int j = jbase + t / 3;
        if (j > w - 1) j = w - 1;
        const float* p = a + (int64_t)b * h * w * Ca + (g * kUpQuads + iq) * 4;
        const int ra = ky == 0 ? r0[0] : (ky == 1 ? r0[1] : r0[2]), rb = ky == 0 ? r1[0] : (ky == 1 ? r1[1] : r1[2]);
        const float la = ky == 0 ? l0[0] : (ky == 1 ? l0[1] : l0[2]), lb = ky == 0 ? l1[0] : (ky == 1 ? l1[1] : l1[2]);
        const float4 v0 = *reinterpret_cast<const float4*>(p + ((int64_t)ra * w + j) * Ca);
        const float4 v1 = *reinterpret_cast<const float4*>(p + ((int64_t)rb * w + j) * Ca);
        s_cell[i] = make_float4(la * v0.x + lb * v1.x, la * v0.y + lb * v1.y, la * v0.z + lb * v1.z, la * v0.w + lb * v1.w);
    }
    float4 k[9];
#pragma unroll
    for (int j = 0; j < 9; ++j) k[j] = *reinterpret_cast<const float4*>(w9 + j * C + c);
    __syncthreads();
    auto col = [&](int sx, float4* v) {
        const bool cok = sx >= 0 && sx < W;
        const float f = rx * (float)(cok ? sx : 0);
        const int j0 = (int)f, j1 = j0 + (j0 < w - 1 ? 1 : 0);
        const float m1 = f - (float)j0, m0 = 1.0f - m1;
        const float4* c0p = s_cell + (j0 - jbase) * 3 * kUpQuads + q;
        const float4* c1p = s_cell + (j1 - jbase) * 3 * kUpQuads + q;
#pragma unroll
        for (int ky = 0; ky < 3; ++ky) {
            if (cok && rok[ky]) {
                const float4 a0 = c0p[ky * kUpQuads], a1 = c1p[ky * kUpQuads];
                v[ky] = make_float4(m0 * a0.x + m1 * a1.x, m0 * a0.y + m1 * a1.y, m0 * a0.z + m1 * a1.z, m0 * a0.w + m1 * a1.w);
            } else {
                v[ky] = zero;
            }
        }
    };
    const int xs = x0 + strip * SX;
    float4 c0[3], c1[3], c2[3];
    col(xs - 1, c0); col(xs, c1);
#pragma unroll
    for (int u = 0; u < SX; ++u) {
        const int xx = xs + u;
        col(xx + 1, c2);
        if (xx < W) {
            float4 acc = zero;
#pragma unroll
            for (int ky = 0; ky < 3; ++ky) {
                const float4 p = c0[ky], m = c1[ky], d = c2[ky];
                const float4 ka = k[ky * 3], km = k[ky * 3 + 1], kd = k[ky * 3 + 2];
                acc.x = fmaf(p.x, ka.x, acc.x); acc.y = fmaf(p.y, ka.y, acc.y); acc.z = fmaf(p.z, ka.z, acc.z); acc.w = fmaf(p.w, ka.w, acc.w);
                acc.x = fmaf(m.x, km.x, acc.x); acc.y = fmaf(m.y, km.y, acc.y); acc.z = fmaf(m.z, km.z, acc.z); acc.w = fmaf(m.w, km.w, acc.w);
                acc.x = fmaf(d.x, kd.x, acc.x); acc.y = fmaf(d.y, kd.y, acc.y); acc.z = fmaf(d.z, kd.z, acc.z); acc.w = fmaf(d.w, kd.w, acc.w);
            }
            *reinterpret_cast<float4*>(out + (((int64_t)b * H + yy) * (int64_t)W + xx) * C + c) = acc;
        }
#pragma unroll
        for (int ky = 0; ky < 3; ++ky) { c0[ky] = c1[ky]; c1[ky] = c2[ky]; }
    }
}

// nn.MaxPool2d(3, stride 2, padding 1) on an NHWC float32 tensor (the ResNet stem; torch's max_pool2d_with_indices also writes
// an int64 index per output — 537 MB per batch of 8 at 1024 x 2048 — that nobody reads).  A lane owns 2 adjacent output pixels
// of one channel quad: 5 columns x 3 rows = 15 loads for 2 outputs.  Out-of-image taps do not take part (-inf padding).
// max that PROPAGATES NaN like nn.MaxPool2d (fmaxf drops it): a NaN from the stem must reach the logits on this path as on torch's
__device__ __forceinline__ float nanmax(float m, float v) { return (v > m || v != v) ? v : m; }

// EPI: relu(max + shift[c]) — the ResNet stem's folded BatchNorm shift and ReLU, moved behind the pooling (both monotone per channel:
// bit-identical to shift -> ReLU -> pool), in the pooling kernel's store instead of a pass of their own over the pooled map
template <bool EPI>
__global__ __launch_bounds__(kThreads)
void maxpool3x3s2_nhwc_kernel(const float* __restrict__ x, int64_t batch, int H, int W, int C, int Ho, int Wo, const float* __restrict__ shift,
                              float* __restrict__ out)
{
    const int c4n = C / 4, wp = (Wo + 1) / 2;
    const int64_t total = batch * Ho * wp * c4n;
    for (int64_t i = (int64_t)blockIdx.x * kThreads + threadIdx.x; i < total; i += (int64_t)gridDim.x * kThreads) {
        const int c4 = (int)(i % c4n);
        int64_t t = i / c4n;
        const int xp = (int)(t % wp); t /= wp;
        const int oy = (int)(t % Ho);
        const int64_t b = t / Ho;
        const float* xb = x + b * (int64_t)H * W * C + c4 * 4;
        const float ninf = -__builtin_inff();
        float4 col[5];
#pragma unroll
        for (int q = 0; q < 5; ++q) {
            const int ix = xp * 4 - 1 + q;                         // input columns 2 ox - 1 .. 2 ox + 1 of ox = 2 xp, 2 xp + 1
            float4 m = make_float4(ninf, ninf, ninf, ninf);
#pragma unroll
            for (int r = 0; r < 3; ++r) {
                const int iy = oy * 2 - 1 + r;
                if (ix >= 0 && ix < W && iy >= 0 && iy < H) {
                    const float4 v = *reinterpret_cast<const float4*>(xb + ((int64_t)iy * W + ix) * C);
                    m.x = nanmax(m.x, v.x); m.y = nanmax(m.y, v.y); m.z = nanmax(m.z, v.z); m.w = nanmax(m.w, v.w);
                }
            }
            col[q] = m;
        }
#pragma unroll
        for (int o = 0; o < 2; ++o) {
            const int ox = xp * 2 + o;
            if (ox >= Wo) break;
            const float4 a = col[2 * o], bq = col[2 * o + 1], c = col[2 * o + 2];
            float4 r = make_float4(nanmax(nanmax(a.x, bq.x), c.x), nanmax(nanmax(a.y, bq.y), c.y), nanmax(nanmax(a.z, bq.z), c.z), nanmax(nanmax(a.w, bq.w), c.w));
            if (EPI) {
                const float4 sh = *reinterpret_cast<const float4*>(shift + c4 * 4);
                r.x = r.x + sh.x; r.y = r.y + sh.y; r.z = r.z + sh.z; r.w = r.w + sh.w;          // bias_act_nhwc's operations, in its order
                r.x = r.x > 0.f ? r.x : 0.f; r.y = r.y > 0.f ? r.y : 0.f; r.z = r.z > 0.f ? r.z : 0.f; r.w = r.w > 0.f ? r.w : 0.f;
            }
            *reinterpret_cast<float4*>(out + ((b * Ho + oy) * (int64_t)Wo + ox) * C + c4 * 4) = r;
        }
    }
}

// Bilinear upsampling of [planes, h, w] float32 maps to [planes, H, W] with torch's upsample_bilinear2d arithmetic (source index,
// weights and the order h0 * (w0 * v00 + w1 * v01) + h1 * (w0 * v10 + w1 * v11)), either corner convention.  DeepLabV3+'s
// segmentation head ends in UpsamplingBilinear2d(x4) on the 19 logit planes (1.27 GB written per batch of 8 at 1024x2048):
// 4 output pixels per lane, one 16-byte store, the <= 2 x 4 source values from L2 (the low-resolution maps are 64x smaller).
__global__ __launch_bounds__(kThreads)
void upsample_bilinear_kernel(const float* __restrict__ low, int h, int w, int H, int W, float sy, float sx, int align,
                              float* __restrict__ out, int64_t planes)
{
    const int wq = (W + 3) / 4;
    const int64_t total = planes * H * wq;
    for (int64_t i = (int64_t)blockIdx.x * kThreads + threadIdx.x; i < total; i += (int64_t)gridDim.x * kThreads) {
        const int xq = (int)(i % wq);
        int64_t t = i / wq;
        const int y = (int)(t % H);
        const int64_t pl = t / H;
        float fy = align ? sy * (float)y : sy * ((float)y + 0.5f) - 0.5f;
        fy = fy < 0.f ? 0.f : fy;
        const int y0 = (int)fy, y1 = y0 + (y0 < h - 1 ? 1 : 0);
        const float ly1 = fy - (float)y0, ly0 = 1.0f - ly1;
        const float* r0 = low + (pl * h + y0) * (int64_t)w;
        const float* r1 = low + (pl * h + y1) * (int64_t)w;
        float v[4];
        if (sx * 3.0f < 1.0f) {
            // upsampling by more than 3: the four pixels' source columns lie in [a, a + 2] — six loads instead of sixteen
            // (the sixteen-load form is bound by the vector-memory pipe: 0.51 ms for 1.27 GB, torch's kernel the same)
            int xa = xq * 4; xa = xa < W ? xa : W - 1;
            float fa = align ? sx * (float)xa : sx * ((float)xa + 0.5f) - 0.5f;
            fa = fa < 0.f ? 0.f : fa;
            const int a = (int)fa;
            const int c1 = a + 1 < w ? a + 1 : w - 1, c2 = a + 2 < w ? a + 2 : w - 1;
            const float t0 = r0[a], t1 = r0[c1], t2 = r0[c2], b0 = r1[a], b1 = r1[c1], b2 = r1[c2];
#pragma unroll
            for (int k = 0; k < 4; ++k) {
                int x = xq * 4 + k; x = x < W ? x : W - 1;
                float fx = align ? sx * (float)x : sx * ((float)x + 0.5f) - 0.5f;
                fx = fx < 0.f ? 0.f : fx;
                const int x0 = (int)fx;
                const float lx1 = fx - (float)x0, lx0 = 1.0f - lx1;
                const bool first = x0 == a;                        // else x0 == a + 1; x1 = x0 + 1 clamped = the next staged column
                const float v00 = first ? t0 : t1, v01 = first ? t1 : t2, v10 = first ? b0 : b1, v11 = first ? b1 : b2;
                v[k] = ly0 * (lx0 * v00 + lx1 * v01) + ly1 * (lx0 * v10 + lx1 * v11);
            }
        } else {
#pragma unroll
            for (int k = 0; k < 4; ++k) {
                int x = xq * 4 + k; x = x < W ? x : W - 1;
                float fx = align ? sx * (float)x : sx * ((float)x + 0.5f) - 0.5f;
                fx = fx < 0.f ? 0.f : fx;
                const int x0 = (int)fx, x1 = x0 + (x0 < w - 1 ? 1 : 0);
                const float lx1 = fx - (float)x0, lx0 = 1.0f - lx1;
                v[k] = ly0 * (lx0 * r0[x0] + lx1 * r0[x1]) + ly1 * (lx0 * r1[x0] + lx1 * r1[x1]);
            }
        }
        float* o = out + (pl * H + y) * (int64_t)W + xq * 4;
        if ((W & 3) == 0) *reinterpret_cast<float4*>(o) = make_float4(v[0], v[1], v[2], v[3]);
        else for (int k = 0; k < 4 && xq * 4 + k < W; ++k) o[k] = v[k];
    }
}

// Upsampling by more than 3 in both directions (W % 4 == 0): a lane owns a 4 x 4 block of output pixels, whose source values lie
// in 3 x 3 cells — nine loads and four 16-byte stores per lane (the row-at-a-time kernel above runs one dependent
// load -> blend -> store chain per 16 bytes and is latency-bound at 2.9 TB/s).  Same expressions: same values.
__global__ __launch_bounds__(kThreads)
void upsample_bilinear_4x4_kernel(const float* __restrict__ low, int h, int w, int H, int W, float sy, float sx, int align,
                                  float* __restrict__ out, int64_t planes, int channels, int64_t ls_b, int64_t ls_c, int64_t ls_y, int64_t ls_x)
{
    const int wq = W / 4, hq = (H + 3) / 4;
    const int64_t total = planes * hq * wq;
    for (int64_t i = (int64_t)blockIdx.x * kThreads + threadIdx.x; i < total; i += (int64_t)gridDim.x * kThreads) {
        const int xq = (int)(i % wq);
        int64_t t = i / wq;
        const int yq = (int)(t % hq);
        const int64_t pl = t / hq;
        auto src = [&](int d, float sc) { float f = align ? sc * (float)d : sc * ((float)d + 0.5f) - 0.5f; return f < 0.f ? 0.f : f; };
        const int ya = (int)src(yq * 4, sy), xa = (int)src(xq * 4, sx);
        float c[3][3];
        // the low-resolution map through its strides (floats): planar NCHW, or the NHWC rows a GEMM wrote — it is small and stays in L2
        const int64_t pb = pl / channels;
        const float* plane = low + pb * ls_b + (pl - pb * channels) * ls_c;
#pragma unroll
        for (int r = 0; r < 3; ++r) {
            const int yy = ya + r < h ? ya + r : h - 1;
            const float* row = plane + yy * ls_y;
#pragma unroll
            for (int q = 0; q < 3; ++q) c[r][q] = row[(xa + q < w ? xa + q : w - 1) * ls_x];
        }
        float lx0[4], lx1[4]; bool fx_first[4];
#pragma unroll
        for (int k = 0; k < 4; ++k) {
            const float fx = src(xq * 4 + k, sx);
            const int x0 = (int)fx;
            lx1[k] = fx - (float)x0; lx0[k] = 1.0f - lx1[k]; fx_first[k] = x0 == xa;
        }
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            const int y = yq * 4 + j;
            if (y >= H) break;
            const float fy = src(y, sy);
            const int y0 = (int)fy;
            const float ly1 = fy - (float)y0, ly0 = 1.0f - ly1;
            const bool fy_first = y0 == ya;
            float v[4];
#pragma unroll
            for (int k = 0; k < 4; ++k) {
                const float t00 = fy_first ? (fx_first[k] ? c[0][0] : c[0][1]) : (fx_first[k] ? c[1][0] : c[1][1]);
                const float t01 = fy_first ? (fx_first[k] ? c[0][1] : c[0][2]) : (fx_first[k] ? c[1][1] : c[1][2]);
                const float t10 = fy_first ? (fx_first[k] ? c[1][0] : c[1][1]) : (fx_first[k] ? c[2][0] : c[2][1]);
                const float t11 = fy_first ? (fx_first[k] ? c[1][1] : c[1][2]) : (fx_first[k] ? c[2][1] : c[2][2]);
                v[k] = ly0 * (lx0[k] * t00 + lx1[k] * t01) + ly1 * (lx0[k] * t10 + lx1[k] * t11);
            }
            *reinterpret_cast<float4*>(out + (pl * H + y) * (int64_t)W + xq * 4) = make_float4(v[0], v[1], v[2], v[3]);
        }
    }
}

// Depth tail of EnsembleModel.forward (PKG/models/model.py:368-371 + :471-478): the DeepLab depth map is predicted at
// stride 16, upsampled bilinearly (align_corners=False) to the input size and combined with the SegFormer map as
// w0*d1 + w1*d2 (or their mean).  One pass: reads d1 and the 64x-smaller low-resolution map, writes both outputs.
__global__ __launch_bounds__(kThreads)
void depth_up_combine_kernel(const float* __restrict__ d1, const float* __restrict__ d2_low, int h, int w, int H, int W,
                             float sy, float sx, const float* __restrict__ weights, float* __restrict__ d2_full,
                             float* __restrict__ d_out)
{
    const int64_t hw = (int64_t)H * W;
    const float* lo = d2_low + (int64_t)blockIdx.y * h * w;
    const float w0 = weights ? weights[0] : 0.f, w1 = weights ? weights[1] : 0.f;
    for (int64_t p = (int64_t)blockIdx.x * kThreads + threadIdx.x; p < hw; p += (int64_t)gridDim.x * kThreads) {
        const int y = (int)(p / W), x = (int)(p - (int64_t)y * W);
        // torch area_pixel_compute_source_index(scale, dst, align_corners=false, cubic=false)
        float fy = sy * ((float)y + 0.5f) - 0.5f; fy = fy < 0.f ? 0.f : fy;
        float fx = sx * ((float)x + 0.5f) - 0.5f; fx = fx < 0.f ? 0.f : fx;
        const int y0 = (int)fy, x0 = (int)fx;
        const int y1 = y0 + (y0 < h - 1 ? 1 : 0), x1 = x0 + (x0 < w - 1 ? 1 : 0);
        const float ly1 = fy - (float)y0, ly0 = 1.0f - ly1, lx1 = fx - (float)x0, lx0 = 1.0f - lx1;
        const float v = ly0 * (lx0 * lo[y0 * w + x0] + lx1 * lo[y0 * w + x1]) + ly1 * (lx0 * lo[y1 * w + x0] + lx1 * lo[y1 * w + x1]);
        const int64_t o = (int64_t)blockIdx.y * hw + p;
        d2_full[o] = v;
        const float a = d1[o];
        d_out[o] = weights ? (w0 * a + w1 * v) : ((a + v) / 2.0f);
    }
}

__global__ __launch_bounds__(kThreads)
void bias_act_nhwc_kernel(float* __restrict__ x, int64_t n_pixels, int C, const float* __restrict__ bias,
                          const float* __restrict__ residual, int act)
{
    const int c4n = C / 4;
    const int64_t total = n_pixels * c4n;
    for (int64_t i = (int64_t)blockIdx.x * kThreads + threadIdx.x; i < total; i += (int64_t)gridDim.x * kThreads) {
        const int c4 = (int)(i % c4n);
        float4 v = reinterpret_cast<float4*>(x)[i];
        if (bias) {
            const float4 b = *reinterpret_cast<const float4*>(bias + c4 * 4);
            v.x += b.x; v.y += b.y; v.z += b.z; v.w += b.w;
        }
        if (residual) {
            const float4 r = reinterpret_cast<const float4*>(residual)[i];
            v.x += r.x; v.y += r.y; v.z += r.z; v.w += r.w;
        }
        reinterpret_cast<float4*>(x)[i] = act4(v, act);
    }
}

// LayerNorm over the last dimension for token matrices [N, C] with small C (MiT: 32..512).  A row is
// owned by L = 8..64 lanes of one wave (float4 chunks, strided by L), statistics are two shuffle
// reductions inside those L lanes, the row never leaves registers: 8 B/element of HBM traffic.
// (torch's kernel gives one row to a whole block: 0.98 ms for 1M x 32 against 0.054 ms of traffic.)
template <int L>
__global__ __launch_bounds__(kThreads)
void layernorm_rows_kernel(const float* __restrict__ x, int64_t n_rows, int C, const float* __restrict__ gamma,
                           const float* __restrict__ beta, float eps, float* __restrict__ out)
{
    constexpr int MAXCH = 4;                               // chunks of 4 floats per lane -> C <= 16 * L
    const int sub = threadIdx.x % L;
    const int64_t rows_per_block = kThreads / L;
    const int nchunk = C / 4;
    const float inv_c = 1.0f / (float)C;
    for (int64_t row = (int64_t)blockIdx.x * rows_per_block + threadIdx.x / L; row < n_rows; row += (int64_t)gridDim.x * rows_per_block) {
        const float* xr = x + row * C;
        float4 v[MAXCH];
        float sum = 0.f;
#pragma unroll
        for (int j = 0; j < MAXCH; ++j) {
            const int ch = sub + j * L;
            if (ch < nchunk) {
                v[j] = *reinterpret_cast<const float4*>(xr + ch * 4);
                sum += (v[j].x + v[j].y) + (v[j].z + v[j].w);
            }
        }
#pragma unroll
        for (int o = L / 2; o > 0; o >>= 1) sum += __shfl_xor(sum, o, L);
        const float mean = sum * inv_c;
        float sq = 0.f;
#pragma unroll
        for (int j = 0; j < MAXCH; ++j) {
            const int ch = sub + j * L;
            if (ch < nchunk) {
                const float a = v[j].x - mean, b = v[j].y - mean, c = v[j].z - mean, d = v[j].w - mean;
                sq += (a * a + b * b) + (c * c + d * d);
            }
        }
#pragma unroll
        for (int o = L / 2; o > 0; o >>= 1) sq += __shfl_xor(sq, o, L);
        const float rstd = rsqrtf(sq * inv_c + eps);
        float* orow = out + row * C;
#pragma unroll
        for (int j = 0; j < MAXCH; ++j) {
            const int ch = sub + j * L;
            if (ch < nchunk) {
                const float4 g = *reinterpret_cast<const float4*>(gamma + ch * 4);
                const float4 bb = *reinterpret_cast<const float4*>(beta + ch * 4);
                float4 r;
                r.x = (v[j].x - mean) * rstd * g.x + bb.x; r.y = (v[j].y - mean) * rstd * g.y + bb.y;
                r.z = (v[j].z - mean) * rstd * g.z + bb.z; r.w = (v[j].w - mean) * rstd * g.w + bb.w;
                *reinterpret_cast<float4*>(orow + ch * 4) = r;
            }
        }
    }
}

}  // namespace

namespace {
// ---- im2col for the strided / patch convolutions (ResNet layer2/3 first blocks, MiT patch embeddings and the
// sequence-reduction convolutions): cols[m][(ky*kw + kx)*C + c] = x[b][oy*s - p + ky*d][ox*s - p + kx*d][c] (0 outside),
// m = (b*Ho + oy)*Wo + ox, row length k_padded >= kh*kw*C (tail zero-filled).  The product with w[N][kh][kw][C] is then
// one GEMM with the convolution's bias / BatchNorm shift / activation in its epilogue (awseg_gemm_split_bias_act or
// awseg_gemm_bias_act): deterministic accumulation order, where MIOpen's pick for these shapes is a split-K igemm that
// sums with atomics (run-to-run different results, tools/check_op_determinism.py).  One float4 per lane per item,
// items ordered (m, tap, c/4) so a wave writes 1 KiB runs of a row.
__global__ __launch_bounds__(kThreads)
void im2col_nhwc_kernel(const float* __restrict__ x, int64_t rows, int H, int W, int C, int Ho, int Wo, int kh, int kw,
                        int stride, int pad, int dil, int kpad, float* __restrict__ cols)
{
    const int q_row = kpad / 4;                       // float4 per output row
    const int c4n = C / 4;
    const int kq = kh * kw * c4n;                     // float4 that carry data
    const int64_t total = rows * q_row;
    for (int64_t i = (int64_t)blockIdx.x * kThreads + threadIdx.x; i < total; i += (int64_t)gridDim.x * kThreads) {
        const int q = (int)(i % q_row);
        const int64_t m = i / q_row;
        float4 v = make_float4(0.f, 0.f, 0.f, 0.f);
        if (q < kq) {
            const int tap = q / c4n, c4 = q - tap * c4n;
            const int ky = tap / kw, kx = tap - ky * kw;
            const int ox = (int)(m % Wo);
            const int64_t t = m / Wo;
            const int oy = (int)(t % Ho);
            const int64_t b = t / Ho;
            const int iy = oy * stride - pad + ky * dil, ix = ox * stride - pad + kx * dil;
            if (iy >= 0 && iy < H && ix >= 0 && ix < W)
                v = *reinterpret_cast<const float4*>(x + ((b * H + iy) * (int64_t)W + ix) * C + c4 * 4);
        }
        *reinterpret_cast<float4*>(cols + m * kpad + q * 4) = v;
    }
}
}  // namespace

AWSEG_API int awseg_im2col_nhwc(const float* x, int64_t batch, int height, int width, int channels, int kernel_h, int kernel_w,
                                int stride, int pad, int dilation, int k_padded, float* cols, awseg_stream_t stream)
{
    if (batch == 0) return 0;
    if (!x || !cols || batch < 0 || height < 1 || width < 1 || channels < 4 || kernel_h < 1 || kernel_w < 1 || stride < 1 ||
        pad < 0 || dilation < 1) return AWSEG_EINVAL;
    if (channels % 4 || k_padded % 4 || k_padded < kernel_h * kernel_w * channels) return AWSEG_ERANGE;
    if (((uintptr_t)x & 15) || ((uintptr_t)cols & 15)) return AWSEG_EALIGN;
    const int ho = (height + 2 * pad - dilation * (kernel_h - 1) - 1) / stride + 1;
    const int wo = (width + 2 * pad - dilation * (kernel_w - 1) - 1) / stride + 1;
    if (ho < 1 || wo < 1) return AWSEG_ERANGE;
    const int64_t rows = batch * ho * wo;
    hipLaunchKernelGGL(im2col_nhwc_kernel, dim3(awseg_grid_1d(rows * (k_padded / 4), kThreads)), dim3(kThreads), 0, awseg_s(stream),
                       x, rows, height, width, channels, ho, wo, kernel_h, kernel_w, stride, pad, dilation, k_padded, cols);
    AWSEG_LAUNCH_CHECK();
    return 0;
}

AWSEG_API int awseg_dwconv3x3_nhwc(const float* x, int64_t batch, int height, int width, int channels, int dilation,
                                   const float* w9, const float* bias, int act, float* out, awseg_stream_t stream)
{
    if (!x || !w9 || !out || batch < 1 || height < 1 || width < 1 || channels < 4 || (channels & 3) || dilation < 1) return AWSEG_EINVAL;
    if (act < 0 || act > 2 || x == out) return AWSEG_EINVAL;
    static const bool gelu_exact = getenv("AWSEG_GELU_EXACT") && atoi(getenv("AWSEG_GELU_EXACT")) != 0;
    if (act == 2 && gelu_exact) act = 3;                          // library erff instead of the 14-instruction erf (act1)
    if (((uintptr_t)x & 15) || ((uintptr_t)w9 & 15) || ((uintptr_t)out & 15) || (bias && ((uintptr_t)bias & 15))) return AWSEG_EALIGN;
    static const bool one_row = getenv("AWSEG_DW_ONE_ROW") != nullptr;
    if (dilation == 1 && width >= 8 && height >= 2 && !one_row) {
        constexpr int SX = 8;
        const int64_t total = batch * ((height + 1) / 2) * ((width + SX - 1) / SX) * (channels / 4);
        hipLaunchKernelGGL((dwconv3x3_nhwc_strip2_kernel<SX>), dim3(awseg_grid_1d(total, kThreads)), dim3(kThreads), 0, awseg_s(stream),
                           x, batch, height, width, channels, w9, bias, act, out);
        AWSEG_LAUNCH_CHECK();
        return 0;
    }
    if (dilation == 1 && width >= 8) {
        constexpr int SX = 8;
        const int64_t total = batch * height * ((width + SX - 1) / SX) * (channels / 4);
        hipLaunchKernelGGL((dwconv3x3_nhwc_strip_kernel<SX>), dim3(awseg_grid_1d(total, kThreads)), dim3(kThreads), 0, awseg_s(stream),
                           x, batch, height, width, channels, w9, bias, act, out);
        AWSEG_LAUNCH_CHECK();
        return 0;
    }
    const int64_t total = batch * height * width * (channels / 4);
    hipLaunchKernelGGL(dwconv3x3_nhwc_kernel, dim3(awseg_grid_1d(total, kThreads)), dim3(kThreads), 0, awseg_s(stream), x, batch,
                       height, width, channels, dilation, w9, bias, act, out);
    AWSEG_LAUNCH_CHECK();
    return 0;
}

AWSEG_API int awseg_bias_act_nhwc(float* x, int64_t n_pixels, int channels, const float* bias, const float* residual,
                                  int act, awseg_stream_t stream)
{
    if (!x || n_pixels < 1 || channels < 4 || (channels & 3) || act < 0 || act > 2) return AWSEG_EINVAL;
    static const bool gelu_exact = getenv("AWSEG_GELU_EXACT") && atoi(getenv("AWSEG_GELU_EXACT")) != 0;
    if (act == 2 && gelu_exact) act = 3;
    if (((uintptr_t)x & 15) || (bias && ((uintptr_t)bias & 15)) || (residual && ((uintptr_t)residual & 15))) return AWSEG_EALIGN;
    const int64_t total = n_pixels * (channels / 4);
    hipLaunchKernelGGL(bias_act_nhwc_kernel, dim3(awseg_grid_1d(total, kThreads)), dim3(kThreads), 0, awseg_s(stream), x, n_pixels,
                       channels, bias, residual, act);
    AWSEG_LAUNCH_CHECK();
    return 0;
}

AWSEG_API int awseg_layernorm_rows(const float* x, int64_t n_rows, int channels, const float* gamma, const float* beta,
                                   float eps, float* out, awseg_stream_t stream)
{
    if (!x || !gamma || !beta || !out || n_rows < 1 || channels < 4 || (channels & 3) || channels > 1024) return AWSEG_EINVAL;
    if (((uintptr_t)x & 15) || ((uintptr_t)out & 15) || ((uintptr_t)gamma & 15) || ((uintptr_t)beta & 15)) return AWSEG_EALIGN;
    const int nchunk = channels / 4;
    int L = 8;
    while (L < 64 && L * 4 < nchunk) L <<= 1;             // at most 4 chunks per lane
    while (L < 64 && L < nchunk && nchunk <= 64) L <<= 1;  // one chunk per lane when the row fits a wave
    if (nchunk > 4 * L) return AWSEG_ERANGE;
    const int64_t rows_per_block = kThreads / L;
    const int grid = awseg_grid_1d(n_rows, (int)rows_per_block);
    hipStream_t s = awseg_s(stream);
#define AWSEG_LN(LV) hipLaunchKernelGGL((layernorm_rows_kernel<LV>), dim3(grid), dim3(kThreads), 0, s, x, n_rows, channels, gamma, beta, eps, out)
    switch (L) {
        case 8: AWSEG_LN(8); break;
        case 16: AWSEG_LN(16); break;
        case 32: AWSEG_LN(32); break;
        default: AWSEG_LN(64); break;
    }
#undef AWSEG_LN
    AWSEG_LAUNCH_CHECK();
    return 0;
}

AWSEG_API int awseg_dwconv3x3_upcat_nhwc(const float* a, int a_height, int a_width, int a_channels, const float* hi, int hi_channels,
                                         int64_t batch, int height, int width, const float* w9, float* out, awseg_stream_t stream)
{
    if (batch == 0) return 0;
    if (!a || !hi || !w9 || !out || batch < 0 || height < 1 || width < 1 || a_height < 1 || a_width < 1) return AWSEG_EINVAL;
    if (a_channels < 4 || (a_channels & 3) || hi_channels < 4 || (hi_channels & 3)) return AWSEG_ERANGE;
    if (((uintptr_t)a & 15) || ((uintptr_t)hi & 15) || ((uintptr_t)w9 & 15) || ((uintptr_t)out & 15)) return AWSEG_EALIGN;
    // torch: area_pixel_compute_scale(in, out, align_corners=true) = out > 1 ? (float)(in - 1) / (out - 1) : 0
    const float ry = height > 1 ? (float)(a_height - 1) / (float)(height - 1) : 0.f;
    const float rx = width > 1 ? (float)(a_width - 1) / (float)(width - 1) : 0.f;
    constexpr int SX = 8;
    int c4_lo = 0;
    static const bool no_lds = getenv("AWSEG_UPCAT_STRIP") != nullptr;
    const int ca4 = a_channels / 4;
    if (!no_lds && ca4 % kUpQuads == 0 && batch * (ca4 / kUpQuads) <= 65535 && height <= 65535) {
        // source columns a 32-column tile (+ its two halo columns) touches: the device's own float expressions
        int n_src = 1;
        for (int x0 = 0; x0 < width; x0 += kUpTile) {
            const int first = (int)(rx * (float)(x0 > 0 ? x0 - 1 : 0));
            const int sx = x0 + kUpTile < width - 1 ? x0 + kUpTile : width - 1;
            int last = (int)(rx * (float)sx);
            last += last < a_width - 1 ? 1 : 0;
            if (last - first + 1 > n_src) n_src = last - first + 1;
        }
        const size_t lds = (size_t)n_src * 3 * kUpQuads * sizeof(float4);
        if (lds <= 64 * 1024) {
            auto kern = dwconv3x3_upcat_lds_kernel<SX>;
            if (hipFuncSetAttribute(reinterpret_cast<const void*>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds) != hipSuccess) return AWSEG_EINVAL;
            dim3 grid((width + kUpTile - 1) / kUpTile, height, (unsigned)(batch * (ca4 / kUpQuads)));
            hipLaunchKernelGGL(kern, grid, dim3(kThreads), lds, awseg_s(stream), a, a_height, a_width, a_channels, hi_channels, height,
                               width, ry, rx, w9, out, n_src);
            AWSEG_LAUNCH_CHECK();
            c4_lo = ca4;
        }
    }
    const int64_t items = batch * height * ((width + SX - 1) / SX) * ((a_channels + hi_channels) / 4 - c4_lo);
    hipLaunchKernelGGL((dwconv3x3_upcat_strip_kernel<SX>), dim3(awseg_grid_1d(items, kThreads)), dim3(kThreads), 0, awseg_s(stream),
                       a, a_height, a_width, a_channels, hi, hi_channels, batch, height, width, ry, rx, w9, out, c4_lo);
    AWSEG_LAUNCH_CHECK();
    return 0;
}

AWSEG_API int awseg_maxpool3x3s2_nhwc(const float* x, int64_t batch, int height, int width, int channels, float* out,
                                      awseg_stream_t stream)
{
    if (batch == 0) return 0;
    if (!x || !out || batch < 0 || height < 1 || width < 1 || channels < 4 || (channels & 3)) return AWSEG_EINVAL;
    if (((uintptr_t)x & 15) || ((uintptr_t)out & 15)) return AWSEG_EALIGN;
    const int ho = (height - 1) / 2 + 1, wo = (width - 1) / 2 + 1;     // floor((n + 2 - 3) / 2) + 1
    const int64_t total = batch * ho * ((wo + 1) / 2) * (channels / 4);
    hipLaunchKernelGGL(maxpool3x3s2_nhwc_kernel<false>, dim3(awseg_grid_1d(total, kThreads)), dim3(kThreads), 0, awseg_s(stream), x, batch,
                       height, width, channels, ho, wo, nullptr, out);
    AWSEG_LAUNCH_CHECK();
    return 0;
}

AWSEG_API int awseg_maxpool3x3s2_bias_relu_nhwc(const float* x, int64_t batch, int height, int width, int channels, const float* shift,
                                                float* out, awseg_stream_t stream)
{
    if (batch == 0) return 0;
    if (!x || !out || !shift || batch < 0 || height < 1 || width < 1 || channels < 4) return AWSEG_EINVAL;
    if (channels % 4) return AWSEG_ERANGE;
    if (((uintptr_t)x & 15) || ((uintptr_t)out & 15) || ((uintptr_t)shift & 15)) return AWSEG_EALIGN;
    const int ho = (height - 1) / 2 + 1, wo = (width - 1) / 2 + 1;
    const int64_t total = batch * ho * ((wo + 1) / 2) * (channels / 4);
    hipLaunchKernelGGL(maxpool3x3s2_nhwc_kernel<true>, dim3(awseg_grid_1d(total, kThreads)), dim3(kThreads), 0, awseg_s(stream), x, batch,
                       height, width, channels, ho, wo, shift, out);
    AWSEG_LAUNCH_CHECK();
    return 0;
}

AWSEG_API int awseg_upsample_bilinear_strided(const float* low, int64_t batch, int channels, int low_height, int low_width,
                                              int64_t stride_b, int64_t stride_c, int64_t stride_y, int64_t stride_x,
                                              int height, int width, int align_corners, float* out, awseg_stream_t stream)
{
    const int64_t planes = batch * channels;
    if (planes == 0) return 0;
    if (!low || !out || batch < 0 || channels < 1 || low_height < 1 || low_width < 1 || height < 1 || width < 1) return AWSEG_EINVAL;
    if (stride_b < 0 || stride_c < 0 || stride_y < 0 || stride_x < 0) return AWSEG_EINVAL;
    const bool planar = stride_x == 1 && stride_y == low_width && stride_c == (int64_t)low_height * low_width &&
                        stride_b == stride_c * channels;
    if ((width & 3) == 0 && ((uintptr_t)out & 15)) return AWSEG_EALIGN;
    // torch area_pixel_compute_scale: align_corners ? (out > 1 ? (in - 1) / (out - 1) : 0) : in / out, in float
    const float sy = align_corners ? (height > 1 ? (float)(low_height - 1) / (float)(height - 1) : 0.f) : (float)low_height / (float)height;
    const float sx = align_corners ? (width > 1 ? (float)(low_width - 1) / (float)(width - 1) : 0.f) : (float)low_width / (float)width;
    if ((width & 3) == 0 && sx * 3.0f < 1.0f && sy * 3.0f < 1.0f) {
        const int64_t blocks4 = planes * ((height + 3) / 4) * (width / 4);
        hipLaunchKernelGGL(upsample_bilinear_4x4_kernel, dim3(awseg_grid_1d(blocks4, kThreads)), dim3(kThreads), 0, awseg_s(stream), low,
                           low_height, low_width, height, width, sy, sx, align_corners ? 1 : 0, out, planes, channels, stride_b, stride_c,
                           stride_y, stride_x);
        AWSEG_LAUNCH_CHECK();
        return 0;
    }
    if (!planar) return AWSEG_ERANGE;                             // the row-at-a-time kernel reads planar maps only
    const int64_t total = planes * height * ((width + 3) / 4);
    hipLaunchKernelGGL(upsample_bilinear_kernel, dim3(awseg_grid_1d(total, kThreads)), dim3(kThreads), 0, awseg_s(stream), low,
                       low_height, low_width, height, width, sy, sx, align_corners ? 1 : 0, out, planes);
    AWSEG_LAUNCH_CHECK();
    return 0;
}

AWSEG_API int awseg_upsample_bilinear(const float* low, int64_t planes, int low_height, int low_width, int height, int width,
                                      int align_corners, float* out, awseg_stream_t stream)
{
    const int64_t hw = (int64_t)low_height * low_width;
    return awseg_upsample_bilinear_strided(low, planes, 1, low_height, low_width, hw, hw, low_width, 1, height, width, align_corners, out, stream);
}

AWSEG_API int awseg_depth_upsample_combine(const float* d1, const float* d2_low, int batch, int low_height, int low_width,
                                           int height, int width, const float* weights, float* d2_full, float* d_out,
                                           awseg_stream_t stream)
{
    if (batch == 0) return 0;
    if (!d1 || !d2_low || !d2_full || !d_out || batch < 0 || low_height < 1 || low_width < 1 || height < 1 || width < 1) return AWSEG_EINVAL;
    if (batch > 65535) return AWSEG_ERANGE;
    // torch area_pixel_compute_scale(in, out, align_corners=false, scale=None) = (float)in / out
    const float sy = (float)low_height / (float)height, sx = (float)low_width / (float)width;
    dim3 grid(awseg_grid_1d((int64_t)height * width, kThreads, 1024), (unsigned)batch);
    hipLaunchKernelGGL(depth_up_combine_kernel, grid, dim3(kThreads), 0, awseg_s(stream), d1, d2_low, low_height, low_width, height, width,
                       sy, sx, weights, d2_full, d_out);
    AWSEG_LAUNCH_CHECK();
    return 0;
}
