// depthfuse.hip — the bilinear FORMS of relu(bn(conv3x3(interpolate_x32(f)))): what lets the SegFormer depth head run as ONE
// launch (wino_split.hip, wino8p_kernel MODE 2) without the 128-channel full-resolution map between its two 3x3 convolutions.
//
// Reference: DepthEstimationHead's first Conv3x3 + BatchNorm + ReLU (PKG/models/model.py:42-45) applied to
// F.interpolate(features, size=(H, W), mode='bilinear', align_corners=False) (:211, :219-221), H = 32 h, W = 32 w.
//
// Upsample and convolution are linear: conv3x3(up(f)) = sum_tap shift_tap(up(G_tap)), G_tap = (W_tap * bn_scale) . f at the
// encoder's resolution (g9, the same operand awseg_upconv3x3_bn_relu takes).  With align_corners=False and a factor of 32 the
// source coordinate of pixel x is (x - 15.5) / 32: pixels 16 + 32 k .. 47 + 32 k interpolate samples k and k + 1 with
// lambda = (t + 0.5) / 32, t = x - (16 + 32 k) — a CELL; pixels 0..15 and 32 w - 16 .. 32 w - 1 are the constant half cells -1 and
// w - 1.  For a pixel whose whole 3x3 window lies inside one cell (both axes) and inside the image, every tap is
//     g00 + gx lx + gy ly + gxy lx ly,  lx = (t + dx + 0.5) / 32,  ly = (s + dy + 0.5) / 32
// so the pre-activation is A + B t + C s + D t s with per-cell, per-channel coefficients.  The other pixels are
//   * special COLUMNS x in {0, 32 w - 1} u {15, 16 mod 32}: x is fixed, the form is E + F s        (rows free in their cell)
//   * special ROWS, the same set along y:                    y is fixed, the form is E' + F' t
//   * their crossings:                                                   a constant
// (taps outside the image are dropped: the zero padding of the convolution).  Tables, per frame, float32, channel innermost:
//     F4 [row selector 0 .. 3h+2][cell column 0 .. w][A B C D][C]      row selector: cell row ky + 1 (0 .. h), then h + 1 + special row id
//     F2 [row selector]          [special column id 0 .. 2w+1][E F][C]
// special ids along an axis of n samples: 0 -> pixel 0, 1 + 2k -> 15 + 32 k, 2 + 2k -> 16 + 32 k, 2n + 1 -> 32 n - 1.  Special rows
// store (E', F', 0, 0) / (const, 0), so one evaluation R = A + C s, S = B + D s, value = R + S t (or E + F s) serves every pixel.
// The folded BatchNorm shift is part of A / E / the constants.  Sums in float64 (the row-tap partial sums pass through float32 once).
#include "awseg_common.h"

namespace {

struct tap1d { int i0, i1; double p, q; bool ok; };

__device__ __forceinline__ tap1d cell_tap(int k, int n, double p, double q)
{
    tap1d t; t.ok = true;
    if (k < 0) { t.i0 = 0; t.i1 = 0; t.p = 0.0; t.q = 0.0; }
    else if (k >= n - 1) { t.i0 = n - 1; t.i1 = n - 1; t.p = 0.0; t.q = 0.0; }
    else { t.i0 = k; t.i1 = k + 1; t.p = p; t.q = q; }
    return t;
}
// tap d of a free interior pixel of cell k: lambda = (d + 0.5) / 32 + t / 32
__device__ __forceinline__ tap1d tap_free(int k, int d, int n) { return cell_tap(k, n, (d + 0.5) / 32.0, 1.0 / 32.0); }
// tap d of the fixed pixel x
__device__ __forceinline__ tap1d tap_fixed(int x, int d, int n)
{
    const int xx = x + d;
    if (xx < 0 || xx >= 32 * n) { tap1d t; t.ok = false; t.i0 = t.i1 = 0; t.p = t.q = 0.0; return t; }
    const int k = xx < 16 ? -1 : (xx - 16) >> 5;
    return cell_tap(k, n, (xx - (16 + 32 * k) + 0.5) / 32.0, 0.0);
}
__device__ __forceinline__ int special_coord(int id, int n)
{
    if (id == 0) return 0;
    if (id == 2 * n + 1) return 32 * n - 1;
    return (id & 1) ? 15 + 32 * ((id - 1) >> 1) : 16 + 32 * ((id - 2) >> 1);
}

// One block = (frame, row selector, slice of CB channels).  The sum over the taps is separable: first the three row taps are
// contracted for every (column tap dx, low-resolution column jx) — H[dx][jx] = const + slope * s, six loads of g9 each — into LDS,
// then every column descriptor combines six H entries.  (A thread per table entry reading its 36 operands itself moved 2.9 GB
// through L2 per batch of 8 and took 0.47 ms; this form reads 0.5 GB.)
constexpr int FCB = 32;
__global__ __launch_bounds__(256)
void upconv_forms_kernel(const float* __restrict__ g9, int h, int w, int C, const float* __restrict__ shift, float* __restrict__ forms,
                         int64_t img_floats, int64_t f4_floats)
{
    extern __shared__ float sH[];                                    // [3 dx][w][const | slope][FCB]
    const int b = blockIdx.z, rs = blockIdx.y, c0 = blockIdx.x * FCB;
    const int tid = threadIdx.x;
    tap1d ry[3];
#pragma unroll
    for (int d = 0; d < 3; ++d) ry[d] = rs <= h ? tap_free(rs - 1, d - 1, h) : tap_fixed(special_coord(rs - (h + 1), h), d - 1, h);
    const float* G = g9 + (int64_t)b * h * w * 9 * C;
    float* F = forms + (int64_t)b * img_floats;
    // (four items per thread in flight: one item's six dependent-free loads alone leave the block waiting on memory latency)
#pragma unroll 4
    for (int item = tid; item < 3 * w * FCB; item += 256) {
        const int c = item % FCB, jx = (item / FCB) % w, dx = item / (FCB * w);
        double hc = 0.0, hs = 0.0;
        if (c0 + c < C) {
#pragma unroll
            for (int iy = 0; iy < 3; ++iy) {
                if (!ry[iy].ok) continue;
                const int tap = iy * 3 + dx;
                const double g0 = G[(((int64_t)ry[iy].i0 * w + jx) * 9 + tap) * C + c0 + c];
                const double g1 = G[(((int64_t)ry[iy].i1 * w + jx) * 9 + tap) * C + c0 + c];
                hc += g0 * (1.0 - ry[iy].p) + g1 * ry[iy].p;
                hs += (g1 - g0) * ry[iy].q;
            }
        }
        sH[((dx * w + jx) * 2 + 0) * FCB + c] = (float)hc;
        sH[((dx * w + jx) * 2 + 1) * FCB + c] = (float)hs;
    }
    __syncthreads();
    for (int item = tid; item < (3 * w + 3) * FCB; item += 256) {
        const int c = item % FCB, cd = item / FCB;                   // column descriptor: cells 0 .. w, then special columns
        if (c0 + c >= C) continue;
        double k0 = (double)shift[c0 + c], kt = 0.0, ks = 0.0, kts = 0.0;
#pragma unroll
        for (int ix = 0; ix < 3; ++ix) {
            const tap1d rx = cd <= w ? tap_free(cd - 1, ix - 1, w) : tap_fixed(special_coord(cd - (w + 1), w), ix - 1, w);
            if (!rx.ok) continue;
            const double h0c = sH[((ix * w + rx.i0) * 2 + 0) * FCB + c], h0s = sH[((ix * w + rx.i0) * 2 + 1) * FCB + c];
            const double h1c = sH[((ix * w + rx.i1) * 2 + 0) * FCB + c], h1s = sH[((ix * w + rx.i1) * 2 + 1) * FCB + c];
            k0 += h0c * (1.0 - rx.p) + h1c * rx.p;
            kt += (h1c - h0c) * rx.q;
            ks += h0s * (1.0 - rx.p) + h1s * rx.p;
            kts += (h1s - h0s) * rx.q;
        }
        if (cd <= w) {
            float* o = F + (((int64_t)rs * (w + 1) + cd) * 4) * C + c0 + c;
            o[0] = (float)k0; o[C] = (float)kt; o[2 * (int64_t)C] = (float)ks; o[3 * (int64_t)C] = (float)kts;
        } else {
            float* o = F + f4_floats + (((int64_t)rs * (2 * w + 2) + (cd - (w + 1))) * 2) * C + c0 + c;
            o[0] = (float)k0; o[C] = (float)ks;                      // (a fixed column has no t terms)
        }
    }
}

}  // namespace

AWSEG_API int64_t awseg_upconv_forms_floats(int h, int w, int cmid)
{
    if (h < 1 || w < 1 || cmid < 1) return -1;
    return (int64_t)(3 * h + 3) * (w + 1) * 4 * cmid + (int64_t)(3 * h + 3) * (2 * w + 2) * 2 * cmid;
}

AWSEG_API int awseg_upconv_forms(const float* g9, int batch, int cmid, int h, int w, const float* shift, float* forms, awseg_stream_t stream)
{
    if (batch == 0) return 0;
    if (!g9 || !shift || !forms || batch < 0 || h < 1 || w < 1 || cmid < 1) return AWSEG_EINVAL;
    if (batch > 65535 || 3 * h + 3 > 65535) return AWSEG_ERANGE;
    const int64_t f4 = (int64_t)(3 * h + 3) * (w + 1) * 4 * cmid;
    const size_t lds = (size_t)3 * w * 2 * FCB * sizeof(float);
    if (lds > 64 * 1024) return AWSEG_ERANGE;                       // w <= 85 (a 2720-pixel-wide frame); wider: the two-launch path
    dim3 grid((cmid + FCB - 1) / FCB, 3 * h + 3, batch);
    hipLaunchKernelGGL(upconv_forms_kernel, grid, dim3(256), lds, awseg_s(stream), g9, h, w, cmid, shift, forms,
                       awseg_upconv_forms_floats(h, w, cmid), f4);
    AWSEG_LAUNCH_CHECK();
    return 0;
}
