// heads.hip — A8 SegFormer head with the x32 upsample fused away, A9 ASPP depthwise fusion.
//
// A8 (PKG/models/model.py:209-214): F.interpolate(bilinear, align_corners=False) ->
// Conv3x3(pad 1) -> BatchNorm(eval) -> ReLU -> Conv1x1.  Upsample and 3x3 are linear, so
//     conv3x3(up(f))[o,y,x] = sum_{tap} [tap inside image] sum_{4 cells} wy*wx * G[tap][cell][o]
// with G = the nine 1x1 products W_tap . f at the encoder's resolution (a 2.4 GFLOP GEMM the
// caller runs once).  The 2.15 GB / image full-resolution 256-channel tensor and 2.47 TFLOP /
// image of the as-written op never exist; per pixel we spend 36 gathers x Cmid + Cmid x Cout.
#include "awseg_common.h"

namespace {

constexpr int kThreads = 256;
constexpr int HPX = 32;            // pixels of one row handled per block

// torch area_pixel_compute_source_index(align_corners=False): src = max(scale*(dst+.5)-.5, 0)
struct src_idx { int i0, i1; float l0, l1; };
__device__ __forceinline__ src_idx bilinear_src(int dst, float scale, int in_size)
{
    float s = scale * ((float)dst + 0.5f) - 0.5f;
    if (s < 0.f) s = 0.f;
    src_idx r;
    r.i0 = (int)s;
    if (r.i0 > in_size - 1) r.i0 = in_size - 1;
    r.i1 = r.i0 + (r.i0 < in_size - 1 ? 1 : 0);
    r.l1 = s - (float)r.i0;
    r.l0 = 1.f - r.l1;
    return r;
}

// v1 (VALU): block = one row segment of HPX pixels; threads sweep the mid channels with
// coalesced G reads, stage relu(bn(mid)) in LDS, then (pixel, class) dot products.
__global__ __launch_bounds__(kThreads)
void segformer_head_kernel(const float* __restrict__ g9, int cmid, int h, int w, int H, int W,
                           const float* __restrict__ scale, const float* __restrict__ shift,
                           const float* __restrict__ w2, const float* __restrict__ b2, int cout,
                           float* __restrict__ out)
{
    extern __shared__ float s_mid[];            // [HPX][cmid]
    __shared__ int s_cell[HPX][36];
    __shared__ float s_coef[HPX][36];
    const int b = blockIdx.z, y = blockIdx.y, x0 = blockIdx.x * HPX;
    const float sh = (float)h / (float)H, sw = (float)w / (float)W;
    const float* g = g9 + (int64_t)b * h * w * 9 * cmid;
    // per (pixel, tap, neighbour): flat offset into g (cell*9+tap)*cmid and weight (0 when padded)
    for (int i = threadIdx.x; i < HPX * 36; i += kThreads) {
        int px = i / 36, r = i - px * 36;
        int tap = r >> 2, nb = r & 3;
        int ky = tap / 3, kx = tap - ky * 3;
        int yy = y + ky - 1, xx = x0 + px + kx - 1;
        float cf = 0.f; int cell = 0;
        if (yy >= 0 && yy < H && xx >= 0 && xx < W && x0 + px < W) {
            src_idx sy = bilinear_src(yy, sh, h), sx = bilinear_src(xx, sw, w);
            int ci = (nb & 2) ? sy.i1 : sy.i0, cj = (nb & 1) ? sx.i1 : sx.i0;
            cf = ((nb & 2) ? sy.l1 : sy.l0) * ((nb & 1) ? sx.l1 : sx.l0);
            cell = (ci * w + cj) * 9 + tap;
        }
        s_cell[px][r] = cell; s_coef[px][r] = cf;
    }
    __syncthreads();
    for (int o = threadIdx.x; o < cmid; o += kThreads) {
        const float sc = scale[o], sf = shift[o];
        for (int px = 0; px < HPX; ++px) {
            float acc = 0.f;
#pragma unroll 4
            for (int r = 0; r < 36; ++r) acc = fmaf(s_coef[px][r], g[(int64_t)s_cell[px][r] * cmid + o], acc);
            float v = fmaf(acc, sc, sf);
            s_mid[px * cmid + o] = v > 0.f ? v : 0.f;
        }
    }
    __syncthreads();
    const int64_t HW = (int64_t)H * W;
    for (int i = threadIdx.x; i < HPX * cout; i += kThreads) {
        int k = i / HPX, px = i - k * HPX;
        if (x0 + px >= W) continue;
        float acc = b2[k];
        const float* wk = w2 + (int64_t)k * cmid;
        const float* m = s_mid + px * cmid;
        for (int o = 0; o < cmid; ++o) acc = fmaf(wk[o], m[o], acc);
        out[((int64_t)b * cout + k) * HW + (int64_t)y * W + x0 + px] = acc;
    }
}

// A9: depthwise atrous 3x3 for the three ASPP rates in one pass over x (NHWC, float4 over C).
__global__ __launch_bounds__(kThreads)
void aspp_dw3_kernel(const float* __restrict__ x, int64_t batch, int h, int w, int C,
                     const float* __restrict__ wdw, int r0, int r1, int r2, float* __restrict__ out)
{
    const int c4n = C / 4;
    const int64_t total = batch * h * w * c4n;
    const int64_t plane = batch * (int64_t)h * w * C;
    const int rates[3] = { r0, r1, r2 };
    for (int64_t i = (int64_t)blockIdx.x * kThreads + threadIdx.x; i < total; i += (int64_t)gridDim.x * kThreads) {
        int c4 = (int)(i % c4n);
        int64_t p = i / c4n;
        int xx = (int)(p % w); int64_t t = p / w;
        int yy = (int)(t % h); int64_t b = t / h;
        const float* xb = x + b * (int64_t)h * w * C;
#pragma unroll
        for (int r = 0; r < 3; ++r) {
            const int d = rates[r];
            float4 acc = make_float4(0.f, 0.f, 0.f, 0.f);
#pragma unroll
            for (int ky = 0; ky < 3; ++ky) {
                int sy = yy + (ky - 1) * d;
                if (sy < 0 || sy >= h) continue;
#pragma unroll
                for (int kx = 0; kx < 3; ++kx) {
                    int sx = xx + (kx - 1) * d;
                    if (sx < 0 || sx >= w) continue;
                    float4 v = *reinterpret_cast<const float4*>(xb + ((int64_t)sy * w + sx) * C + c4 * 4);
                    float4 k = *reinterpret_cast<const float4*>(wdw + ((int64_t)r * 9 + ky * 3 + kx) * C + c4 * 4);
                    acc.x = fmaf(v.x, k.x, acc.x); acc.y = fmaf(v.y, k.y, acc.y);
                    acc.z = fmaf(v.z, k.z, acc.z); acc.w = fmaf(v.w, k.w, acc.w);
                }
            }
            *reinterpret_cast<float4*>(out + (int64_t)r * plane + p * C + c4 * 4) = acc;
        }
    }
}

}  // namespace

AWSEG_API int awseg_segformer_head_fused(const float* g9, int64_t batch, int cmid, int h, int w, int height, int width,
                                         const float* scale, const float* shift, const float* w2, const float* b2,
                                         int cout, float* out, awseg_stream_t stream)
{
    if (!g9 || !scale || !shift || !w2 || !b2 || !out) return AWSEG_EINVAL;
    if (batch < 1 || cmid < 1 || h < 1 || w < 1 || height < 1 || width < 1 || cout < 1 || cout > 32) return AWSEG_EINVAL;
    if (batch > 65535 || height > 65535) return AWSEG_ERANGE;
    if ((int64_t)h * w * 9 * cmid > 0x7fffffffLL) return AWSEG_ERANGE;
    const size_t lds = (size_t)HPX * cmid * sizeof(float);
    if (lds > 96 * 1024) return AWSEG_ERANGE;
    dim3 grid((width + HPX - 1) / HPX, height, (unsigned)batch);
    hipLaunchKernelGGL(segformer_head_kernel, grid, dim3(kThreads), lds, awseg_s(stream), g9, cmid, h, w, height, width,
                       scale, shift, w2, b2, cout, out);
    AWSEG_LAUNCH_CHECK();
    return 0;
}

AWSEG_API int awseg_aspp_depthwise3(const float* x, int64_t batch, int h, int w, int channels, const float* wdw,
                                    int rate0, int rate1, int rate2, float* out, awseg_stream_t stream)
{
    if (!x || !wdw || !out || batch < 1 || h < 1 || w < 1 || channels < 4 || (channels & 3)) return AWSEG_EINVAL;
    if (((uintptr_t)x & 15) || ((uintptr_t)wdw & 15) || ((uintptr_t)out & 15)) return AWSEG_EALIGN;
    const int64_t total = batch * h * w * (channels / 4);
    hipLaunchKernelGGL(aspp_dw3_kernel, dim3(awseg_grid_1d(total, kThreads)), dim3(kThreads), 0, awseg_s(stream), x, batch,
                       h, w, channels, wdw, rate0, rate1, rate2, out);
    AWSEG_LAUNCH_CHECK();
    return 0;
}
