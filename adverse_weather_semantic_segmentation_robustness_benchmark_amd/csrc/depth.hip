// depth.hip — SURVEY §8(f) "next" #2: DepthEstimationPreprocessor.estimate_depth
// (PKG/data/preprocessing.py:304-367), the per-sample depth TARGET the loader attaches when
// include_depth is set (PKG/data/loader.py:270-272).
//
//   gray  = cv2.cvtColor(image, COLOR_RGB2GRAY)                        :338
//   base  = (y/h)*0.8 + 0.2; rows < h//3 -> 1.0; rows >= h//2 -> *0.5  :340-354
//   tex   = cv2.Laplacian(gray, CV_64F)                                :358
//   depth = clip(base - 0.3*|tex|/(max|tex| + 1e-8), 0, 1)             :359-363
//   depth = scipy gaussian_filter(depth, sigma=2)                      :366
//
// Two launches per batch: (1) per-image max |Laplacian| — integer, order-free, one atomicMax
// per block; (2) a tile kernel that rebuilds gray + Laplacian for the tile and its 8-pixel halo
// in LDS, applies the float64 ladder and runs scipy's 17-tap separable Gaussian (axis 0 first,
// centre tap then symmetric pairs outermost-in) — the same staging as the fog depth kernel.
// OpenCV arithmetic restated (cv2 is not in the image; parity unpinned for these two steps):
// 8-bit RGB2GRAY = (R*9798 + G*19235 + B*3735 + 2^14) >> 15; Laplacian ksize=1 = 3x3 cross,
// BORDER_REFLECT_101.
#include "awseg_common.h"

namespace {

constexpr int kThreads = 256;
constexpr int FR = AWSEG_GAUSS_RADIUS;       // 8
constexpr int TW = 64, TH = 32;
constexpr int IW = TW + 2 * FR;              // 80 staged depth columns
constexpr int IH = TH + 2 * FR;              // 48 staged depth rows
constexpr int GW = IW + 2, GH = IH + 2;      // gray window: one more pixel for the Laplacian

struct gauss_taps { double w[2 * FR + 1]; };

__device__ __forceinline__ int reflect_sym(int i, int n)          // scipy 'reflect'
{
    if (n == 1) return 0;
    const int period = 2 * n;
    i %= period;
    if (i < 0) i += period;
    return i < n ? i : period - 1 - i;
}
__device__ __forceinline__ int reflect_101(int i, int n)          // BORDER_REFLECT_101, |offset| == 1
{
    if (n == 1) return 0;
    if (i < 0) return -i;
    if (i >= n) return 2 * n - 2 - i;
    return i;
}
__device__ __forceinline__ int reflect_101n(int i, int n)         // BORDER_REFLECT_101, any offset (cv::borderInterpolate)
{
    if (n == 1) return 0;
    while (i < 0 || i >= n) i = i < 0 ? -i : 2 * n - 2 - i;
    return i;
}
__device__ __forceinline__ int gray15(const uint8_t* __restrict__ px)
{
    return ((int)px[0] * 9798 + (int)px[1] * 19235 + (int)px[2] * 3735 + (1 << 14)) >> 15;
}

// Gray window of a tile: rows [ylo, ylo+gh), cols [xlo, xlo+gw) of the image, clamped to the image.
// Every pixel whose Laplacian the tile needs (tile + halo after scipy reflection) and its four
// REFLECT_101 neighbours lie inside it.
struct gray_win { int ylo, xlo, gh, gw; };
__device__ __forceinline__ gray_win window_for(int y0, int x0, int th, int tw, int halo, int H, int W)
{
    gray_win g;
    g.ylo = y0 - halo - 1; if (g.ylo < 0) g.ylo = 0;
    g.xlo = x0 - halo - 1; if (g.xlo < 0) g.xlo = 0;
    int yhi = y0 + th + halo; if (yhi > H - 1) yhi = H - 1;
    int xhi = x0 + tw + halo; if (xhi > W - 1) xhi = W - 1;
    g.gh = yhi - g.ylo + 1; g.gw = xhi - g.xlo + 1;
    return g;
}
template <int SW>
__device__ __forceinline__ void stage_gray(const uint8_t* __restrict__ src, int W, const gray_win& g, uint8_t* s_gray)
{
    for (int i = threadIdx.x; i < g.gh * g.gw; i += kThreads) {
        int ty = i / g.gw, tx = i - ty * g.gw;
        s_gray[ty * SW + tx] = (uint8_t)gray15(src + ((int64_t)(g.ylo + ty) * W + (g.xlo + tx)) * 3);
    }
}
template <int SW>
__device__ __forceinline__ int abs_laplacian(const uint8_t* s_gray, const gray_win& g, int gy, int gx, int H, int W)
{
    const int ym = reflect_101(gy - 1, H) - g.ylo, yp = reflect_101(gy + 1, H) - g.ylo;
    const int xm = reflect_101(gx - 1, W) - g.xlo, xp = reflect_101(gx + 1, W) - g.xlo;
    const int yc = gy - g.ylo, xc = gx - g.xlo;
    int v = (int)s_gray[ym * SW + xc] + (int)s_gray[yp * SW + xc] + (int)s_gray[yc * SW + xm] + (int)s_gray[yc * SW + xp]
          - 4 * (int)s_gray[yc * SW + xc];
    return v < 0 ? -v : v;
}

// (1) per-image max |Laplacian(gray)|.  tex_max[b] must be zero on entry (the launcher clears it).
__global__ __launch_bounds__(kThreads)
void texture_max_kernel(const uint8_t* __restrict__ imgs, int H, int W, int* __restrict__ tex_max)
{
    __shared__ uint8_t s_gray[(TH + 2) * (TW + 2)];
    __shared__ int s_red[kThreads / AWSEG_WAVE];
    const int64_t hw = (int64_t)H * W;
    const uint8_t* src = imgs + (int64_t)blockIdx.z * hw * 3;
    const int x0 = blockIdx.x * TW, y0 = blockIdx.y * TH;
    const gray_win g = window_for(y0, x0, TH, TW, 0, H, W);
    stage_gray<TW + 2>(src, W, g, s_gray);
    __syncthreads();
    int m = 0;
    for (int i = threadIdx.x; i < TH * TW; i += kThreads) {
        int ty = i / TW, tx = i - ty * TW;
        int gy = y0 + ty, gx = x0 + tx;
        if (gy < H && gx < W) {
            int v = abs_laplacian<TW + 2>(s_gray, g, gy, gx, H, W);
            m = v > m ? v : m;
        }
    }
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) { int t = __shfl_down(m, o, 64); m = t > m ? t : m; }
    if ((threadIdx.x & 63) == 0) s_red[threadIdx.x >> 6] = m;
    __syncthreads();
    if (threadIdx.x == 0) {
        for (int k = 1; k < kThreads / AWSEG_WAVE; ++k) m = s_red[k] > m ? s_red[k] : m;
        if (m > 0) atomicMax(tex_max + blockIdx.z, m);
    }
}

// (2) depth tile: float64 ladder on tile + halo, separable Gaussian, float64 and/or float32 out.
__global__ __launch_bounds__(kThreads)
void depth_estimate_kernel(const uint8_t* __restrict__ imgs, int H, int W, const int* __restrict__ tex_max, gauss_taps taps,
                           double* __restrict__ out64, float* __restrict__ out32)
{
    __shared__ double s_in[IH * IW];          // 30 KB
    __shared__ double s_v[TH * IW];           // 20 KB
    __shared__ uint8_t s_gray[GH * GW];       // 4 KB
    const int64_t hw = (int64_t)H * W;
    const uint8_t* src = imgs + (int64_t)blockIdx.z * hw * 3;
    const int x0 = blockIdx.x * TW, y0 = blockIdx.y * TH;
    const gray_win g = window_for(y0, x0, TH, TW, FR, H, W);
    stage_gray<GW>(src, W, g, s_gray);
    const double denom = (double)tex_max[blockIdx.z] + 1e-8;      // :359
    const int y_last = (y0 + TH < H ? y0 + TH : H) - 1, x_last = (x0 + TW < W ? x0 + TW : W) - 1;
    __syncthreads();
    // phase 1: clipped pre-smoothing depth for the tile and its halo (scipy reflect at the border)
    for (int i = threadIdx.x; i < IH * IW; i += kThreads) {
        int ty = i / IW, tx = i - ty * IW;
        // cells only outputs beyond the image edge would read are not produced (their reflections
        // may fall outside the staged gray window)
        if (y0 - FR + ty > y_last + FR || x0 - FR + tx > x_last + FR) { s_in[i] = 0.0; continue; }
        int gy = reflect_sym(y0 - FR + ty, H), gx = reflect_sym(x0 - FR + tx, W);
        double base = ((double)gy / (double)H) * 0.8 + 0.2;        // :348-349
        if (gy < H / 3) base = 1.0;                                // :353
        if (gy >= H / 2) base = base * 0.5;                        // :354
        double ts = (double)abs_laplacian<GW>(s_gray, g, gy, gx, H, W) / denom;
        double adj = -0.3 * ts;                                    // :362
        double v = base + adj;
        s_in[i] = v < 0.0 ? 0.0 : (v > 1.0 ? 1.0 : v);             // :363
    }
    __syncthreads();
    // phase 2: axis-0 pass
    for (int i = threadIdx.x; i < TH * IW; i += kThreads) {
        int ty = i / IW, tx = i - ty * IW;
        const double* c = s_in + (ty + FR) * IW + tx;
        double o = c[0] * taps.w[FR];
#pragma unroll
        for (int j = -FR; j < 0; ++j) { double s = c[j * IW] + c[-j * IW]; double m = s * taps.w[FR + j]; o = o + m; }
        s_v[i] = o;
    }
    __syncthreads();
    // phase 3: axis-1 pass, 4 adjacent outputs per lane
    double* d64 = out64 ? out64 + (int64_t)blockIdx.z * hw : nullptr;
    float* d32 = out32 ? out32 + (int64_t)blockIdx.z * hw : nullptr;
    for (int q = threadIdx.x; q < TH * (TW / 4); q += kThreads) {
        int ty = q / (TW / 4), tq = q - ty * (TW / 4);
        int gy = y0 + ty, gx = x0 + tq * 4;
        if (gy >= H || gx >= W) continue;
        double win[4 + 2 * FR];
        const double* row = s_v + ty * IW + tq * 4;
#pragma unroll
        for (int k = 0; k < 4 + 2 * FR; ++k) win[k] = row[k];
        double depth[4];
#pragma unroll
        for (int k = 0; k < 4; ++k) {
            double o = win[k + FR] * taps.w[FR];
#pragma unroll
            for (int j = -FR; j < 0; ++j) { double s = win[k + FR + j] + win[k + FR - j]; double m = s * taps.w[FR + j]; o = o + m; }
            depth[k] = o;
        }
        const int nvalid = (W - gx) < 4 ? (W - gx) : 4;
        const int64_t p = (int64_t)gy * W + gx;
        if (nvalid == 4 && (p & 3) == 0) {
            if (d64) { double2* d2 = reinterpret_cast<double2*>(d64 + p); d2[0] = make_double2(depth[0], depth[1]); d2[1] = make_double2(depth[2], depth[3]); }
            if (d32) *reinterpret_cast<float4*>(d32 + p) = make_float4((float)depth[0], (float)depth[1], (float)depth[2], (float)depth[3]);
        } else {
            for (int k = 0; k < nvalid; ++k) {
                if (d64) d64[p + k] = depth[k];
                if (d32) d32[p + k] = (float)depth[k];               // torch.from_numpy(depth).float(), loader.py:290
            }
        }
    }
}

// get_fog_density_map's local contrast (PKG/data/preprocessing.py:270-278): gray float32 in [0,1],
// local_mean = 5x5 box (cv2.filter2D, kernel ones/25, BORDER_REFLECT_101), local_variance = 5x5 box
// of (gray - local_mean)^2, contrast = sqrt(variance).  One tile kernel: gray for tile + 4-pixel
// halo, mean and squared deviation for tile + 2-pixel halo, variance for the tile, all in LDS.
// Taps are accumulated row-major as k*v with k = 1/25 in float32 (OpenCV's own SIMD order is not
// specified, so this step is parity-unpinned to ~1e-6).
constexpr int CT = 32;                         // 32 x 32 outputs per block
constexpr int CG = CT + 8, CM = CT + 4;
__global__ __launch_bounds__(kThreads)
void local_contrast_kernel(const uint8_t* __restrict__ imgs, int H, int W, float* __restrict__ out)
{
    __shared__ float s_g[CG * CG];
    __shared__ float s_d[CM * CM];
    const int64_t hw = (int64_t)H * W;
    const uint8_t* src = imgs + (int64_t)blockIdx.z * hw * 3;
    const int x0 = blockIdx.x * CT, y0 = blockIdx.y * CT;
    // gray at the positions a REFLECT_101 border maps the window onto (two nested 2-pixel filters:
    // the outer filter reflects positions first, the inner one reflects around those)
    for (int i = threadIdx.x; i < CG * CG; i += kThreads) {
        int ty = i / CG, tx = i - ty * CG;
        int gy = y0 - 4 + ty, gx = x0 - 4 + tx;
        float v = 0.f;
        if (gy >= -2 && gy < H + 2 && gx >= -2 && gx < W + 2) {        // only positions some in-image mean needs
            const int ry = reflect_101n(gy, H), rx = reflect_101n(gx, W);
            v = (float)gray15(src + ((int64_t)ry * W + rx) * 3) / 255.0f;   // :271-272
        }
        s_g[i] = v;
    }
    __syncthreads();
    const float k = 1.0f / 25.0f;
    // squared deviation at in-image positions of tile + 2 halo
    for (int i = threadIdx.x; i < CM * CM; i += kThreads) {
        int ty = i / CM, tx = i - ty * CM;
        int gy = y0 - 2 + ty, gx = x0 - 2 + tx;
        float d = 0.f;
        if (gy >= 0 && gy < H && gx >= 0 && gx < W) {
            float m = 0.f;
#pragma unroll
            for (int dy = 0; dy < 5; ++dy)
#pragma unroll
                for (int dx = 0; dx < 5; ++dx) { float t = k * s_g[(ty + dy) * CG + tx + dx]; m = m + t; }
            float c = s_g[(ty + 2) * CG + tx + 2] - m;
            d = c * c;                                                       // :277
        }
        s_d[i] = d;
    }
    __syncthreads();
    float* dst = out + (int64_t)blockIdx.z * hw;
    for (int i = threadIdx.x; i < CT * CT; i += kThreads) {
        int ty = i / CT, tx = i - ty * CT;
        int gy = y0 + ty, gx = x0 + tx;
        if (gy >= H || gx >= W) continue;
        float v = 0.f;
#pragma unroll
        for (int dy = -2; dy <= 2; ++dy)
#pragma unroll
            for (int dx = -2; dx <= 2; ++dx) {
                const int ry = reflect_101n(gy + dy, H), rx = reflect_101n(gx + dx, W);   // REFLECT_101 of the deviation map
                float t = k * s_d[(ry - y0 + 2) * CM + (rx - x0 + 2)];
                v = v + t;
            }
        dst[(int64_t)gy * W + gx] = sqrtf(v);                                // :278
    }
}

}  // namespace

AWSEG_API size_t awseg_depth_estimate_workspace(int batch)
{
    return batch > 0 ? (size_t)batch * sizeof(int) : 0;
}

AWSEG_API int awseg_depth_estimate(const uint8_t* imgs, int batch, int height, int width, const double* taps_host,
                                   void* workspace, double* depth_f64, float* depth_f32, awseg_stream_t stream)
{
    if (batch == 0) return 0;
    if (!imgs || !taps_host || !workspace || (!depth_f64 && !depth_f32) || batch < 0 || height < 1 || width < 1) return AWSEG_EINVAL;
    if (batch > 65535 || (height + TH - 1) / TH > 65535) return AWSEG_ERANGE;
    if (((uintptr_t)workspace & 3) || (depth_f64 && ((uintptr_t)depth_f64 & 15)) || (depth_f32 && ((uintptr_t)depth_f32 & 15))) return AWSEG_EALIGN;
    hipStream_t s = awseg_s(stream);
    gauss_taps t;
    for (int i = 0; i < 2 * FR + 1; ++i) t.w[i] = taps_host[i];
    hipError_t e = hipMemsetAsync(workspace, 0, (size_t)batch * sizeof(int), s);
    if (e != hipSuccess) return (int)e;
    dim3 grid((width + TW - 1) / TW, (height + TH - 1) / TH, (unsigned)batch);
    hipLaunchKernelGGL(texture_max_kernel, grid, dim3(kThreads), 0, s, imgs, height, width, (int*)workspace);
    AWSEG_LAUNCH_CHECK();
    hipLaunchKernelGGL(depth_estimate_kernel, grid, dim3(kThreads), 0, s, imgs, height, width, (const int*)workspace, t, depth_f64, depth_f32);
    AWSEG_LAUNCH_CHECK();
    return 0;
}

AWSEG_API int awseg_local_contrast(const uint8_t* imgs, int batch, int height, int width, float* contrast, awseg_stream_t stream)
{
    if (batch == 0) return 0;
    if (!imgs || !contrast || batch < 0 || height < 1 || width < 1) return AWSEG_EINVAL;
    if (batch > 65535 || (height + CT - 1) / CT > 65535) return AWSEG_ERANGE;
    dim3 grid((width + CT - 1) / CT, (height + CT - 1) / CT, (unsigned)batch);
    hipLaunchKernelGGL(local_contrast_kernel, grid, dim3(kThreads), 0, awseg_s(stream), imgs, height, width, contrast);
    AWSEG_LAUNCH_CHECK();
    return 0;
}
