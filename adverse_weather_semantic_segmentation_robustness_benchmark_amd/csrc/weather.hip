// weather.hip — A1-A7 kernels: synthetic weather corruption of uint8 HWC frames.
//
// Reference arithmetic (PKG/data/preprocessing.py): fog :113-123 + depth :235-246,
// night :209-225, rain :131-168, snow :176-202; normalise PKG/data/loader.py:195-198.
//
// Every kernel is batched: `imgs` is the whole [B,H,W,3] uint8 batch, a device array of
// per-image jobs says which frames get this effect and with which parameters, and
// blockIdx.z walks the jobs — one launch covers every frame of a batch that drew the same
// weather condition, so a launch moves tens of MB instead of one 6 MB frame.
// All arithmetic follows the reference's dtype ladder exactly (float32 where numpy stays in
// float32, float64 where numpy promotes) and the file is built with -ffp-contract=off.
#include "awseg_common.h"

namespace {

constexpr int kThreads = 256;

struct norm_consts { float mean[3]; float std[3]; };

// Per-image jobs travel in the kernel arguments (<= 16 per launch): no device-side job array, no
// dependent global load at the top of every block, no host->device copy per call.
constexpr int kMaxJobs = 16;
template <typename J> struct job_pack { J j[kMaxJobs]; };

__device__ __forceinline__ uint8_t quant_f64(double v)
{
    // (np.clip(v,0,1)*255).astype(np.uint8)
    double c = v < 0.0 ? 0.0 : (v > 1.0 ? 1.0 : v);
    return (uint8_t)(int)(c * 255.0);
}
__device__ __forceinline__ uint8_t quant_f32(float v)
{
    float c = v < 0.f ? 0.f : (v > 1.f ? 1.f : v);
    return (uint8_t)(int)(c * 255.0f);
}
__device__ __forceinline__ float norm1(uint8_t q, float mean, float std)
{
    float v = (float)q / 255.0f;
    float d = v - mean;
    return d / std;
}

// Per-block lookup tables (4 KB of LDS): u8 -> (float)u/255 (preprocessing.py:81) and, per channel,
// u8 -> ((float)q/255 - mean)/std (loader.py:195-198).  Entries are produced by the very same
// separately rounded float32 operations, so a lookup is bit-identical to computing in place — it
// just replaces two IEEE divisions per value by one ds_read.
struct weather_lut { float in[256]; float nrm[3][256]; };
__device__ __forceinline__ void lut_fill(weather_lut& L, const norm_consts& nc, bool want_norm)
{
    for (int i = threadIdx.x; i < 256; i += blockDim.x) {
        L.in[i] = (float)i / 255.0f;
        if (want_norm) {
#pragma unroll
            for (int c = 0; c < 3; ++c) L.nrm[c][i] = norm1((uint8_t)i, nc.mean[c], nc.std[c]);
        }
    }
}

// scipy 'reflect' (d c b a | a b c d | d c b a)
__device__ __forceinline__ int reflect_sym(int i, int n)
{
    if (n == 1) return 0;
    int period = 2 * n;
    i %= period; if (i < 0) i += period;
    return i < n ? i : period - 1 - i;
}
// OpenCV BORDER_REFLECT_101 (gfedcb|abcdefgh|gfedcba)
__device__ __forceinline__ int reflect_101(int i, int n)
{
    if (n == 1) return 0;
    while (i < 0 || i >= n) { if (i < 0) i = -i; else i = 2 * n - 2 - i; }
    return i;
}

// ------------------------------------------------------------------------------------ A7
// 4 pixels per lane: 12 B in (3 dwords), three float4 out (one per channel plane).
// (bodies take their block coordinates as arguments: the same code runs under its own launcher and, kind by kind, inside the ONE
// launch of awseg_weather_batch at the end of this file)
__device__ __forceinline__ void normalize_body(const uint8_t* __restrict__ imgs, int64_t hw, int64_t img, const norm_consts& nc,
                                               float* __restrict__ out, bool vec, int bx, int nbx)
{
    const uint8_t* src = imgs + img * hw * 3;
    float* dst = out + img * hw * 3;
    if (!vec) {
        // any H x W the reference's transform accepts (preprocessing.py:61-92): with hw % 4 != 0 (or unaligned bases) the
        // per-image bases are not dword / float4 aligned, so every pixel goes the scalar way — same operations, same bytes
        for (int64_t p = (int64_t)bx * kThreads + threadIdx.x; p < hw; p += (int64_t)nbx * kThreads)
            for (int c = 0; c < 3; ++c) dst[(int64_t)c * hw + p] = norm1(src[p * 3 + c], nc.mean[c], nc.std[c]);
        return;
    }
    const int64_t nquad = hw / 4;
    for (int64_t q = (int64_t)bx * kThreads + threadIdx.x; q < nquad; q += (int64_t)nbx * kThreads) {
        const uint32_t* p = reinterpret_cast<const uint32_t*>(src + q * 12);
        uint32_t w0 = p[0], w1 = p[1], w2 = p[2];
        uint8_t b[12];
#pragma unroll
        for (int k = 0; k < 4; ++k) { b[k] = (w0 >> (8 * k)) & 0xFF; b[4 + k] = (w1 >> (8 * k)) & 0xFF; b[8 + k] = (w2 >> (8 * k)) & 0xFF; }
#pragma unroll
        for (int c = 0; c < 3; ++c) {
            float4 v = make_float4(norm1(b[c], nc.mean[c], nc.std[c]), norm1(b[3 + c], nc.mean[c], nc.std[c]),
                                   norm1(b[6 + c], nc.mean[c], nc.std[c]), norm1(b[9 + c], nc.mean[c], nc.std[c]));
            *reinterpret_cast<float4*>(dst + (int64_t)c * hw + q * 4) = v;
        }
    }
}

__global__ __launch_bounds__(kThreads)
void normalize_kernel(const uint8_t* __restrict__ imgs, int64_t hw, const int32_t* __restrict__ sel,
                      norm_consts nc, float* __restrict__ out, bool vec)
{
    normalize_body(imgs, hw, sel ? sel[blockIdx.y] : blockIdx.y, nc, out, vec, blockIdx.x, gridDim.x);
}

// ------------------------------------------------------------------------- next #4 (style LUT)
// WeatherAugmentationPipeline._apply_style_transfer (PKG/data/loader.py:360-387) is a per-channel
// uint8 -> uint8 map (cv2.convertScaleAbs, then an optional gain on channel 2), so the host builds
// the 3 x 256 table with the reference's arithmetic and the device applies it: 12 B in, 12 B out
// per lane, table in LDS.  lut_of[b] < 0 leaves frame b unchanged.
__global__ __launch_bounds__(kThreads)
void lut3_kernel(const uint8_t* __restrict__ imgs, int64_t hw, const uint8_t* __restrict__ luts,
                 const int32_t* __restrict__ lut_of, uint8_t* __restrict__ out)
{
    __shared__ uint8_t s_lut[3 * 256];
    const int which = lut_of[blockIdx.y];
    const uint8_t* src = imgs + (int64_t)blockIdx.y * hw * 3;
    uint8_t* dst = out + (int64_t)blockIdx.y * hw * 3;
    if (which < 0 && src == dst) return;
    for (int i = threadIdx.x; i < 768; i += kThreads) s_lut[i] = which < 0 ? (uint8_t)(i & 255) : luts[(int64_t)which * 768 + i];
    __syncthreads();
    const int64_t nquad = hw / 4;                                  // 4 pixels = 3 dwords
    for (int64_t q = (int64_t)blockIdx.x * kThreads + threadIdx.x; q < nquad; q += (int64_t)gridDim.x * kThreads) {
        const uint32_t* p = reinterpret_cast<const uint32_t*>(src + q * 12);
        uint32_t w[3] = {p[0], p[1], p[2]}, r[3];
#pragma unroll
        for (int d = 0; d < 3; ++d) {
            uint32_t o = 0;
#pragma unroll
            for (int k = 0; k < 4; ++k) {
                const int c = (d * 4 + k) % 3;                     // byte index within the 12-byte group -> channel
                o |= (uint32_t)s_lut[c * 256 + ((w[d] >> (8 * k)) & 0xFF)] << (8 * k);
            }
            r[d] = o;
        }
        uint32_t* o4 = reinterpret_cast<uint32_t*>(dst + q * 12);
        o4[0] = r[0]; o4[1] = r[1]; o4[2] = r[2];
    }
    if (blockIdx.x == 0)
        for (int64_t i = nquad * 12 + threadIdx.x; i < hw * 3; i += kThreads) dst[i] = s_lut[(i % 3) * 256 + src[i]];
}

// ------------------------------------------------------------------------- A2 + A3 (fog)
// Tile TW x TH outputs; the 17-tap separable float64 Gaussian needs an 8-pixel halo, staged
// in LDS.  Pass order and summation order are scipy's (axis 0 first; centre tap, then the
// symmetric pairs from the outermost inwards), which is what makes depth bit-identical.
constexpr int FR = AWSEG_GAUSS_RADIUS;       // 8
constexpr int FTW = 64, FTH = 32;
constexpr int FIW = FTW + 2 * FR;            // 80 staged columns
constexpr int FIH = FTH + 2 * FR;            // 48 staged rows

struct gauss_taps { double w[2 * FR + 1]; };

template <bool PHILOX>
__device__ __forceinline__ double fog_noise_at(const double* __restrict__ noise, uint64_t seed, int gy, int gx, int W)
{
    if (!PHILOX) return noise[(int64_t)gy * W + gx];
    uint32_t r[4];
    uint64_t e = (uint64_t)gy * (uint64_t)W + (uint64_t)gx;
    awseg_philox::gen(seed, e, 0x0F06u, r);
    float n0, n1;
    awseg_box_muller(r[0], r[1], n0, n1);
    return (double)n0 * 10.0;                 // N(0, 10), preprocessing.py:239
}

// MODE 0: depth only (writes depth_out); MODE 1: fused depth + fog.
template <bool PHILOX, int MODE>
__global__ __launch_bounds__(kThreads)
void fog_kernel(const uint8_t* __restrict__ imgs, int H, int W, job_pack<awseg_fog_job> jobs, int job0,
                const double* __restrict__ noise_all, gauss_taps taps,
                uint8_t* __restrict__ out, float* __restrict__ norm_out, double* __restrict__ depth_out,
                norm_consts nc)
{
    __shared__ double s_in[FIH * FIW];        // 30 KB
    __shared__ double s_v[FTH * FIW];         // 20 KB
    const awseg_fog_job job = jobs.j[blockIdx.z];
    const int64_t hw = (int64_t)H * W;
    const double* noise = PHILOX ? nullptr : noise_all + (int64_t)(job0 + blockIdx.z) * hw;
    const int x0 = blockIdx.x * FTW, y0 = blockIdx.y * FTH;

    // phase 1: depth_base + noise for the tile and its halo (scipy reflect at the image border)
    for (int i = threadIdx.x; i < FIH * FIW; i += kThreads) {
        int ty = i / FIW, tx = i - ty * FIW;
        int gy = reflect_sym(y0 - FR + ty, H), gx = reflect_sym(x0 - FR + tx, W);
        double base = ((double)gy / (double)H) * 100.0;           // preprocessing.py:236
        s_in[i] = base + fog_noise_at<PHILOX>(noise, job.seed, gy, gx, W);
    }
    __syncthreads();
    // phase 2: axis-0 pass for rows of the tile, all staged columns
    for (int i = threadIdx.x; i < FTH * FIW; i += kThreads) {
        int ty = i / FIW, tx = i - ty * FIW;
        const double* c = s_in + (ty + FR) * FIW + tx;
        double o = c[0] * taps.w[FR];
#pragma unroll
        for (int j = -FR; j < 0; ++j) { double s = c[j * FIW] + c[-j * FIW]; double m = s * taps.w[FR + j]; o = o + m; }
        s_v[i] = o;
    }
    __syncthreads();
    // phase 3: axis-1 pass, 4 horizontally adjacent outputs per lane (window of 20 LDS reads),
    // then transmission / blend / quantise on the same 4 pixels (12 B in, 12 B out).
    const uint8_t* src = imgs + (int64_t)job.image * hw * 3;
    uint8_t* dst = out ? out + (int64_t)job.image * hw * 3 : nullptr;
    float* ndst = norm_out ? norm_out + (int64_t)job.image * hw * 3 : nullptr;
    double* ddst = depth_out ? depth_out + (int64_t)(job0 + blockIdx.z) * hw : nullptr;
    const double A32 = (double)(float)job.atmos;                   // A*ones_like(f32 image), :118
    for (int q = threadIdx.x; q < FTH * (FTW / 4); q += kThreads) {
        int ty = q / (FTW / 4), tq = q - ty * (FTW / 4);
        int gy = y0 + ty, gx = x0 + tq * 4;
        if (gy >= H || gx >= W) continue;
        double win[4 + 2 * FR];
        const double* row = s_v + ty * FIW + tq * 4;
#pragma unroll
        for (int k = 0; k < 4 + 2 * FR; ++k) win[k] = row[k];
        double depth[4];
#pragma unroll
        for (int k = 0; k < 4; ++k) {
            double o = win[k + FR] * taps.w[FR];
#pragma unroll
            for (int j = -FR; j < 0; ++j) { double s = win[k + FR + j] + win[k + FR - j]; double m = s * taps.w[FR + j]; o = o + m; }
            depth[k] = o > 1.0 ? o : 1.0;                          // np.maximum(depth, 1.0), :246
        }
        const int nvalid = (W - gx) < 4 ? (W - gx) : 4;
        const int64_t p = (int64_t)gy * W + gx;
        if (ddst) for (int k = 0; k < nvalid; ++k) ddst[p + k] = depth[k];
        if (MODE == 0) continue;
        uint8_t px[12], res[12];
        if (nvalid == 4 && (((uintptr_t)(src + p * 3)) & 3) == 0) {
            const uint32_t* s4 = reinterpret_cast<const uint32_t*>(src + p * 3);
            uint32_t w0 = s4[0], w1 = s4[1], w2 = s4[2];
#pragma unroll
            for (int k = 0; k < 4; ++k) { px[k] = (w0 >> (8 * k)) & 0xFF; px[4 + k] = (w1 >> (8 * k)) & 0xFF; px[8 + k] = (w2 >> (8 * k)) & 0xFF; }
        } else {
            for (int k = 0; k < nvalid * 3; ++k) px[k] = src[p * 3 + k];
        }
#pragma unroll
        for (int k = 0; k < 4; ++k) {
            double t = exp(-job.beta * depth[k]);                  // :117
            double omt = 1.0 - t;
            double hazeterm = A32 * omt;
#pragma unroll
            for (int c = 0; c < 3; ++c) {
                float v = (float)px[k * 3 + c] / 255.0f;           // :81
                double a = (double)v * t;
                res[k * 3 + c] = quant_f64(a + hazeterm);          // :120-123
            }
        }
        if (dst) {
            if (nvalid == 4 && (((uintptr_t)(dst + p * 3)) & 3) == 0) {
                uint32_t* d4 = reinterpret_cast<uint32_t*>(dst + p * 3);
#pragma unroll
                for (int w = 0; w < 3; ++w)
                    d4[w] = (uint32_t)res[4 * w] | ((uint32_t)res[4 * w + 1] << 8) | ((uint32_t)res[4 * w + 2] << 16) | ((uint32_t)res[4 * w + 3] << 24);
            } else {
                for (int k = 0; k < nvalid * 3; ++k) dst[p * 3 + k] = res[k];
            }
        }
        if (ndst) {
            for (int c = 0; c < 3; ++c)
                for (int k = 0; k < nvalid; ++k) ndst[(int64_t)c * hw + p + k] = norm1(res[k * 3 + c], nc.mean[c], nc.std[c]);
        }
    }
}

// Throughput-mode fog (noise == NULL): same algorithm with the noise drawn in-kernel (Philox, four
// normals per call) and the depth pipeline in float32 — parity is only defined for host-drawn
// noise (the float64 kernel above), so this variant trades the float64 ladder, whose ~110
// double-precision operations per pixel cap that kernel well below the HBM roof, for 4x cheaper
// arithmetic.  320 threads: 80 staged columns x 4 row segments in the vertical pass (8 outputs per
// lane from a 24-value register window), 4 adjacent outputs per lane in the horizontal pass.
constexpr int kFogFastThreads = 320;
struct gauss_taps_f32 { float w[2 * FR + 1]; };

__global__ __launch_bounds__(kFogFastThreads)
void fog_fast_kernel(const uint8_t* __restrict__ imgs, int H, int W, job_pack<awseg_fog_job> jobs, int job0,
                     gauss_taps_f32 taps, uint8_t* __restrict__ out, float* __restrict__ norm_out,
                     double* __restrict__ depth_out, norm_consts nc)
{
    __shared__ float s_in[FIH * FIW];         // 15 KB
    __shared__ float s_v[FTH * FIW];          // 10 KB
    __shared__ weather_lut L;
    lut_fill(L, nc, norm_out != nullptr);
    const awseg_fog_job job = jobs.j[blockIdx.z];
    const int64_t hw = (int64_t)H * W;
    const int x0 = blockIdx.x * FTW, y0 = blockIdx.y * FTH;
    const float inv_h = 100.0f / (float)H;
    const bool interior_x = (x0 - FR >= 0) && (x0 + FTW + FR <= W);
    // the source pixels of this thread's output quads (phase 3) are requested now, so the load latency hides behind the
    // noise synthesis and the two filter passes instead of sitting in front of the blend
    const uint8_t* src = imgs + (int64_t)job.image * hw * 3;
    uint32_t pre[2][3] = { { 0u, 0u, 0u }, { 0u, 0u, 0u } };
#pragma unroll
    for (int it = 0; it < 2; ++it) {
        const int q = threadIdx.x + it * kFogFastThreads;
        if (q < FTH * (FTW / 4)) {
            const int ty = q / (FTW / 4), tq = q - ty * (FTW / 4);
            const int gy = y0 + ty, gx = x0 + tq * 4;
            const int64_t p = (int64_t)gy * W + gx;
            if (gy < H && gx + 4 <= W && ((((uintptr_t)(src + p * 3)) & 3) == 0)) {
                const uint32_t* s4 = reinterpret_cast<const uint32_t*>(src + p * 3);
                pre[it][0] = s4[0]; pre[it][1] = s4[1]; pre[it][2] = s4[2];
            }
        }
    }

    // phase 1: (y/H)*100 + sigma-10 white noise for the tile and its halo; one Philox call = 8 samples = 8 columns.
    // The samples are NOT Gaussian: each is the difference of two independent bytes, scaled to unit variance (a
    // triangular law; 3 instructions instead of the ~16 of a Box-Muller value, four of them transcendental).  What the
    // transform uses is this field filtered by the 17 x 17 Gaussian below: a weighted sum with 1 / sum(w^2) ~ 50 effective
    // terms, whose covariance depends on the second moments only (identical) and whose marginals are Gaussian up to an
    // excess kurtosis of -0.6 / 50 (central limit theorem) — parity in this mode is in distribution anyway.
    const int64_t Wo = (W + 7) / 8;
    const float kTri = 0.009584116f;                                 // 1 / sqrt(2 * (256^2 - 1) / 12): unit variance
    if (interior_x) {
        for (int i = threadIdx.x; i < FIH * (FIW / 8); i += kFogFastThreads) {
            const int ty = i / (FIW / 8), to = i - ty * (FIW / 8);
            const int gy = reflect_sym(y0 - FR + ty, H), gx = x0 - FR + to * 8;      // x0 - 8 is a multiple of 8
            uint32_t r[4]; float n[8];
            awseg_philox::gen(job.seed, (uint64_t)gy * Wo + (gx >> 3), 0x0F06u, r);
#pragma unroll
            for (int k = 0; k < 4; ++k) {
                n[2 * k] = ((float)(r[k] & 0xFFu) - (float)((r[k] >> 8) & 0xFFu)) * kTri;          // v_cvt_f32_ubyte0/1
                n[2 * k + 1] = ((float)((r[k] >> 16) & 0xFFu) - (float)(r[k] >> 24)) * kTri;         // v_cvt_f32_ubyte2/3
            }
            const float base = (float)gy * inv_h;
            float* d = s_in + ty * FIW + to * 8;
            *reinterpret_cast<float4*>(d) = make_float4(base + 10.f * n[0], base + 10.f * n[1], base + 10.f * n[2], base + 10.f * n[3]);
            *reinterpret_cast<float4*>(d + 4) = make_float4(base + 10.f * n[4], base + 10.f * n[5], base + 10.f * n[6], base + 10.f * n[7]);
        }
    } else {
        for (int i = threadIdx.x; i < FIH * FIW; i += kFogFastThreads) {
            const int ty = i / FIW, tx = i - ty * FIW;
            const int gy = reflect_sym(y0 - FR + ty, H), gx = reflect_sym(x0 - FR + tx, W);
            uint32_t r[4]; float n0, n1;
            awseg_philox::gen(job.seed, (uint64_t)gy * Wo + (gx >> 3), 0x0F06u, r);
            const int sel = gx & 7;
            const uint32_t word = (sel >> 1) == 0 ? r[0] : ((sel >> 1) == 1 ? r[1] : ((sel >> 1) == 2 ? r[2] : r[3]));
            n0 = ((float)(word & 0xFFu) - (float)((word >> 8) & 0xFFu)) * kTri;
            n1 = ((float)((word >> 16) & 0xFFu) - (float)(word >> 24)) * kTri;
            s_in[i] = (float)gy * inv_h + 10.f * ((sel & 1) ? n1 : n0);
        }
    }
    __syncthreads();
    // phase 2: axis-0 pass; lane = (staged column, segment of 8 rows), 24-value register window
    {
        const int col = threadIdx.x % FIW, seg = threadIdx.x / FIW;     // 80 x 4
        float win[8 + 2 * FR];
#pragma unroll
        for (int k = 0; k < 8 + 2 * FR; ++k) win[k] = s_in[(seg * 8 + k) * FIW + col];
#pragma unroll
        for (int k = 0; k < 8; ++k) {
            float o = win[k + FR] * taps.w[FR];
#pragma unroll
            for (int j = 1; j <= FR; ++j) o = fmaf(win[k + FR - j] + win[k + FR + j], taps.w[FR + j], o);
            s_v[(seg * 8 + k) * FIW + col] = o;
        }
    }
    __syncthreads();
    // phase 3: axis-1 pass (4 adjacent outputs per lane) + transmission / blend / quantise
    uint8_t* dst = out ? out + (int64_t)job.image * hw * 3 : nullptr;
    float* ndst = norm_out ? norm_out + (int64_t)job.image * hw * 3 : nullptr;
    double* ddst = depth_out ? depth_out + (int64_t)(job0 + blockIdx.z) * hw : nullptr;
    const float beta = (float)job.beta, A32 = (float)job.atmos;
#pragma unroll
    for (int it = 0; it < 2; ++it) {
        const int q = threadIdx.x + it * kFogFastThreads;
        if (q >= FTH * (FTW / 4)) break;
        const int ty = q / (FTW / 4), tq = q - ty * (FTW / 4);
        const int gy = y0 + ty, gx = x0 + tq * 4;
        if (gy >= H || gx >= W) continue;
        float win[4 + 2 * FR];
        const float* row = s_v + ty * FIW + tq * 4;
#pragma unroll
        for (int k = 0; k < (4 + 2 * FR) / 4; ++k) {
            float4 t = *reinterpret_cast<const float4*>(row + 4 * k);
            win[4 * k] = t.x; win[4 * k + 1] = t.y; win[4 * k + 2] = t.z; win[4 * k + 3] = t.w;
        }
        float depth[4];
#pragma unroll
        for (int k = 0; k < 4; ++k) {
            float o = win[k + FR] * taps.w[FR];
#pragma unroll
            for (int j = 1; j <= FR; ++j) o = fmaf(win[k + FR - j] + win[k + FR + j], taps.w[FR + j], o);
            depth[k] = o > 1.0f ? o : 1.0f;
        }
        const int nvalid = (W - gx) < 4 ? (W - gx) : 4;
        const int64_t p = (int64_t)gy * W + gx;
        if (ddst) for (int k = 0; k < nvalid; ++k) ddst[p + k] = (double)depth[k];
        uint8_t px[12] = { 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0 }, res[12];
        const bool vec = (nvalid == 4) && ((((uintptr_t)(src + p * 3)) & 3) == 0) && (!dst || (((uintptr_t)(dst + p * 3)) & 3) == 0);
        if (vec) {
            const uint32_t w0 = pre[it][0], w1 = pre[it][1], w2 = pre[it][2];
#pragma unroll
            for (int k = 0; k < 4; ++k) { px[k] = (w0 >> (8 * k)) & 0xFF; px[4 + k] = (w1 >> (8 * k)) & 0xFF; px[8 + k] = (w2 >> (8 * k)) & 0xFF; }
        } else {
            for (int k = 0; k < nvalid * 3; ++k) px[k] = src[p * 3 + k];
        }
        // blend and quantise in byte units: trunc(clip(px/255 * t + A (1 - t), 0, 1) * 255) = trunc(clip(px * t + 255 A (1 - t), 0, 255)) up to
        // float32 rounding — one v_cvt_f32_ubyteN + one FMA + one v_med3 + one conversion per value, no table
#pragma unroll
        for (int k = 0; k < 4; ++k) {
            const float t = __expf(-beta * depth[k]);
            const float hz = (255.0f * A32) * (1.0f - t);
#pragma unroll
            for (int c = 0; c < 3; ++c)
                res[k * 3 + c] = (uint8_t)(int)__builtin_amdgcn_fmed3f(fmaf((float)px[k * 3 + c], t, hz), 0.f, 255.f);
        }
        if (dst) {
            if (vec) {
                uint32_t* d4 = reinterpret_cast<uint32_t*>(dst + p * 3);
#pragma unroll
                for (int w = 0; w < 3; ++w)
                    d4[w] = (uint32_t)res[4 * w] | ((uint32_t)res[4 * w + 1] << 8) | ((uint32_t)res[4 * w + 2] << 16) | ((uint32_t)res[4 * w + 3] << 24);
            } else {
                for (int k = 0; k < nvalid * 3; ++k) dst[p * 3 + k] = res[k];
            }
        }
        if (ndst) {
            if (nvalid == 4 && (p & 3) == 0 && (hw & 3) == 0) {
#pragma unroll
                for (int c = 0; c < 3; ++c)
                    *reinterpret_cast<float4*>(ndst + (int64_t)c * hw + p) =
                        make_float4(L.nrm[c][res[c]], L.nrm[c][res[3 + c]], L.nrm[c][res[6 + c]], L.nrm[c][res[9 + c]]);
            } else {
                for (int c = 0; c < 3; ++c)
                    for (int k = 0; k < nvalid; ++k) ndst[(int64_t)c * hw + p + k] = L.nrm[c][res[k * 3 + c]];
            }
        }
    }
}

// Throughput-mode fog as a STRIP kernel (W % 4 == 0): a lane owns 4 adjacent pixels and walks the rows of a strip; a wave is
// 64 adjacent pixel quads of which the inner 60 produce output (two halo quads per side feed the horizontal taps).
//   noise        one Philox call per (row pair, quad): its 16 bytes are the 2 x 4 byte-difference samples of the lane's four
//                pixels in two consecutive rows — drawn once per sample of the strip (+ the 16 halo rows), not once per tile.
//   horizontal   17 taps over the 20-value window the lane assembles from its own quad and its neighbours' (wave shifts by one
//                lane: v_mov_b32 dpp wave_shr / wave_shl, no LDS); scipy's 'reflect' at the image border = the reflected quad's
//                samples in reverse order, chosen per lane.
//   vertical     17 taps over the last 17 horizontal results, kept in a per-lane LDS column (one 16-byte write, 17 reads per
//                row at a dynamic slot: no register ring to rotate, no 17-fold unrolled body).
//   blend        in byte units as in the tile kernel.
// The depth pipeline is float32 and the order of the two passes is swapped against the tile kernel (separable filter); parity
// in this mode is in distribution (tests: the depth field against scipy on the SAME Philox samples, and its moments).
__device__ __forceinline__ float dpp_from_prev(float v)     // lane i <- lane i - 1 (lane 0 <- 0)
{
    return __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), 0x138, 0xF, 0xF, true));
}
__device__ __forceinline__ float dpp_from_next(float v)     // lane i <- lane i + 1 (lane 63 <- 0)
{
    return __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), 0x130, 0xF, 0xF, true));
}
constexpr int kFogStripQuads = 60;                           // output quads per wave
constexpr int kFogRing = 2 * FR + 1;                         // 17

__device__ __forceinline__ void fog_strip_body(const uint8_t* __restrict__ imgs, int H, int W, const awseg_fog_job job, int job_slot,
                                               const gauss_taps_f32& taps, uint8_t* __restrict__ out, float* __restrict__ norm_out,
                                               double* __restrict__ depth_out, const norm_consts& nc, int rows_per_strip,
                                               weather_lut& L, float4* s_ring, int bx, int by)
{
    lut_fill(L, nc, norm_out != nullptr);
    __syncthreads();
    const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
    const int Wq = W >> 2;
    const int q = bx * kFogStripQuads - 2 + lane;            // this lane's pixel quad (may lie outside the row)
    const int y0 = (by * 4 + wv) * rows_per_strip;
    if (y0 >= H) return;                                     // wave-uniform
    const int rows_out = (H - y0) < rows_per_strip ? (H - y0) : rows_per_strip;
    // scipy 'reflect' (d c b a | a b c d | d c b a) on pixels = the mirrored quad, its four samples in reverse order
    int qs = q; bool rev = false;
    if (q < 0) { qs = -q - 1; rev = true; } else if (q >= Wq) { qs = 2 * Wq - 1 - q; rev = true; }
    qs = qs < 0 ? 0 : (qs > Wq - 1 ? Wq - 1 : qs);
    const bool own = lane >= 2 && lane < 62 && q < Wq;
    const int gx = q * 4;
    const int64_t hw = (int64_t)H * W;
    const uint8_t* src = imgs + (int64_t)job.image * hw * 3;
    uint8_t* dst = out ? out + (int64_t)job.image * hw * 3 : nullptr;
    float* ndst = norm_out ? norm_out + (int64_t)job.image * hw * 3 : nullptr;
    double* ddst = depth_out ? depth_out + (int64_t)job_slot * hw : nullptr;
    const float beta = (float)job.beta, A255 = 255.0f * (float)job.atmos;
    const float inv_h = 100.0f / (float)H;
    const float kAmp = 10.0f * 0.009584116f;                 // sigma 10 x 1 / sqrt(2 (256^2 - 1) / 12): unit-variance byte differences
    float4* col = s_ring + threadIdx.x;                      // this lane's column: slot s at col[s * 256]
    int pair = -1; uint32_t r[4] = { 0u, 0u, 0u, 0u };
    for (int i = 0; i < rows_out + 2 * FR; ++i) {
        const int gy = reflect_sym(y0 - FR + i, H);
        uint32_t px0 = 0u, px1 = 0u, px2 = 0u;               // the output row this iteration closes: its source pixels, requested early
        const int oy = y0 + i - 2 * FR;
        if (i >= 2 * FR && own) {
            const uint32_t* s4 = reinterpret_cast<const uint32_t*>(src + ((int64_t)oy * W + gx) * 3);
            px0 = s4[0]; px1 = s4[1]; px2 = s4[2];
        }
        if ((gy >> 1) != pair) {                             // wave-uniform
            pair = gy >> 1;
            awseg_philox::gen(job.seed, (uint64_t)pair * Wq + qs, 0x0F07u, r);
        }
        const uint32_t w0 = (gy & 1) ? r[2] : r[0], w1 = (gy & 1) ? r[3] : r[1];
        float n0 = (float)(w0 & 0xFFu) - (float)((w0 >> 8) & 0xFFu), n1 = (float)((w0 >> 16) & 0xFFu) - (float)(w0 >> 24);
        float n2 = (float)(w1 & 0xFFu) - (float)((w1 >> 8) & 0xFFu), n3 = (float)((w1 >> 16) & 0xFFu) - (float)(w1 >> 24);
        const float base = (float)gy * inv_h;
        float win[20];
        win[8] = fmaf(kAmp, rev ? n3 : n0, base); win[9] = fmaf(kAmp, rev ? n2 : n1, base);
        win[10] = fmaf(kAmp, rev ? n1 : n2, base); win[11] = fmaf(kAmp, rev ? n0 : n3, base);
#pragma unroll
        for (int k = 0; k < 4; ++k) {
            win[4 + k] = dpp_from_prev(win[8 + k]); win[k] = dpp_from_prev(win[4 + k]);
            win[12 + k] = dpp_from_next(win[8 + k]); win[16 + k] = dpp_from_next(win[12 + k]);
        }
        float4 hres;
        {
            float o[4];
#pragma unroll
            for (int k = 0; k < 4; ++k) {
                float a = win[k + FR] * taps.w[FR];
#pragma unroll
                for (int j = 1; j <= FR; ++j) a = fmaf(win[k + FR - j] + win[k + FR + j], taps.w[FR + j], a);
                o[k] = a;
            }
            hres = make_float4(o[0], o[1], o[2], o[3]);
        }
        col[(i % kFogRing) * 256] = hres;                    // only this lane reads its column: no barrier
        if (i < 2 * FR) continue;
        // vertical taps over input rows i-16 .. i (centre i-8)
        // (an f16 ring of the noise part, read back through v_fma_mix_f32, halves the LDS footprint but the grid — not LDS —
        // bounds residency at this strip height, and 68 mixed FMAs replace 34 packed float32 ones: measured 93 us against 79)
        float d[4];
        {
            const int c = (i - FR) % kFogRing;
            const float4 ctr = col[c * 256];
            d[0] = ctr.x * taps.w[FR]; d[1] = ctr.y * taps.w[FR]; d[2] = ctr.z * taps.w[FR]; d[3] = ctr.w * taps.w[FR];
#pragma unroll
            for (int j = 1; j <= FR; ++j) {
                int up = c - j; up += up < 0 ? kFogRing : 0;
                int dn = c + j; dn -= dn >= kFogRing ? kFogRing : 0;
                const float4 a = col[up * 256], b = col[dn * 256];
                d[0] = fmaf(a.x + b.x, taps.w[FR + j], d[0]); d[1] = fmaf(a.y + b.y, taps.w[FR + j], d[1]);
                d[2] = fmaf(a.z + b.z, taps.w[FR + j], d[2]); d[3] = fmaf(a.w + b.w, taps.w[FR + j], d[3]);
            }
        }
        if (!own) continue;
#pragma unroll
        for (int k = 0; k < 4; ++k) d[k] = d[k] > 1.0f ? d[k] : 1.0f;                 // np.maximum(depth, 1.0), :246
        const int64_t p = (int64_t)oy * W + gx;
        if (ddst) { for (int k = 0; k < 4; ++k) ddst[p + k] = (double)d[k]; }
        uint8_t res[12];
#pragma unroll
        for (int k = 0; k < 4; ++k) {
            const float t = __expf(-beta * d[k]);
            const float hz = A255 * (1.0f - t);
#pragma unroll
            for (int c = 0; c < 3; ++c) {
                const int e = k * 3 + c;
                const uint32_t word = (e >> 2) == 0 ? px0 : ((e >> 2) == 1 ? px1 : px2);
                res[e] = (uint8_t)(int)__builtin_amdgcn_fmed3f(fmaf((float)((word >> (8 * (e & 3))) & 0xFFu), t, hz), 0.f, 255.f);
            }
        }
        if (dst) {
            uint32_t* d4 = reinterpret_cast<uint32_t*>(dst + p * 3);
#pragma unroll
            for (int wq = 0; wq < 3; ++wq)
                d4[wq] = (uint32_t)res[4 * wq] | ((uint32_t)res[4 * wq + 1] << 8) | ((uint32_t)res[4 * wq + 2] << 16) | ((uint32_t)res[4 * wq + 3] << 24);
        }
        if (ndst) {
#pragma unroll
            for (int c = 0; c < 3; ++c)
                *reinterpret_cast<float4*>(ndst + (int64_t)c * hw + p) =
                    make_float4(L.nrm[c][res[c]], L.nrm[c][res[3 + c]], L.nrm[c][res[6 + c]], L.nrm[c][res[9 + c]]);
        }
    }
}

__global__ __launch_bounds__(256)
void fog_strip_kernel(const uint8_t* __restrict__ imgs, int H, int W, job_pack<awseg_fog_job> jobs, int job0,
                      gauss_taps_f32 taps, uint8_t* __restrict__ out, float* __restrict__ norm_out,
                      double* __restrict__ depth_out, norm_consts nc, int rows_per_strip)
{
    __shared__ weather_lut L;
    extern __shared__ float4 s_ring[];                       // [17][256]
    fog_strip_body(imgs, H, W, jobs.j[blockIdx.z], job0 + blockIdx.z, taps, out, norm_out, depth_out, nc, rows_per_strip, L, s_ring, blockIdx.x, blockIdx.y);
}

// fog from a caller-provided depth map (the two-step form of the reference).
__global__ __launch_bounds__(kThreads)
void fog_apply_kernel(const uint8_t* __restrict__ imgs, int64_t hw, job_pack<awseg_fog_job> jobs, int job0,
                      const double* __restrict__ depth_all, uint8_t* __restrict__ out, float* __restrict__ norm_out,
                      norm_consts nc)
{
    const awseg_fog_job job = jobs.j[blockIdx.y];
    const uint8_t* src = imgs + (int64_t)job.image * hw * 3;
    const double* depth = depth_all + (int64_t)(job0 + blockIdx.y) * hw;
    uint8_t* dst = out ? out + (int64_t)job.image * hw * 3 : nullptr;
    float* ndst = norm_out ? norm_out + (int64_t)job.image * hw * 3 : nullptr;
    const double A32 = (double)(float)job.atmos;
    for (int64_t p = (int64_t)blockIdx.x * kThreads + threadIdx.x; p < hw; p += (int64_t)gridDim.x * kThreads) {
        double t = exp(-job.beta * depth[p]);
        double omt = 1.0 - t;
        double hz = A32 * omt;
#pragma unroll
        for (int c = 0; c < 3; ++c) {
            float v = (float)src[p * 3 + c] / 255.0f;
            double a = (double)v * t;
            uint8_t q = quant_f64(a + hz);
            if (dst) dst[p * 3 + c] = q;
            if (ndst) ndst[(int64_t)c * hw + p] = norm1(q, nc.mean[c], nc.std[c]);
        }
    }
}

// ------------------------------------------------------------------------------ A6 night
// 4 pixels per lane (12 B in as three dwords, 12 B and/or three float4 out).  Throughput mode draws
// its 12 normals from three Philox calls; parity mode reads the host's float64 draws.
template <bool PHILOX>
__device__ __forceinline__ void night_body(const uint8_t* __restrict__ imgs, int64_t hw, const awseg_night_job job, int job_slot,
                                           const double* __restrict__ noise_all, float g0, float g1, float g2,
                                           uint8_t* __restrict__ out, float* __restrict__ norm_out, const norm_consts& nc, bool vec,
                                           weather_lut& L, int bx, int nbx)
{
    lut_fill(L, nc, norm_out != nullptr);
    __syncthreads();
    const uint8_t* src = imgs + (int64_t)job.image * hw * 3;
    const double* noise = PHILOX ? nullptr : noise_all + (int64_t)job_slot * hw * 3;
    uint8_t* dst = out ? out + (int64_t)job.image * hw * 3 : nullptr;
    float* ndst = norm_out ? norm_out + (int64_t)job.image * hw * 3 : nullptr;
    const float bf = (float)job.brightness;                       // Python float x f32 array -> f32, :213
    const float gains[3] = { g0, g1, g2 };
    const double sigma = 5.0 / 255.0;                             // :222
    const double ni = job.intensity;
    const int64_t nquad = (hw + 3) / 4;
    for (int64_t q = (int64_t)bx * kThreads + threadIdx.x; q < nquad; q += (int64_t)nbx * kThreads) {
        const int64_t p = q * 4;
        const int nvalid = (hw - p) < 4 ? (int)(hw - p) : 4;
        const bool v4 = nvalid == 4 && vec;                       // vec: per-image bases are dword (float4, double2) aligned
        uint8_t px[12] = { 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0 }, res[12];
        if (v4) {
            const uint32_t* s4 = reinterpret_cast<const uint32_t*>(src + p * 3);
            uint32_t w0 = s4[0], w1 = s4[1], w2 = s4[2];
#pragma unroll
            for (int k = 0; k < 4; ++k) { px[k] = (w0 >> (8 * k)) & 0xFF; px[4 + k] = (w1 >> (8 * k)) & 0xFF; px[8 + k] = (w2 >> (8 * k)) & 0xFF; }
        } else {
            for (int k = 0; k < nvalid * 3; ++k) px[k] = src[p * 3 + k];
        }
        double nz[12];
        if (PHILOX) {
            // throughput mode: 12 normals from 6 of the 8 words of two Philox calls, all float32
            // (parity is only defined for host-drawn noise)
            float nf[12];
            uint32_t r[8];
            awseg_philox::gen(job.seed, (uint64_t)q * 2, 0x0A17u, r);
            awseg_philox::gen(job.seed, (uint64_t)q * 2 + 1, 0x0A17u, r + 4);
#pragma unroll
            for (int j = 0; j < 6; ++j) awseg_box_muller16(r[j], nf[2 * j], nf[2 * j + 1]);
            // byte units, as in the throughput-mode fog: trunc(clip(px/255 * b * g + n, 0, 1) * 255) = trunc(clip(px * (b g) + 255 n, 0, 255))
            const float amp = (float)(sigma * ni * 0.5 * 255.0);
            const float kg[3] = { bf * g0, bf * g1, bf * g2 };
#pragma unroll
            for (int k = 0; k < 12; ++k)
                res[k] = (uint8_t)(int)__builtin_amdgcn_fmed3f(fmaf(nf[k], amp, (float)px[k] * kg[k % 3]), 0.f, 255.f);
        } else if (v4) {
            const double2* n2 = reinterpret_cast<const double2*>(noise + p * 3);
#pragma unroll
            for (int k = 0; k < 6; ++k) { double2 t = n2[k]; nz[2 * k] = t.x; nz[2 * k + 1] = t.y; }
        } else {
            for (int k = 0; k < 12; ++k) nz[k] = k < nvalid * 3 ? noise[p * 3 + k] : 0.0;
        }
        if (!PHILOX) {
#pragma unroll
            for (int k = 0; k < 12; ++k) {
                float v = L.in[px[k]];
                v = v * bf;
                v = v * gains[k % 3];                              // :217-219
                double n = nz[k] * ni;
                n = n * 0.5;                                       // :223
                res[k] = quant_f64((double)v + n);
            }
        }
        if (dst) {
            if (v4) {
                uint32_t* d4 = reinterpret_cast<uint32_t*>(dst + p * 3);
#pragma unroll
                for (int w = 0; w < 3; ++w)
                    d4[w] = (uint32_t)res[4 * w] | ((uint32_t)res[4 * w + 1] << 8) | ((uint32_t)res[4 * w + 2] << 16) | ((uint32_t)res[4 * w + 3] << 24);
            } else {
                for (int k = 0; k < nvalid * 3; ++k) dst[p * 3 + k] = res[k];
            }
        }
        if (ndst) {
            if (v4 && (hw & 3) == 0) {
#pragma unroll
                for (int c = 0; c < 3; ++c)
                    *reinterpret_cast<float4*>(ndst + (int64_t)c * hw + p) =
                        make_float4(L.nrm[c][res[c]], L.nrm[c][res[3 + c]], L.nrm[c][res[6 + c]], L.nrm[c][res[9 + c]]);
            } else {
                for (int c = 0; c < 3; ++c)
                    for (int k = 0; k < nvalid; ++k) ndst[(int64_t)c * hw + p + k] = L.nrm[c][res[k * 3 + c]];
            }
        }
    }
}

template <bool PHILOX>
__global__ __launch_bounds__(kThreads)
void night_kernel(const uint8_t* __restrict__ imgs, int64_t hw, job_pack<awseg_night_job> jobs, int job0,
                  const double* __restrict__ noise_all, float g0, float g1, float g2,
                  uint8_t* __restrict__ out, float* __restrict__ norm_out, norm_consts nc, bool vec)
{
    __shared__ weather_lut L;
    night_body<PHILOX>(imgs, hw, jobs.j[blockIdx.y], job0 + blockIdx.y, noise_all, g0, g1, g2, out, norm_out, nc, vec, L, blockIdx.x, gridDim.x);
}

// ---------------------------------------------------------------- rasteriser (A4 / A5)
// Integer restatement of OpenCV's drawing primitives (PARITY UNPINNED, see the oracle header),
// producing the same pixel sets as the scanline code in oracle/awseg_oracle.c.  A primitive is
// rasterised by ONE WAVE into an LDS coverage mask (tile + blur halo, clipped to the image): the
// inherently serial parts (midpoint-circle recurrence, polygon edge walk) run wave-uniformly, the
// span fills and the fixed-point line steps are spread over the 64 lanes.
struct tile_mask {
    uint8_t* m; int x0, y0, w, h;   // rect origin / size in image coordinates
    int W, H;                       // image size (OpenCV clips to the image first)
    uint32_t* bits; int wd;         // frame-wide 1-bit coverage map (wd dwords per row) instead of the byte rect, or NULL
};
__device__ __forceinline__ void m_set(const tile_mask& k, int rx, int ry)
{
    if (k.bits) atomicOr(&k.bits[(int64_t)ry * k.wd + (rx >> 5)], 1u << (rx & 31));
    else k.m[ry * k.w + rx] = 1;
}
__device__ __forceinline__ void m_hline(const tile_mask& k, int y, int xa, int xb, int lane)
{
    if (y < 0 || y >= k.H) return;
    if (xa < 0) xa = 0;
    if (xb >= k.W) xb = k.W - 1;
    const int ry = y - k.y0;
    if (ry < 0 || ry >= k.h) return;
    int a = xa - k.x0, b = xb - k.x0;
    if (a < 0) a = 0;
    if (b >= k.w) b = k.w - 1;
    for (int x = a + lane; x <= b; x += 64) m_set(k, x, ry);
}
__device__ __forceinline__ void m_point(const tile_mask& k, int x, int y)
{
    if (x < 0 || x >= k.W || y < 0 || y >= k.H) return;
    const int rx = x - k.x0, ry = y - k.y0;
    if (rx < 0 || rx >= k.w || ry < 0 || ry >= k.h) return;
    m_set(k, rx, ry);
}
// cv::Line (LineIterator, 8-connected, left to right): closed form per step -> one step per lane.
__device__ void m_line_thin(const tile_mask& k, int x0, int y0, int x1, int y1, int lane)
{
    if (x1 < x0) { int t = x0; x0 = x1; x1 = t; t = y0; y0 = y1; y1 = t; }
    int dx = x1 - x0, dy = y1 - y0;
    const int sy = dy < 0 ? -1 : 1;
    if (dy < 0) dy = -dy;
    if (dx >= dy) {
        if (dx == 0) { if (lane == 0) m_point(k, x0, y0); return; }
        for (int i = lane; i <= dx; i += 64) m_point(k, x0 + i, y0 + sy * ((2 * dy * i + dx - 1) / (2 * dx)));   // |values| < 2^31 for any image
    } else {
        for (int i = lane; i <= dy; i += 64) m_point(k, x0 + (2 * dx * i + dy - 1) / (2 * dy), y0 + sy * i);
    }
}
// cv::Circle(fill): the midpoint recurrence is wave-uniform, each of its spans is filled by the lanes.
__device__ void m_disc(const tile_mask& k, int cx, int cy, int r, int lane)
{
    int err = 0, dx = r, dy = 0, plus = 1, minus = (r << 1) - 1;
    while (dx >= dy) {
        m_hline(k, cy - dy, cx - dx, cx + dx, lane);
        m_hline(k, cy + dy, cx - dx, cx + dx, lane);
        m_hline(k, cy - dx, cx - dy, cx + dy, lane);
        m_hline(k, cy + dx, cx - dy, cx + dy, lane);
        dy++;
        err += plus; plus += 2;
        int mask = (err <= 0) - 1;
        err -= minus & mask;
        dx += mask;
        minus -= mask & 2;
    }
}
constexpr int XY_SHIFT = 16;
constexpr int XY_ONE = 1 << XY_SHIFT;
// Truncating int64 division for |a| < 2^52: the correctly rounded double quotient truncates to the
// exact integer quotient (a non-integer quotient is >= 1/b away from an integer, more than its
// ulp), and costs a fraction of the 64-bit integer division sequence.
__device__ __forceinline__ int64_t div_trunc(int64_t a, int64_t b) { return (int64_t)((double)a / (double)b); }
// cv::Line2 (16.16 fixed-point DDA): position after j steps is start + j*step -> one step per lane.
__device__ void m_line2(const tile_mask& k, int64_t x1, int64_t y1, int64_t x2, int64_t y2, int lane)
{
    int64_t dx = x2 - x1, dy = y2 - y1;
    const int64_t ax = dx < 0 ? -dx : dx, ay = dy < 0 ? -dy : dy;
    if (lane == 0) m_point(k, (int)((x2 + (XY_ONE >> 1)) >> XY_SHIFT), (int)((y2 + (XY_ONE >> 1)) >> XY_SHIFT));
    if (ax > ay) {
        if (dx < 0) { int64_t t = x1; x1 = x2; x2 = t; t = y1; y1 = y2; y2 = t; dy = -dy; }
        const int64_t y_step = div_trunc(dy * XY_ONE, ax | 1);
        const int ecount = (int)((x2 - x1) >> XY_SHIFT);
        x1 += XY_ONE >> 1; y1 += XY_ONE >> 1;
        const int64_t xb = x1 >> XY_SHIFT;
        for (int j = lane; j <= ecount; j += 64) m_point(k, (int)(xb + j), (int)((y1 + j * y_step) >> XY_SHIFT));
    } else {
        if (dy < 0) { int64_t t = x1; x1 = x2; x2 = t; t = y1; y1 = y2; y2 = t; dx = -dx; }
        const int64_t x_step = div_trunc(dx * XY_ONE, ay | 1);
        const int ecount = (int)((y2 - y1) >> XY_SHIFT);
        x1 += XY_ONE >> 1; y1 += XY_ONE >> 1;
        const int64_t yb = y1 >> XY_SHIFT;
        for (int j = lane; j <= ecount; j += 64) m_point(k, (int)((x1 + j * x_step) >> XY_SHIFT), (int)(yb + j));
    }
}
// cv::FillConvexPoly for the 4-point thick-line body: wave-uniform edge walk, lane-parallel spans.
// Vertices are picked with select chains and the two edge states are plain scalars: a runtime-indexed
// private array would live in scratch memory (hundreds of cycles per access).
__device__ __forceinline__ int64_t sel4(const int64_t* v, int i)
{
    return i == 0 ? v[0] : (i == 1 ? v[1] : (i == 2 ? v[2] : v[3]));
}
__device__ void m_fill_convex4(const tile_mask& k, const int64_t* vx, const int64_t* vy, int lane)
{
    const int npts = 4;
    const int64_t delta = XY_ONE >> 1;
    int imin = 0, edges = npts;
    int64_t xmin = vx[0], xmax = vx[0], ymin = vy[0], ymax = vy[0];
    int64_t px = vx[3], py = vy[3];
#pragma unroll
    for (int i = 0; i < npts; ++i) {
        if (vy[i] < ymin) { ymin = vy[i]; imin = i; }
        if (vy[i] > ymax) ymax = vy[i];
        if (vx[i] > xmax) xmax = vx[i];
        if (vx[i] < xmin) xmin = vx[i];
        m_line2(k, px, py, vx[i], vy[i], lane);
        px = vx[i]; py = vy[i];
    }
    xmin = (xmin + delta) >> XY_SHIFT; xmax = (xmax + delta) >> XY_SHIFT;
    ymin = (ymin + delta) >> XY_SHIFT; ymax = (ymax + delta) >> XY_SHIFT;
    if ((int)xmax < 0 || (int)ymax < 0 || (int)xmin >= k.W || (int)ymin >= k.H) return;
    if (ymax > k.H - 1) ymax = k.H - 1;
    int y = (int)ymin;
    int idxA = imin, idxB = imin, yeA = y, yeB = y;          // edge A walks +1 through the vertices, edge B -1
    int64_t xA = -XY_ONE, xB = -XY_ONE, dxA = 0, dxB = 0;
    do {
#pragma unroll
        for (int i = 0; i < 2; ++i) {
            int& e_idx = i == 0 ? idxA : idxB;
            int& e_ye = i == 0 ? yeA : yeB;
            int64_t& e_x = i == 0 ? xA : xB;
            int64_t& e_dx = i == 0 ? dxA : dxB;
            const int di = i == 0 ? 1 : npts - 1;
            if (y >= e_ye) {
                int idx0 = e_idx;
                int idx = idx0 + di; if (idx >= npts) idx -= npts;
                for (; edges-- > 0;) {
                    const int ty = (int)((sel4(vy, idx) + delta) >> XY_SHIFT);
                    if (ty > y) {
                        const int64_t xs = sel4(vx, idx0), xe = sel4(vx, idx);
                        e_ye = ty;
                        e_dx = div_trunc((xe - xs) * 2 + (ty - y), 2 * (ty - y));
                        e_x = xs;
                        e_idx = idx;
                        break;
                    }
                    idx0 = idx; idx += di; if (idx >= npts) idx -= npts;
                }
            }
        }
        if (edges < 0) break;
        if (y >= 0) {
            const int64_t xl = xA > xB ? xB : xA, xr = xA > xB ? xA : xB;
            const int xx1 = (int)((xl + (XY_ONE >> 1)) >> XY_SHIFT);
            const int xx2 = (int)((xr + (XY_ONE >> 1)) >> XY_SHIFT);
            if (xx2 >= 0 && xx1 < k.W) m_hline(k, y, xx1, xx2, lane);
        }
        xA += dxA;
        xB += dxB;
    } while (++y <= (int)ymax);
}
// The same polygon fill with the SCANLINES spread over the lanes (coverage pre-pass: one wave owns the whole primitive).
// The edge walk above changes state only at the scanlines where an edge ends; between two such events an edge's abscissa
// is x_start + (y - y_start) * dx in exact integer arithmetic — what the serial walk reaches by adding dx once per scanline.
// Every lane runs the (wave-uniform, at most npts + 1 iterations) event loop with the serial code's own transition
// statements, and draws its scanline when the loop passes over it.  A lane writes its whole span itself.
__device__ __forceinline__ void m_hline_lane(const tile_mask& k, int y, int xa, int xb)
{
    if (y < 0 || y >= k.H) return;
    if (xa < 0) xa = 0;
    if (xb >= k.W) xb = k.W - 1;
    const int ry = y - k.y0;
    if (ry < 0 || ry >= k.h) return;
    int a = xa - k.x0, b = xb - k.x0;
    if (a < 0) a = 0;
    if (b >= k.w) b = k.w - 1;
    if (k.bits) {
        for (int dw = a >> 5; dw <= (b >> 5); ++dw) {
            const int lo = a > dw * 32 ? a - dw * 32 : 0, hi = b < dw * 32 + 31 ? b - dw * 32 : 31;
            if (hi < lo) continue;
            const uint32_t m = (hi - lo == 31) ? 0xFFFFFFFFu : (((1u << (hi - lo + 1)) - 1u) << lo);
            atomicOr(&k.bits[(int64_t)ry * k.wd + dw], m);
        }
    } else {
        for (int x = a; x <= b; ++x) k.m[ry * k.w + x] = 1;
    }
}
__device__ void m_fill_convex4_lanes(const tile_mask& k, const int64_t* vx, const int64_t* vy, int lane)
{
    const int npts = 4;
    const int64_t delta = XY_ONE >> 1;
    int imin = 0, edges = npts;
    int64_t xmin = vx[0], xmax = vx[0], ymin = vy[0], ymax = vy[0];
    int64_t px = vx[3], py = vy[3];
#pragma unroll
    for (int i = 0; i < npts; ++i) {
        if (vy[i] < ymin) { ymin = vy[i]; imin = i; }
        if (vy[i] > ymax) ymax = vy[i];
        if (vx[i] > xmax) xmax = vx[i];
        if (vx[i] < xmin) xmin = vx[i];
        m_line2(k, px, py, vx[i], vy[i], lane);
        px = vx[i]; py = vy[i];
    }
    xmin = (xmin + delta) >> XY_SHIFT; xmax = (xmax + delta) >> XY_SHIFT;
    ymin = (ymin + delta) >> XY_SHIFT; ymax = (ymax + delta) >> XY_SHIFT;
    if ((int)xmax < 0 || (int)ymax < 0 || (int)xmin >= k.W || (int)ymin >= k.H) return;
    if (ymax > k.H - 1) ymax = k.H - 1;
    int y = (int)ymin;
    int idxA = imin, idxB = imin, yeA = y, yeB = y;
    int ysA = y, ysB = y;                                        // scanline at which the current edge of a side was entered
    int64_t xsA = -XY_ONE, xsB = -XY_ONE, dxA = 0, dxB = 0;      // abscissa there, slope per scanline
    while (y <= (int)ymax) {
#pragma unroll
        for (int i = 0; i < 2; ++i) {
            int& e_idx = i == 0 ? idxA : idxB;
            int& e_ye = i == 0 ? yeA : yeB;
            int& e_ys = i == 0 ? ysA : ysB;
            int64_t& e_x = i == 0 ? xsA : xsB;
            int64_t& e_dx = i == 0 ? dxA : dxB;
            const int di = i == 0 ? 1 : npts - 1;
            if (y >= e_ye) {
                int idx0 = e_idx;
                int idx = idx0 + di; if (idx >= npts) idx -= npts;
                for (; edges-- > 0;) {
                    const int ty = (int)((sel4(vy, idx) + delta) >> XY_SHIFT);
                    if (ty > y) {
                        const int64_t xs = sel4(vx, idx0), xe = sel4(vx, idx);
                        e_ye = ty;
                        e_dx = div_trunc((xe - xs) * 2 + (ty - y), 2 * (ty - y));
                        e_x = xs; e_ys = y;
                        e_idx = idx;
                        break;
                    }
                    idx0 = idx; idx += di; if (idx >= npts) idx -= npts;
                }
            }
        }
        if (edges < 0) break;
        // scanlines y .. ynext-1 keep both edges; lanes take them 64 at a time
        int ynext = yeA < yeB ? yeA : yeB;
        if (ynext > (int)ymax + 1) ynext = (int)ymax + 1;
        for (int yy = y + lane; yy < ynext; yy += 64) {
            if (yy >= 0) {
                const int64_t xa = xsA + (int64_t)(yy - ysA) * dxA, xb = xsB + (int64_t)(yy - ysB) * dxB;
                const int64_t xl = xa > xb ? xb : xa, xr = xa > xb ? xa : xb;
                const int xx1 = (int)((xl + (XY_ONE >> 1)) >> XY_SHIFT);
                const int xx2 = (int)((xr + (XY_ONE >> 1)) >> XY_SHIFT);
                if (xx2 >= 0 && xx1 < k.W) m_hline_lane(k, yy, xx1, xx2);
            }
        }
        y = ynext;
    }
}
template <bool LANES>
__device__ void m_line_thick_t(const tile_mask& k, int x0, int y0, int x1, int y1, int thickness, int lane)
{
    int64_t p0x = (int64_t)x0 << XY_SHIFT, p0y = (int64_t)y0 << XY_SHIFT;
    int64_t p1x = (int64_t)x1 << XY_SHIFT, p1y = (int64_t)y1 << XY_SHIFT;
    const double INV = 1.0 / XY_ONE;
    double dx = (double)(p0x - p1x) * INV, dy = (double)(p1y - p0y) * INV;
    double r = dx * dx + dy * dy;
    int odd = thickness & 1;
    int64_t th = (int64_t)thickness << (XY_SHIFT - 1);
    if (fabs(r) > 2.220446049250313e-16) {
        r = ((double)th + odd * XY_ONE * 0.5) / sqrt(r);
        int64_t dpx = (int64_t)rint(dy * r), dpy = (int64_t)rint(dx * r);
        int64_t vx[4] = { p0x + dpx, p0x - dpx, p1x - dpx, p1x + dpx };
        int64_t vy[4] = { p0y + dpy, p0y - dpy, p1y - dpy, p1y + dpy };
        if (LANES) m_fill_convex4_lanes(k, vx, vy, lane); else m_fill_convex4(k, vx, vy, lane);
    }
    int rad = (int)((th + (XY_ONE >> 1)) >> XY_SHIFT);
    m_disc(k, x0, y0, rad, lane);
    m_disc(k, x1, y1, rad, lane);
}
__device__ void m_line_thick(const tile_mask& k, int x0, int y0, int x1, int y1, int thickness, int lane)
{
    m_line_thick_t<false>(k, x0, y0, x1, y1, thickness, lane);
}

// ---------------------------------------------------------------------- A4 rain / A5 snow
// Tile 64 x 32, halo R (1 for 3x3, 3 for 7x7).  The float32 pre-blur image (haze / brightness
// applied through the LUT, primitives painted) is staged in LDS as [row][72 px][3] — the float
// image of the frame's byte rows, so interior tiles stage with aligned dword loads; then OpenCV's
// separable blur: row pass over every staged row, column pass over the tile, 4 pixels (12 values)
// per lane with 16-byte LDS accesses, 12 B / 3 x 16 B global stores.
constexpr int BTW = 64, BTH = 32, BRMAX = 3;
constexpr int kSThreads = 512;                 // one lane per (tile row, 4 px) in the column pass
constexpr int BSW = 72;                         // staged pixels per row (>= 64 + 2*3, multiple of 4)
constexpr int MAXHIT = 64;

struct blur_taps { float k[2 * BRMAX + 1]; int r; };

// RR = blur radius (1: 3x3, 3: 7x7).  RR == 1 fuses the row and the column pass in registers
// (no s_row, 4 blocks per CU); RR == 3 keeps the two LDS passes.
// BITS = true: the primitives were rasterised beforehand, once per frame, into a 1-bit coverage map (raster_kernel below;
// bits[job][H][wd] dwords) and this kernel is a pure stencil — no bounding-box scan, no per-tile rasterisation, two barriers
// fewer.  (With the rasteriser inside, a tile a thick drop touches keeps 7 of its 8 waves waiting on the wave that walks the
// drop's scanlines, and 44 % of the tiles of a rain frame are touched: 197 us per 8 frames against 128 us without drops.)
// BITS = false (no workspace given) keeps the self-contained form; both give the same bytes.
template <bool SNOW, int RR, bool BITS>
__global__ __launch_bounds__(kSThreads)
void streak_kernel(const uint8_t* __restrict__ imgs, int H, int W, job_pack<awseg_prim_job> jobs,
                   const int32_t* __restrict__ prims, blur_taps bt,
                   uint8_t* __restrict__ out, float* __restrict__ norm_out, norm_consts nc,
                   const uint32_t* __restrict__ bits_all, int wd)
{
    constexpr int R = RR;
    constexpr int sw = BTW + 2 * R, sh = BTH + 2 * R;
    constexpr int NQ = (sw + 3) / 4;                                        // staged pixel quads per row
    constexpr int RAWD = 56;                                                // dwords per raw staged row
    __shared__ __attribute__((aligned(16))) float s_src[sh * BSW * 3];
    __shared__ __attribute__((aligned(16))) float s_row[RR == 1 ? 4 : sh * BTW * 3];
    __shared__ uint32_t s_raw[sh * RAWD];
    __shared__ uint8_t s_mask[BITS ? 4 : sh * BSW];
    __shared__ uint32_t s_mbits[BITS ? sh * 4 : 1];                          // BITS: the <= 4 coverage dwords a staged row touches
    __shared__ int s_hits[BITS ? 1 : MAXHIT];
    __shared__ int s_nhit;
    __shared__ weather_lut L;
    const awseg_prim_job job = jobs.j[blockIdx.z];
    const int64_t hw = (int64_t)H * W;
    const uint8_t* src = imgs + (int64_t)job.image * hw * 3;
    const int x0 = blockIdx.x * BTW, y0 = blockIdx.y * BTH;
    const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;

    // Interior tiles: issue the staging loads (aligned dwords of the byte rows) right away; their
    // latency hides behind the LUT fill, the primitive scan and the rasterisation.
    const bool interior = (x0 - R >= 0) && (x0 + BTW + R <= W) && (y0 - R >= 0) && (y0 + BTH + R <= H) && ((W * 3) % 4 == 0);
    const int b0 = (x0 - R) * 3;
    constexpr int nbytes = sw * 3;
    const int d0 = b0 >> 2, roff = b0 & 3;
    const int nd = ((b0 + nbytes + 3) >> 2) - d0;                           // <= 55
    constexpr int kPre = (sh * RAWD + kSThreads - 1) / kSThreads;
    uint32_t pre[kPre];
    if (interior) {
#pragma unroll
        for (int u = 0; u < kPre; ++u) {
            const int i = threadIdx.x + u * kSThreads;
            const int ty = i / RAWD, di = i - ty * RAWD;
            if (ty < sh && di < nd) pre[u] = reinterpret_cast<const uint32_t*>(src + (int64_t)(y0 - R + ty) * W * 3)[d0 + di];
            else pre[u] = 0u;
        }
    }
    // BITS: coverage dwords of the staged rows (columns x0-R .. x0+BTW+R-1 span at most 4 dwords)
    const uint32_t* bits = BITS ? bits_all + (int64_t)blockIdx.z * H * wd : nullptr;
    const int mb0 = (x0 - R) >> 5;                                          // first coverage dword of an interior tile's staged rows
    uint32_t mpre = 0u;
    if (BITS && interior && (int)threadIdx.x < sh * 4) {
        const int ty = threadIdx.x >> 2, di = threadIdx.x & 3;
        const int dw = mb0 + di;
        mpre = dw < wd ? bits[(int64_t)(y0 - R + ty) * wd + dw] : 0u;
    }
    // this lane's primitive for the bounding-box scan
    const int32_t* pl = prims + (int64_t)job.prim_offset * (SNOW ? 3 : 5);
    constexpr int PW = SNOW ? 3 : 5;
    int pv[5] = { 0, 0, 0, 0, 0 };
    if (!BITS && (int)threadIdx.x < job.prim_count) {
#pragma unroll
        for (int f = 0; f < PW; ++f) pv[f] = pl[threadIdx.x * PW + f];
    }
    // pre-blur value of an input byte: haze (:134-135) or brightness boost + clip (:179-180)
    float pm, pa;
    if (SNOW) { pm = 1.f; pa = (float)(job.intensity * 0.2); }
    else { double haze = job.intensity * 0.3; pm = (float)(1.0 - haze); pa = (float)(haze * 0.7); }
    for (int i = threadIdx.x; i < 256; i += kSThreads) {
        float v = (float)i / 255.0f;
        if (SNOW) { v = v + pa; v = v < 0.f ? 0.f : (v > 1.f ? 1.f : v); }
        else { v = v * pm; v = v + pa; }
        L.in[i] = v;
        if (norm_out) {
#pragma unroll
            for (int c = 0; c < 3; ++c) L.nrm[c][i] = norm1((uint8_t)i, nc.mean[c], nc.std[c]);
        }
    }
    if (interior) {
#pragma unroll
        for (int u = 0; u < kPre; ++u) {
            const int i = threadIdx.x + u * kSThreads;
            if (i < sh * RAWD) s_raw[i] = pre[u];
        }
        if (BITS && (int)threadIdx.x < sh * 4) s_mbits[threadIdx.x] = mpre;
    }
    // coverage mask over the part of tile+halo that lies inside the image
    tile_mask mk;
    mk.m = s_mask; mk.W = W; mk.H = H; mk.bits = nullptr; mk.wd = 0;
    mk.x0 = x0 - R < 0 ? 0 : x0 - R;
    mk.y0 = y0 - R < 0 ? 0 : y0 - R;
    const int xe = x0 + BTW + R > W ? W : x0 + BTW + R, ye = y0 + BTH + R > H ? H : y0 + BTH + R;
    mk.w = xe - mk.x0; mk.h = ye - mk.y0;
    if (!BITS) {
    for (int i = threadIdx.x; i < mk.w * mk.h; i += kSThreads) s_mask[i] = 0;
    if (threadIdx.x == 0) s_nhit = 0;
    __syncthreads();
    // 1. which primitives touch this tile (bounding boxes, all lanes)
    for (int i = threadIdx.x, u = 0; i < job.prim_count; i += kSThreads, ++u) {
        int lx, hx, ly, hy;
        int f0, f1, f2, f3 = 0, f4 = 0;
        if (u == 0) { f0 = pv[0]; f1 = pv[1]; f2 = pv[2]; f3 = pv[3]; f4 = pv[4]; }
        else { f0 = pl[i * PW]; f1 = pl[i * PW + 1]; f2 = pl[i * PW + 2]; if (!SNOW) { f3 = pl[i * PW + 3]; f4 = pl[i * PW + 4]; } }
        if (SNOW) {
            const int cx = f0, cy = f1, r = f2;
            lx = cx - r; hx = cx + r; ly = cy - r; hy = cy + r;
        } else {
            const int ax = f0, ay = f1, bx = f2, by = f3, th = f4;
            const int mg = th <= 1 ? 0 : th + 2;
            lx = (ax < bx ? ax : bx) - mg; hx = (ax > bx ? ax : bx) + mg;
            ly = (ay < by ? ay : by) - mg; hy = (ay > by ? ay : by) + mg;
        }
        if (!(hx < mk.x0 || lx >= xe || hy < mk.y0 || ly >= ye)) {
            const int slot = atomicAdd(&s_nhit, 1);
            if (slot < MAXHIT) s_hits[slot] = i;
        }
    }
    __syncthreads();
    // 2. rasterise: one wave per hit (a tile sees a handful at most; more than MAXHIT -> every primitive)
    const int nhit = s_nhit;
    const int nwork = nhit <= MAXHIT ? nhit : job.prim_count;
    for (int hidx = wv; hidx < nwork; hidx += kSThreads / 64) {
        const int i = nhit <= MAXHIT ? s_hits[hidx] : hidx;
        if (SNOW) m_disc(mk, pl[i * 3], pl[i * 3 + 1], pl[i * 3 + 2], lane);
        else {
            const int ax = pl[i * 5], ay = pl[i * 5 + 1], bx = pl[i * 5 + 2], by = pl[i * 5 + 3], th = pl[i * 5 + 4];
            if (th <= 1) m_line_thin(mk, ax, ay, bx, by, lane);
            else m_line_thick(mk, ax, ay, bx, by, th, lane);
        }
    }
    }   // !BITS
    __syncthreads();
    // 3. stage the pre-blur float image of tile + halo: one lane per (row, pixel quad)
    const float col[3] = { SNOW ? 1.0f : 0.8f, SNOW ? 1.0f : 0.9f, 1.0f };
    if (interior) {
        for (int i = threadIdx.x; i < sh * NQ; i += kSThreads) {
            const int ty = i / NQ, tq = i - ty * NQ;
            // 12 bytes of 4 pixels, unaligned by roff inside the raw dword row: 4 dword reads + byte-align
            const int bo = tq * 12 + roff;
            const uint32_t* rw = s_raw + ty * RAWD + (bo >> 2);
            const uint32_t w0 = rw[0], w1 = rw[1], w2 = rw[2], w3 = rw[3];
            const int sft = (bo & 3) * 8;
            uint32_t d[3];
            d[0] = sft ? (w0 >> sft) | (w1 << (32 - sft)) : w0;
            d[1] = sft ? (w1 >> sft) | (w2 << (32 - sft)) : w1;
            d[2] = sft ? (w2 >> sft) | (w3 << (32 - sft)) : w2;
            const uint8_t* mrow = s_mask + (BITS ? 0 : (y0 - R + ty - mk.y0) * mk.w + (x0 - R + tq * 4 - mk.x0));
            uint32_t cbits = 0u;                                            // BITS: coverage of the quad's four pixels in bits 0..3
            if (BITS) {
                const int bo2 = (x0 - R + tq * 4) - (mb0 << 5);             // bit offset inside the row's staged dwords (0 .. 127)
                const uint32_t* mr = s_mbits + ty * 4 + (bo2 >> 5);
                const uint32_t lo = mr[0], hi = (bo2 >> 5) < 3 ? mr[1] : 0u;
                cbits = (uint32_t)((((uint64_t)hi << 32) | lo) >> (bo2 & 31));
            }
            float v[12];
#pragma unroll
            for (int e = 0; e < 12; ++e) {
                const int k = e / 3, c = e - k * 3;
                const bool cov = (tq * 4 + k < sw) && (BITS ? ((cbits >> k) & 1u) != 0u : mrow[k] != 0);
                v[e] = cov ? col[c] : L.in[(d[e >> 2] >> ((e & 3) * 8)) & 0xFF];
            }
            float4* o4 = reinterpret_cast<float4*>(s_src + (ty * BSW + tq * 4) * 3);
            o4[0] = make_float4(v[0], v[1], v[2], v[3]); o4[1] = make_float4(v[4], v[5], v[6], v[7]); o4[2] = make_float4(v[8], v[9], v[10], v[11]);
        }
    } else {
        for (int i = threadIdx.x; i < sh * sw; i += kSThreads) {
            const int ty = i / sw, tx = i - ty * sw;
            const int gy = reflect_101(y0 - R + ty, H), gx = reflect_101(x0 - R + tx, W);
            const bool inrect = (gy >= mk.y0 && gy < ye && gx >= mk.x0 && gx < xe);
            const bool cov = BITS ? ((bits[(int64_t)gy * wd + (gx >> 5)] >> (gx & 31)) & 1u) != 0u
                                  : (inrect && s_mask[(gy - mk.y0) * mk.w + (gx - mk.x0)]);
            const uint8_t* px = src + ((int64_t)gy * W + gx) * 3;
#pragma unroll
            for (int c = 0; c < 3; ++c) s_src[(ty * BSW + tx) * 3 + c] = cov ? col[c] : L.in[px[c]];
        }
    }
    __syncthreads();
    // row pass of one staged row for 4 output pixels: k[c]*s0 + sum k[c+j]*(s[-j] + s[+j])
    auto row_pass = [&](int ty, int tq, float* o) {
        const float* sp = s_src + (ty * BSW + tq * 4) * 3;              // window starts R pixels left of the outputs
        constexpr int nwin = (4 + 2 * R) * 3;
        float win[(nwin + 3) / 4 * 4];
#pragma unroll
        for (int k = 0; k < (nwin + 3) / 4; ++k) {
            const float4 t = *reinterpret_cast<const float4*>(sp + 4 * k);
            win[4 * k] = t.x; win[4 * k + 1] = t.y; win[4 * k + 2] = t.z; win[4 * k + 3] = t.w;
        }
#pragma unroll
        for (int e = 0; e < 12; ++e) {
            float acc = bt.k[R] * win[e + 3 * R];
#pragma unroll
            for (int j = 1; j <= R; ++j) { float ab = win[e + 3 * R - 3 * j] + win[e + 3 * R + 3 * j]; float m = bt.k[R + j] * ab; acc = acc + m; }
            o[e] = acc;
        }
    };
    uint8_t* dst = out ? out + (int64_t)job.image * hw * 3 : nullptr;
    float* ndst = norm_out ? norm_out + (int64_t)job.image * hw * 3 : nullptr;
    const int ty = threadIdx.x / (BTW / 4), tq = threadIdx.x - ty * (BTW / 4);   // one (tile row, 4 px) per lane
    const int gy = y0 + ty, gx = x0 + tq * 4;
    float acc[12];
    if (RR == 1) {
        // 4. + 5. fused: the three row-pass results this lane's column pass needs are recomputed in
        // registers (same operations, same order as the two-pass form)
        float up[12], ctr[12], dn[12];
        row_pass(ty, tq, up); row_pass(ty + 1, tq, ctr); row_pass(ty + 2, tq, dn);
#pragma unroll
        for (int e = 0; e < 12; ++e) { float a0 = bt.k[1] * ctr[e]; float ab = up[e] + dn[e]; float m = bt.k[2] * ab; acc[e] = a0 + m; }
    } else {
        for (int i = threadIdx.x; i < sh * (BTW / 4); i += kSThreads) {
            const int ry = i / (BTW / 4), rq = i - ry * (BTW / 4);
            float o[12];
            row_pass(ry, rq, o);
            float4* d = reinterpret_cast<float4*>(s_row + (ry * BTW + rq * 4) * 3);
            d[0] = make_float4(o[0], o[1], o[2], o[3]); d[1] = make_float4(o[4], o[5], o[6], o[7]); d[2] = make_float4(o[8], o[9], o[10], o[11]);
        }
        __syncthreads();
        auto ldrow = [&](int rr, float* v) {
            const float4* p4 = reinterpret_cast<const float4*>(s_row + (rr * BTW + tq * 4) * 3);
            float4 a = p4[0], b = p4[1], c = p4[2];
            v[0] = a.x; v[1] = a.y; v[2] = a.z; v[3] = a.w; v[4] = b.x; v[5] = b.y; v[6] = b.z; v[7] = b.w;
            v[8] = c.x; v[9] = c.y; v[10] = c.z; v[11] = c.w;
        };
        float ctr[12];
        ldrow(ty + R, ctr);
#pragma unroll
        for (int e = 0; e < 12; ++e) acc[e] = bt.k[R] * ctr[e];
#pragma unroll
        for (int j = 1; j <= R; ++j) {
            float up[12], dn[12];
            ldrow(ty + R - j, up); ldrow(ty + R + j, dn);
#pragma unroll
            for (int e = 0; e < 12; ++e) { float ab = up[e] + dn[e]; float m = bt.k[R + j] * ab; acc[e] = acc[e] + m; }
        }
    }
    if (gy < H && gx < W) {
        uint8_t res[12];
#pragma unroll
        for (int e = 0; e < 12; ++e) res[e] = quant_f32(acc[e]);
        const int nvalid = (W - gx) < 4 ? (W - gx) : 4;
        const int64_t p = (int64_t)gy * W + gx;
        if (dst) {
            if (nvalid == 4 && (((uintptr_t)(dst + p * 3)) & 3) == 0) {
                uint32_t* d4 = reinterpret_cast<uint32_t*>(dst + p * 3);
#pragma unroll
                for (int w = 0; w < 3; ++w)
                    d4[w] = (uint32_t)res[4 * w] | ((uint32_t)res[4 * w + 1] << 8) | ((uint32_t)res[4 * w + 2] << 16) | ((uint32_t)res[4 * w + 3] << 24);
            } else {
                for (int k = 0; k < nvalid * 3; ++k) dst[p * 3 + k] = res[k];
            }
        }
        if (ndst) {
            if (nvalid == 4 && (p & 3) == 0 && (hw & 3) == 0) {
#pragma unroll
                for (int c = 0; c < 3; ++c)
                    *reinterpret_cast<float4*>(ndst + (int64_t)c * hw + p) =
                        make_float4(L.nrm[c][res[c]], L.nrm[c][res[3 + c]], L.nrm[c][res[6 + c]], L.nrm[c][res[9 + c]]);
            } else {
                for (int c = 0; c < 3; ++c)
                    for (int k = 0; k < nvalid; ++k) ndst[(int64_t)c * hw + p + k] = L.nrm[c][res[k * 3 + c]];
            }
        }
    }
}

// The same transform with the coverage map, as a STRIP kernel: a lane owns 4 adjacent pixels and walks NR rows of them; no
// float image in LDS, no barrier after the table fill.  Per input row a lane loads the (4 + 2R) pixels of its window as
// aligned dwords (bytes picked with compile-time shifts), maps them through the LDS table (or the streak colour where the
// coverage bit is set), runs OpenCV's row pass for its 12 values and keeps the last 2R + 1 row-pass results in registers;
// every row after the first 2R closes one output row: column pass, quantise, 12 B / 3 x 16 B stores.  The row loop is
// unrolled by 2R + 1, so the ring slots are register names.  Same operations in the same order as the tile kernel (the
// file is built with -ffp-contract=off): byte-identical output.  ~265 instructions per pixel quad against ~750: the tile
// kernel converts 1.2 pixels per output, makes two LDS passes over the float image and pays its prologue per 2048 pixels.
template <int RR> struct strip_rows { static constexpr int value = RR == 1 ? 16 : 15; };   // NR + 2R is a multiple of 2R + 1

// FAST (W % 4 == 0, W >= 16; chosen by the launcher): every lane takes the aligned-dword window.  A lane at the left / right
// image edge loads with its dword indices clamped into the row and then copies the BORDER_REFLECT_101 source pixels over the
// out-of-image ones (pixel -j <- pixel j, pixel W-1+j <- pixel W-1-j: 3R selects per side) — no divergent second path.
template <bool SNOW, int RR, bool FAST>
__device__ __forceinline__ void streak_strip_body(const uint8_t* __restrict__ imgs, int H, int W, const awseg_prim_job job, const blur_taps& bt,
                                                  uint8_t* __restrict__ out, float* __restrict__ norm_out, const norm_consts& nc,
                                                  const uint32_t* __restrict__ bits, int wd, weather_lut& L, float* s_in, int bx, int by)
{
    // s_in [256 + 4]: pre-blur value of a byte; entries 256..258: the streak colour per channel, so a covered pixel is an index
    // select in front of ONE table read
    constexpr int R = RR, NR = strip_rows<RR>::value, NW = 4 + 2 * R, NBYTE = NW * 3, RING = 2 * R + 1;
    constexpr int OFF = (4 - ((3 * R) & 3)) & 3;                 // (gx - R) * 3 mod 4 for gx % 4 == 0: 1 (R = 1), 3 (R = 3)
    constexpr int ND = (OFF + NBYTE + 3) / 4;                    // dwords that hold the window's bytes: 5 / 9
    float pm, pa;
    if (SNOW) { pm = 1.f; pa = (float)(job.intensity * 0.2); }
    else { double haze = job.intensity * 0.3; pm = (float)(1.0 - haze); pa = (float)(haze * 0.7); }
    for (int i = threadIdx.x; i < 256; i += 256) {
        float v = (float)i / 255.0f;
        if (SNOW) { v = v + pa; v = v < 0.f ? 0.f : (v > 1.f ? 1.f : v); }
        else { v = v * pm; v = v + pa; }
        s_in[i] = v;
        if (i < 3) s_in[256 + i] = i == 2 ? 1.0f : (SNOW ? 1.0f : (i == 0 ? 0.8f : 0.9f));
        if (norm_out) {
#pragma unroll
            for (int c = 0; c < 3; ++c) L.nrm[c][i] = norm1((uint8_t)i, nc.mean[c], nc.std[c]);
        }
    }
    __syncthreads();
    const int64_t hw = (int64_t)H * W;
    const int gx = (bx * 64 + (threadIdx.x & 63)) * 4;
    const int y0 = (by * 4 + (threadIdx.x >> 6)) * NR;
    if (gx >= W || y0 >= H) return;
    const uint8_t* src = imgs + (int64_t)job.image * hw * 3;
    uint8_t* dst = out ? out + (int64_t)job.image * hw * 3 : nullptr;
    float* ndst = norm_out ? norm_out + (int64_t)job.image * hw * 3 : nullptr;
    const int nvalid = (W - gx) < 4 ? (W - gx) : 4;
    const bool fast = FAST;
    const float col[3] = { SNOW ? 1.0f : 0.8f, SNOW ? 1.0f : 0.9f, 1.0f };
    const bool edge_l = gx - R < 0, edge_r = gx + 3 + R >= W;     // FAST: W >= 16, so never both
    const int d0 = ((gx - R) * 3) >> 2;                           // first dword of the window in a row (arithmetic shift: -1 / -3 at the left edge)
    int doff[ND];                                                 // byte offsets of the window's dwords in a row, clamped into the row
#pragma unroll
    for (int d = 0; d < ND; ++d) {
        int di = d0 + d;
        di = di < 0 ? 0 : (di > (W * 3) / 4 - 1 ? (W * 3) / 4 - 1 : di);
        doff[d] = di * 4;
    }
    const int mb = (gx - R) >> 5, ms = (gx - R) & 31;             // coverage dword / shift of the window's first pixel (mb = -1 at the left edge)

    uint32_t raw[ND]; uint32_t mlo = 0u, mhi = 0u;
    auto issue = [&](int i) {                                     // loads of input row i (image row reflect(y0 - R + i))
        const int gy = reflect_101(y0 - R + i, H);
        if (fast) {
            const uint8_t* rp = src + (int64_t)gy * W * 3;
#pragma unroll
            for (int d = 0; d < ND; ++d) raw[d] = *reinterpret_cast<const uint32_t*>(rp + doff[d]);
            const uint32_t* mp = bits + (int64_t)gy * wd;
            mlo = mb >= 0 ? mp[mb] : 0u; mhi = (mb + 1 < wd) ? mp[mb + 1] : 0u;
        }
    };
    // pre-blur values of the window (NW pixels x 3) of input row i: from the registers `issue` filled, or pixel by pixel
    auto window = [&](int i, float* win) {
        if (fast) {
            const uint32_t cb = (uint32_t)((((uint64_t)mhi << 32) | mlo) >> ms);
#pragma unroll
            for (int e = 0; e < NBYTE; ++e) {
                const int k = e / 3, c = e - k * 3, bp = OFF + e;
                const uint32_t byte = (raw[bp >> 2] >> (8 * (bp & 3))) & 0xFFu;
                win[e] = s_in[((cb >> k) & 1u) ? 256u + c : byte];
            }
#pragma unroll
            for (int j = 1; j <= R; ++j)
#pragma unroll
                for (int c = 0; c < 3; ++c) {
                    win[(R - j) * 3 + c] = edge_l ? win[(R + j) * 3 + c] : win[(R - j) * 3 + c];
                    win[(R + 3 + j) * 3 + c] = edge_r ? win[(R + 3 - j) * 3 + c] : win[(R + 3 + j) * 3 + c];
                }
        } else {
            const int gy = reflect_101(y0 - R + i, H);
#pragma unroll
            for (int k = 0; k < NW; ++k) {
                const int xk = reflect_101(gx - R + k, W);
                const bool cov = ((bits[(int64_t)gy * wd + (xk >> 5)] >> (xk & 31)) & 1u) != 0u;
                const uint8_t* px = src + ((int64_t)gy * W + xk) * 3;
#pragma unroll
                for (int c = 0; c < 3; ++c) win[k * 3 + c] = cov ? col[c] : s_in[px[c]];
            }
        }
    };
    float ring[RING][12];
    auto row_pass = [&](const float* win, float* o) {
#pragma unroll
        for (int e = 0; e < 12; ++e) {
            float acc = bt.k[R] * win[e + 3 * R];
#pragma unroll
            for (int j = 1; j <= R; ++j) { float ab = win[e + 3 * R - 3 * j] + win[e + 3 * R + 3 * j]; float m = bt.k[R + j] * ab; acc = acc + m; }
            o[e] = acc;
        }
    };
    // output row o (image row y0 + o) from the ring; `ctr` = ring slot of its centre row
    auto emit = [&](int o, int ctr) {
        const int gy = y0 + o;
        float acc[12];
#pragma unroll
        for (int e = 0; e < 12; ++e) acc[e] = bt.k[R] * ring[ctr][e];
#pragma unroll
        for (int j = 1; j <= R; ++j) {
            const int up = (ctr - j + 2 * RING) % RING, dn = (ctr + j) % RING;
#pragma unroll
            for (int e = 0; e < 12; ++e) { float ab = ring[up][e] + ring[dn][e]; float m = bt.k[R + j] * ab; acc[e] = acc[e] + m; }
        }
        uint8_t res[12];
#pragma unroll
        for (int e = 0; e < 12; ++e) res[e] = (uint8_t)(int)(__builtin_amdgcn_fmed3f(acc[e], 0.f, 1.f) * 255.0f);   // quant_f32 with one v_med3
        const int64_t p = (int64_t)gy * W + gx;
        if (dst) {
            if (FAST || (nvalid == 4 && (((uintptr_t)(dst + p * 3)) & 3) == 0)) {
                uint32_t* d4 = reinterpret_cast<uint32_t*>(dst + p * 3);
#pragma unroll
                for (int q = 0; q < 3; ++q)
                    d4[q] = (uint32_t)res[4 * q] | ((uint32_t)res[4 * q + 1] << 8) | ((uint32_t)res[4 * q + 2] << 16) | ((uint32_t)res[4 * q + 3] << 24);
            } else {
                for (int k = 0; k < nvalid * 3; ++k) dst[p * 3 + k] = res[k];
            }
        }
        if (ndst) {
            if (FAST || (nvalid == 4 && (p & 3) == 0 && (hw & 3) == 0)) {
#pragma unroll
                for (int c = 0; c < 3; ++c)
                    *reinterpret_cast<float4*>(ndst + (int64_t)c * hw + p) =
                        make_float4(L.nrm[c][res[c]], L.nrm[c][res[3 + c]], L.nrm[c][res[6 + c]], L.nrm[c][res[9 + c]]);
            } else {
                for (int c = 0; c < 3; ++c)
                    for (int k = 0; k < nvalid; ++k) ndst[(int64_t)c * hw + p + k] = L.nrm[c][res[k * 3 + c]];
            }
        }
    };
    const int rows_out = (H - y0) < NR ? (H - y0) : NR;           // output rows of this strip (wave-uniform)
    issue(0);
    for (int i0 = 0; i0 < NR + 2 * R; i0 += RING) {
#pragma unroll
        for (int u = 0; u < RING; ++u) {
            const int i = i0 + u;                                  // ring slot of input row i is u (i0 is a multiple of RING)
            if (i - 2 * R >= rows_out) return;                     // nothing left to emit (wave-uniform)
            float win[NBYTE];
            window(i, win);
            if (i + 1 < NR + 2 * R) issue(i + 1);                  // next row's loads fly during this row's arithmetic
            row_pass(win, ring[u]);
            if (i >= 2 * R) emit(i - 2 * R, (u - R + RING) % RING);
        }
    }
}

template <bool SNOW, int RR, bool FAST>
__global__ __launch_bounds__(256)
void streak_strip_kernel(const uint8_t* __restrict__ imgs, int H, int W, job_pack<awseg_prim_job> jobs, blur_taps bt,
                         uint8_t* __restrict__ out, float* __restrict__ norm_out, norm_consts nc,
                         const uint32_t* __restrict__ bits_all, int wd)
{
    __shared__ weather_lut L;
    __shared__ float s_in[256 + 4];
    streak_strip_body<SNOW, RR, FAST>(imgs, H, W, jobs.j[blockIdx.z], bt, out, norm_out, nc, bits_all + (int64_t)blockIdx.z * H * wd, wd, L, s_in,
                                      blockIdx.x, blockIdx.y);
}

// Coverage pre-pass: one wave per primitive of a frame, the same integer rasteriser writing a frame-wide 1-bit map
// (atomicOr; bits[job][H][wd]).  The map is zeroed by the launcher.
template <bool SNOW>
__global__ __launch_bounds__(256)
void raster_kernel(int H, int W, job_pack<awseg_prim_job> jobs, const int32_t* __restrict__ prims, uint32_t* __restrict__ bits_all, int wd)
{
    const awseg_prim_job job = jobs.j[blockIdx.y];
    const int lane = threadIdx.x & 63, i = blockIdx.x * 4 + (threadIdx.x >> 6);
    if (i >= job.prim_count) return;                                        // wave-uniform
    tile_mask mk;
    mk.m = nullptr; mk.W = W; mk.H = H; mk.x0 = 0; mk.y0 = 0; mk.w = W; mk.h = H;
    mk.bits = bits_all + (int64_t)blockIdx.y * H * wd; mk.wd = wd;
    const int32_t* pl = prims + (int64_t)job.prim_offset * (SNOW ? 3 : 5);
    if (SNOW) m_disc(mk, pl[i * 3], pl[i * 3 + 1], pl[i * 3 + 2], lane);
    else {
        const int ax = pl[i * 5], ay = pl[i * 5 + 1], bx = pl[i * 5 + 2], by = pl[i * 5 + 3], th = pl[i * 5 + 4];
        if (th <= 1) m_line_thin(mk, ax, ay, bx, by, lane);
        else m_line_thick_t<true>(mk, ax, ay, bx, by, th, lane);
    }
}

// -------------------------------------------------------------------------------- A16
__global__ __launch_bounds__(kThreads)
void density_field_kernel(int64_t hw, uint64_t seed, const float* __restrict__ scale_off, const float* __restrict__ uniform,
                          float* __restrict__ density)
{
    const int64_t img = blockIdx.y;
    const float sc = scale_off[img * 2], of = scale_off[img * 2 + 1];
    float* dst = density + img * hw;
    const float* usrc = uniform ? uniform + img * hw : nullptr;      // parity mode: the host's torch.rand draws
    const int64_t nquad = (hw + 3) / 4;
    for (int64_t q = (int64_t)blockIdx.x * kThreads + threadIdx.x; q < nquad; q += (int64_t)gridDim.x * kThreads) {
        uint32_t r[4] = {0u, 0u, 0u, 0u};
        if (!usrc) awseg_philox::gen(seed, (uint64_t)(img * nquad + q), 0x0DE5u, r);
#pragma unroll
        for (int k = 0; k < 4; ++k) {
            int64_t p = q * 4 + k;
            // torch.rand(h, w) * a + b in float32, two separately rounded operations (trainer.py:503-509); the "else"
            // branch of the reference has no "+ b": u * 0.1 + 0.0f is the same float (u >= 0)
            if (p < hw) { float u = usrc ? usrc[p] : awseg_u01(r[k]); float v = u * sc; dst[p] = v + of; }
        }
    }
}

template <typename J> job_pack<J> pack_jobs(const J* jobs, int first, int count)
{
    job_pack<J> p;
    for (int i = 0; i < kMaxJobs; ++i) p.j[i] = jobs[first + (i < count ? i : 0)];
    return p;
}

norm_consts make_nc(const float* mean, const float* std)
{
    norm_consts nc;
    for (int c = 0; c < 3; ++c) { nc.mean[c] = mean ? mean[c] : 0.f; nc.std[c] = std ? std[c] : 1.f; }
    return nc;
}

// cv::getGaussianKernel(ksize, sigma > 0, CV_32F)
blur_taps make_blur(int ksize, double sigma)
{
    blur_taps b; b.r = ksize / 2;
    double t[2 * BRMAX + 1], sum = 0.0, s2 = -0.5 / (sigma * sigma);
    for (int i = 0; i < ksize; ++i) { double x = i - (ksize - 1) * 0.5; t[i] = exp(s2 * x * x); sum += t[i]; }
    sum = 1.0 / sum;
    for (int i = 0; i < 2 * BRMAX + 1; ++i) b.k[i] = 0.f;
    for (int i = 0; i < ksize; ++i) b.k[i] = (float)(t[i] * sum);
    return b;
}

int grid_for(int64_t items, int64_t jobs)
{
    int64_t want = (items + kThreads - 1) / kThreads;
    int64_t cap = (AWSEG_CUS * 8 + jobs - 1) / (jobs < 1 ? 1 : jobs);
    if (want > cap) want = cap;
    return (int)(want < 1 ? 1 : want);
}

// ---------------------------------------------------------------------------------------------------------------------------
// ONE launch for a batch of frames of MIXED kinds (throughput mode: in-kernel Philox noise).  A step of the evaluation loop
// draws a condition per frame (PKG/data/loader.py:265), so with a launch per kind each launch covers one or two frames and the five
// of them run back to back, every one bound by its own launch latency and tail (0.19 ms for 252 MB of algorithmic traffic, round 3).
// Here every frame is a JOB with a block range; a block looks its job up in the table that travels in the kernel arguments and runs
// that kind's body — the bodies above, unchanged, so the bytes are those of the per-kind launchers.  Long-running kinds (the fog
// strips) get the first block indices.  Rain / snow read the coverage maps of the raster pre-pass (two small launches in front).
struct wb_entry {
    int kind, b0, nbx, nby, slot, strip_rows;
    awseg_fog_job fog; awseg_night_job night; awseg_prim_job prim;
};
struct wb_table { int n; int pad; wb_entry e[kMaxJobs]; };

__global__ __launch_bounds__(256)
void weather_batch_kernel(const uint8_t* __restrict__ imgs, int H, int W, wb_table tab, gauss_taps_f32 taps, blur_taps bt_rain, blur_taps bt_snow,
                          float g0, float g1, float g2, uint8_t* __restrict__ out, float* __restrict__ norm_out, norm_consts nc,
                          const uint32_t* __restrict__ bits_all, int wd)
{
    __shared__ weather_lut L;
    __shared__ float s_in[256 + 4];
    extern __shared__ float4 s_ring[];                              // fog: [17][256]
    const int b = blockIdx.x;
    int j = 0;
    for (int k = 1; k < tab.n; ++k) j = b >= tab.e[k].b0 ? k : j;    // entries are sorted by their first block
    const wb_entry e = tab.e[j];
    const int local = b - e.b0, bx = local % e.nbx, by = local / e.nbx;
    const int64_t hw = (int64_t)H * W;
    switch (e.kind) {
    case AWSEG_WEATHER_CLEAN: normalize_body(imgs, hw, e.fog.image, nc, norm_out, true, bx, e.nbx); break;
    case AWSEG_WEATHER_FOG: fog_strip_body(imgs, H, W, e.fog, 0, taps, out, norm_out, nullptr, nc, e.strip_rows, L, s_ring, bx, by); break;
    case AWSEG_WEATHER_RAIN:
        streak_strip_body<false, 1, true>(imgs, H, W, e.prim, bt_rain, out, norm_out, nc, bits_all + (int64_t)e.slot * H * wd, wd, L, s_in, bx, by); break;
    case AWSEG_WEATHER_SNOW:
        streak_strip_body<true, 1, true>(imgs, H, W, e.prim, bt_snow, out, norm_out, nc, bits_all + (int64_t)e.slot * H * wd, wd, L, s_in, bx, by); break;
    default: night_body<true>(imgs, hw, e.night, 0, nullptr, g0, g1, g2, out, norm_out, nc, true, L, bx, e.nbx); break;
    }
}

}  // namespace

AWSEG_API int awseg_weather_batch(const uint8_t* imgs, int height, int width, const awseg_weather_job* jobs, int n_jobs,
                                  const int32_t* rain_drops, const int32_t* snow_flakes, const double* taps_host, const float* gains_host,
                                  uint8_t* out, float* norm_out, const float* mean_host, const float* std_host, void* workspace,
                                  awseg_stream_t stream)
{
    if (n_jobs == 0) return 0;
    if (!imgs || !jobs || !norm_out || !mean_host || !std_host || !taps_host || !gains_host || height < 1 || width < 1 || n_jobs < 0) return AWSEG_EINVAL;
    if (n_jobs > kMaxJobs) return AWSEG_ERANGE;                       // the caller splits larger batches
    const int H = height, W = width;
    const int64_t hw = (int64_t)H * W;
    if ((W & 3) || W < 16) return AWSEG_ERANGE;                       // the strip bodies' aligned-dword windows; other sizes: the per-kind launchers
    if (out == imgs) return AWSEG_EINVAL;
    if (((uintptr_t)imgs & 3) || ((uintptr_t)norm_out & 15) || (out && ((uintptr_t)out & 3)) || (workspace && ((uintptr_t)workspace & 3))) return AWSEG_EALIGN;
    hipStream_t s = awseg_s(stream);
    const int wd = (W + 31) / 32;
    int n_fog = 0, n_rain = 0, n_snow = 0;
    for (int i = 0; i < n_jobs; ++i) {
        const int k = jobs[i].kind;
        if (k < AWSEG_WEATHER_CLEAN || k > AWSEG_WEATHER_NIGHT || jobs[i].image < 0) return AWSEG_EINVAL;
        n_fog += k == AWSEG_WEATHER_FOG; n_rain += k == AWSEG_WEATHER_RAIN; n_snow += k == AWSEG_WEATHER_SNOW;
    }
    if ((n_rain && !rain_drops) || (n_snow && !snow_flakes) || ((n_rain || n_snow) && !workspace)) return AWSEG_EINVAL;
    uint32_t* bits = reinterpret_cast<uint32_t*>(workspace);
    // fog strip height: as fog_common (one round of two blocks per CU; short strips when few frames are foggy)
    int fog_rows = 0, fog_nbx = 0, fog_nby = 0;
    if (n_fog) {
        fog_nbx = ((W >> 2) + kFogStripQuads - 1) / kFogStripQuads;
        const int64_t per_round = (int64_t)AWSEG_CUS * 2 / fog_nbx / n_fog;
        fog_rows = per_round >= 1 ? (int)((H + 4 * per_round - 1) / (4 * per_round)) : H;
        const int min_rows = n_fog <= 2 ? 8 : 24;
        if (fog_rows < min_rows) fog_rows = min_rows;
        fog_nby = ((H + fog_rows - 1) / fog_rows + 3) / 4;
    }
    const int nr = strip_rows<1>::value;
    const int st_nbx = (W + 255) / 256, st_nby = (H + 4 * nr - 1) / (4 * nr);
    int pw_blocks = (int)((hw / 4 + kThreads - 1) / kThreads);         // pointwise kinds: grid-stride over pixel quads
    const int pw_cap = AWSEG_CUS * 8 / n_jobs > 64 ? AWSEG_CUS * 8 / n_jobs : 64;
    if (pw_blocks > pw_cap) pw_blocks = pw_cap;
    if ((hw & 3) != 0) return AWSEG_ERANGE;
    wb_table tab;
    tab.n = 0; tab.pad = 0;
    job_pack<awseg_prim_job> rain_pk, snow_pk;
    awseg_prim_job rsel[kMaxJobs], ssel[kMaxJobs];
    int rcnt = 0, scnt = 0, rmax = 0, smax = 0;
    int next_block = 0;
    const int order[5] = { AWSEG_WEATHER_FOG, AWSEG_WEATHER_RAIN, AWSEG_WEATHER_SNOW, AWSEG_WEATHER_NIGHT, AWSEG_WEATHER_CLEAN };
    for (int o = 0; o < 5; ++o)
        for (int i = 0; i < n_jobs; ++i) {
            const awseg_weather_job& jb = jobs[i];
            if (jb.kind != order[o]) continue;
            wb_entry& e = tab.e[tab.n++];
            e.kind = jb.kind; e.b0 = next_block; e.slot = 0; e.strip_rows = fog_rows;
            e.fog.image = jb.image; e.fog._pad = 0; e.fog.beta = jb.a; e.fog.atmos = jb.b; e.fog.seed = jb.seed;
            e.night.image = jb.image; e.night._pad = 0; e.night.brightness = jb.a; e.night.intensity = jb.b; e.night.seed = jb.seed;
            e.prim.image = jb.image; e.prim.prim_offset = jb.prim_offset; e.prim.prim_count = jb.prim_count; e.prim.blur_ksize = 3; e.prim.intensity = jb.a;
            if (jb.kind == AWSEG_WEATHER_FOG) { e.nbx = fog_nbx; e.nby = fog_nby; }
            else if (jb.kind == AWSEG_WEATHER_RAIN || jb.kind == AWSEG_WEATHER_SNOW) {
                if (jb.prim_count < 0 || jb.prim_offset < 0) return AWSEG_EINVAL;
                e.nbx = st_nbx; e.nby = st_nby;
                if (jb.kind == AWSEG_WEATHER_RAIN) { e.slot = rcnt; rsel[rcnt++] = e.prim; rmax = jb.prim_count > rmax ? jb.prim_count : rmax; }
                else { e.slot = n_rain + scnt; ssel[scnt++] = e.prim; smax = jb.prim_count > smax ? jb.prim_count : smax; }
            }
            else { e.nbx = pw_blocks; e.nby = 1; }
            next_block += e.nbx * e.nby;
        }
    if (n_rain + n_snow) {
        if (hipMemsetAsync(bits, 0, (size_t)(n_rain + n_snow) * H * wd * sizeof(uint32_t), s) != hipSuccess) return AWSEG_EINVAL;
        if (rcnt && rmax > 0) {
            rain_pk = pack_jobs(rsel, 0, rcnt);
            hipLaunchKernelGGL((raster_kernel<false>), dim3((rmax + 3) / 4, rcnt), dim3(256), 0, s, H, W, rain_pk, rain_drops, bits, wd);
            AWSEG_LAUNCH_CHECK();
        }
        if (scnt && smax > 0) {
            snow_pk = pack_jobs(ssel, 0, scnt);
            hipLaunchKernelGGL((raster_kernel<true>), dim3((smax + 3) / 4, scnt), dim3(256), 0, s, H, W, snow_pk, snow_flakes, bits + (size_t)n_rain * H * wd, wd);
            AWSEG_LAUNCH_CHECK();
        }
    }
    gauss_taps_f32 tf;
    for (int i = 0; i < 2 * FR + 1; ++i) tf.w[i] = (float)taps_host[i];
    const size_t lds = n_fog ? (size_t)kFogRing * 256 * sizeof(float4) : 0;
    static bool attr_set = false;
    if (lds && !attr_set) {
        if (hipFuncSetAttribute(reinterpret_cast<const void*>(weather_batch_kernel), hipFuncAttributeMaxDynamicSharedMemorySize, (int)((size_t)kFogRing * 256 * sizeof(float4))) != hipSuccess) return AWSEG_EINVAL;
        attr_set = true;
    }
    hipLaunchKernelGGL(weather_batch_kernel, dim3((unsigned)next_block), dim3(256), lds, s, imgs, H, W, tab, tf, make_blur(3, 0.5), make_blur(3, 1.0),
                       gains_host[0], gains_host[1], gains_host[2], out, norm_out, make_nc(mean_host, std_host), bits, wd);
    AWSEG_LAUNCH_CHECK();
    return 0;
}

AWSEG_API int awseg_normalize(const uint8_t* imgs, int64_t batch, int height, int width, const int32_t* sel, int n_sel,
                              const float* mean_host, const float* std_host, float* out, awseg_stream_t stream)
{
    if (!imgs || !out || !mean_host || !std_host || height < 1 || width < 1 || batch < 1) return AWSEG_EINVAL;
    const int64_t hw = (int64_t)height * width;
    const int64_t n = sel ? n_sel : batch;
    if (n < 1) return 0;
    if (n > 65535) return AWSEG_ERANGE;
    // the 4-pixel path needs dword / float4 aligned per-image bases: hw % 4 == 0 and aligned buffers; anything else is scalar
    const bool vec = (hw & 3) == 0 && ((uintptr_t)imgs & 3) == 0 && ((uintptr_t)out & 15) == 0;
    dim3 grid(grid_for(vec ? hw / 4 : hw, n), (unsigned)n);
    hipLaunchKernelGGL(normalize_kernel, grid, dim3(kThreads), 0, awseg_s(stream), imgs, hw, sel, make_nc(mean_host, std_host), out, vec);
    AWSEG_LAUNCH_CHECK();
    return 0;
}

AWSEG_API int awseg_lut3_apply(const uint8_t* imgs, int batch, int height, int width, const uint8_t* luts, int n_luts,
                               const int32_t* lut_of, uint8_t* out, awseg_stream_t stream)
{
    if (batch == 0) return 0;
    if (!imgs || !out || !lut_of || (n_luts > 0 && !luts) || batch < 0 || n_luts < 0 || height < 1 || width < 1) return AWSEG_EINVAL;
    if (batch > 65535) return AWSEG_ERANGE;
    const int64_t hw = (int64_t)height * width;
    if (((hw * 3) & 3) || ((uintptr_t)imgs & 3) || ((uintptr_t)out & 3)) return AWSEG_EALIGN;
    dim3 grid(grid_for(hw / 4 > 0 ? hw / 4 : 1, batch), (unsigned)batch);
    hipLaunchKernelGGL(lut3_kernel, grid, dim3(kThreads), 0, awseg_s(stream), imgs, hw, luts, lut_of, out);
    AWSEG_LAUNCH_CHECK();
    return 0;
}

static int fog_common(int mode, const uint8_t* imgs, int H, int W, const awseg_fog_job* jobs, int n_jobs,
                      const double* noise, const double* taps_host, uint8_t* out, float* norm_out, double* depth_out,
                      const float* mean_host, const float* std_host, hipStream_t s)
{
    if (!jobs || !taps_host || H < 1 || W < 1 || n_jobs < 0) return AWSEG_EINVAL;
    if (mode == 1 && !imgs) return AWSEG_EINVAL;
    if (mode == 0 && !depth_out) return AWSEG_EINVAL;
    if (n_jobs == 0) return 0;
    gauss_taps t;
    for (int i = 0; i < 2 * FR + 1; ++i) t.w[i] = taps_host[i];
    if ((H + FTH - 1) / FTH > 65535) return AWSEG_ERANGE;
    norm_consts nc = make_nc(mean_host, std_host);
    gauss_taps_f32 tf;
    for (int i = 0; i < 2 * FR + 1; ++i) tf.w[i] = (float)taps_host[i];
    for (int j0 = 0; j0 < n_jobs; j0 += kMaxJobs) {
        const int cnt = n_jobs - j0 < kMaxJobs ? n_jobs - j0 : kMaxJobs;
        const job_pack<awseg_fog_job> pk = pack_jobs(jobs, j0, cnt);
        dim3 grid((W + FTW - 1) / FTW, (H + FTH - 1) / FTH, cnt);
#define AWSEG_FOG(P, M) hipLaunchKernelGGL((fog_kernel<P, M>), grid, dim3(kThreads), 0, s, imgs, H, W, pk, j0, noise, t, out, norm_out, depth_out, nc)
        if (noise) { if (mode) AWSEG_FOG(false, 1); else AWSEG_FOG(false, 0); }
        else if (mode == 0) AWSEG_FOG(true, 0);      // depth-only request keeps the float64 pipeline
        else {
            const char* sr_env = getenv("AWSEG_FOG_STRIP_ROWS");             // rows per strip; 0: the tile kernel (read per call: tests vary it)
            int strip_rows = sr_env ? atoi(sr_env) : -1;
            if (strip_rows < 0) {
                // two 256-thread blocks fit a CU (70 KB LDS ring each): the strip height that makes the grid ONE round of them
                const int64_t per_round = (int64_t)AWSEG_CUS * 2 / (((W >> 2) + kFogStripQuads - 1) / kFogStripQuads) / cnt;   // y-blocks (4 strips each) available
                strip_rows = per_round >= 1 ? (int)((H + 4 * per_round - 1) / (4 * per_round)) : H;
                // few frames per launch (the in-step case: 1-2 frames of a kind): short strips — the halo rows triple the noise work,
                // but the chip is otherwise idle and a wave's serial row walk is the launch's duration; many frames: >= 24 rows per strip
                const int min_rows = cnt <= 2 ? 8 : 24;
                if (strip_rows < min_rows) strip_rows = min_rows;
            }
            const bool strip_ok = strip_rows >= 1 && (W & 3) == 0 && W >= 16 && (((uintptr_t)imgs & 3) == 0) &&
                                  (!out || ((uintptr_t)out & 3) == 0) && (!norm_out || ((uintptr_t)norm_out & 15) == 0);
            const int strips = strip_ok ? (H + strip_rows - 1) / strip_rows : 0;
            if (strip_ok && (strips + 3) / 4 <= 65535) {
                const size_t lds = (size_t)kFogRing * 256 * sizeof(float4);
                static bool attr_set = false;
                if (!attr_set) {
                    if (hipFuncSetAttribute(reinterpret_cast<const void*>(fog_strip_kernel), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds) != hipSuccess) return AWSEG_EINVAL;
                    attr_set = true;
                }
                dim3 sgrid(((W >> 2) + kFogStripQuads - 1) / kFogStripQuads, (strips + 3) / 4, cnt);
                hipLaunchKernelGGL(fog_strip_kernel, sgrid, dim3(256), lds, s, imgs, H, W, pk, j0, tf, out, norm_out, depth_out, nc, strip_rows);
            }
            else hipLaunchKernelGGL(fog_fast_kernel, grid, dim3(kFogFastThreads), 0, s, imgs, H, W, pk, j0, tf, out, norm_out, depth_out, nc);
        }
#undef AWSEG_FOG
        AWSEG_LAUNCH_CHECK();
    }
    return 0;
}

AWSEG_API int awseg_synthetic_depth(int height, int width, const awseg_fog_job* jobs, int n_jobs, const double* noise,
                                    const double* taps_host, double* depth_out, awseg_stream_t stream)
{
    return fog_common(0, nullptr, height, width, jobs, n_jobs, noise, taps_host, nullptr, nullptr, depth_out, nullptr,
                      nullptr, awseg_s(stream));
}

AWSEG_API int awseg_fog_fused(const uint8_t* imgs, int height, int width, const awseg_fog_job* jobs, int n_jobs,
                              const double* noise, const double* taps_host, uint8_t* out, float* norm_out,
                              double* depth_out, const float* mean_host, const float* std_host, awseg_stream_t stream)
{
    if (!out && !norm_out) return AWSEG_EINVAL;
    if (norm_out && (!mean_host || !std_host)) return AWSEG_EINVAL;
    return fog_common(1, imgs, height, width, jobs, n_jobs, noise, taps_host, out, norm_out, depth_out, mean_host,
                      std_host, awseg_s(stream));
}

AWSEG_API int awseg_fog_apply(const uint8_t* imgs, int height, int width, const awseg_fog_job* jobs, int n_jobs,
                              const double* depth, uint8_t* out, float* norm_out, const float* mean_host,
                              const float* std_host, awseg_stream_t stream)
{
    if (!imgs || !jobs || !depth || (!out && !norm_out) || height < 1 || width < 1 || n_jobs < 0) return AWSEG_EINVAL;
    if (norm_out && (!mean_host || !std_host)) return AWSEG_EINVAL;
    if (n_jobs == 0) return 0;
    if (n_jobs > 65535) return AWSEG_ERANGE;
    const int64_t hw = (int64_t)height * width;
    for (int j0 = 0; j0 < n_jobs; j0 += kMaxJobs) {
        const int cnt = n_jobs - j0 < kMaxJobs ? n_jobs - j0 : kMaxJobs;
        dim3 grid(grid_for(hw, cnt), cnt);
        hipLaunchKernelGGL(fog_apply_kernel, grid, dim3(kThreads), 0, awseg_s(stream), imgs, hw, pack_jobs(jobs, j0, cnt), j0, depth,
                           out, norm_out, make_nc(mean_host, std_host));
        AWSEG_LAUNCH_CHECK();
    }
    return 0;
}

AWSEG_API int awseg_night_apply(const uint8_t* imgs, int height, int width, const awseg_night_job* jobs, int n_jobs,
                                const double* noise, const float* gains_host, uint8_t* out, float* norm_out,
                                const float* mean_host, const float* std_host, awseg_stream_t stream)
{
    if (!imgs || !jobs || !gains_host || (!out && !norm_out) || height < 1 || width < 1 || n_jobs < 0) return AWSEG_EINVAL;
    if (norm_out && (!mean_host || !std_host)) return AWSEG_EINVAL;
    if (n_jobs == 0) return 0;
    if (n_jobs > 65535) return AWSEG_ERANGE;
    const int64_t hw = (int64_t)height * width;
    // 4 pixels per lane as dwords / float4 / double2 need aligned per-image bases (hw % 4 == 0 covers all three); any other
    // size the reference accepts (preprocessing.py:204-225 takes every H x W) takes the scalar accesses of the same kernel
    const bool vec = (hw & 3) == 0 && ((uintptr_t)imgs & 3) == 0 && (!out || ((uintptr_t)out & 3) == 0) &&
                     (!noise || ((uintptr_t)noise & 15) == 0) && (!norm_out || ((uintptr_t)norm_out & 15) == 0);
    norm_consts nc = make_nc(mean_host, std_host);
    for (int j0 = 0; j0 < n_jobs; j0 += kMaxJobs) {
        const int cnt = n_jobs - j0 < kMaxJobs ? n_jobs - j0 : kMaxJobs;
        const job_pack<awseg_night_job> pk = pack_jobs(jobs, j0, cnt);
        dim3 grid(grid_for((hw + 3) / 4, cnt), cnt);
        if (noise)
            hipLaunchKernelGGL((night_kernel<false>), grid, dim3(kThreads), 0, awseg_s(stream), imgs, hw, pk, j0, noise,
                               gains_host[0], gains_host[1], gains_host[2], out, norm_out, nc, vec);
        else
            hipLaunchKernelGGL((night_kernel<true>), grid, dim3(kThreads), 0, awseg_s(stream), imgs, hw, pk, j0, noise,
                               gains_host[0], gains_host[1], gains_host[2], out, norm_out, nc, vec);
        AWSEG_LAUNCH_CHECK();
    }
    return 0;
}

static int streak_common(bool snow, const uint8_t* imgs, int H, int W, const awseg_prim_job* jobs, int n_jobs,
                         const int32_t* prims, uint8_t* out, float* norm_out, const float* mean_host,
                         const float* std_host, void* workspace, hipStream_t s)
{
    if (workspace && ((uintptr_t)workspace & 3)) return AWSEG_EALIGN;
    uint32_t* bits = reinterpret_cast<uint32_t*>(workspace);
    const int wd = (W + 31) / 32;
    if (!imgs || !jobs || (!out && !norm_out) || H < 1 || W < 1 || n_jobs < 0) return AWSEG_EINVAL;
    if (norm_out && (!mean_host || !std_host)) return AWSEG_EINVAL;
    if (n_jobs == 0) return 0;
    if (n_jobs > 65535) return AWSEG_ERANGE;
    if (out == imgs) return AWSEG_EINVAL;            // the blur reads neighbours: not in-place safe
    if (((uintptr_t)imgs & 3) || (out && ((uintptr_t)out & 3)) || (norm_out && ((uintptr_t)norm_out & 15))) return AWSEG_EALIGN;
    if ((H + BTH - 1) / BTH > 65535) return AWSEG_ERANGE;
    norm_consts nc = make_nc(mean_host, std_host);
    // jobs are grouped by blur radius (snow draws 3x3 or 7x7 per frame, preprocessing.py:197)
    for (int pass = 0; pass < 2; ++pass) {
        const int want7 = pass;
        if (!snow && want7) break;
        awseg_prim_job sel[kMaxJobs];
        int cnt = 0;
        for (int j = 0; j <= n_jobs; ++j) {
            const bool take = j < n_jobs && ((snow && jobs[j].blur_ksize == 7) ? 1 : 0) == want7;
            if (take) sel[cnt++] = jobs[j];
            if (cnt == kMaxJobs || (j == n_jobs && cnt > 0)) {
                const job_pack<awseg_prim_job> pk = pack_jobs(sel, 0, cnt);
                dim3 grid((W + BTW - 1) / BTW, (H + BTH - 1) / BTH, cnt);
                if (bits) {
                    // the group's coverage maps live at the start of the workspace (groups run back to back on the stream)
                    int maxp = 0;
                    for (int q = 0; q < cnt; ++q) maxp = sel[q].prim_count > maxp ? sel[q].prim_count : maxp;
                    if (hipMemsetAsync(bits, 0, (size_t)cnt * H * wd * sizeof(uint32_t), s) != hipSuccess) return AWSEG_EINVAL;
                    if (maxp > 0) {
                        dim3 rgrid((maxp + 3) / 4, cnt);
                        if (snow) hipLaunchKernelGGL((raster_kernel<true>), rgrid, dim3(256), 0, s, H, W, pk, prims, bits, wd);
                        else hipLaunchKernelGGL((raster_kernel<false>), rgrid, dim3(256), 0, s, H, W, pk, prims, bits, wd);
                        AWSEG_LAUNCH_CHECK();
                    }
                    static const bool tiles = getenv("AWSEG_STREAK_TILES") != nullptr;   // measurements: the tile kernel on the coverage map
                    const int nr = want7 ? strip_rows<3>::value : strip_rows<1>::value;
                    dim3 sgrid((W + 255) / 256, (H + 4 * nr - 1) / (4 * nr), cnt);
                    if (tiles || sgrid.y > 65535 || want7) {            // 7x7 as a strip needs 222 registers (two waves per SIMD): 303 us against 161
                        if (!snow) hipLaunchKernelGGL((streak_kernel<false, 1, true>), grid, dim3(kSThreads), 0, s, imgs, H, W, pk, prims, make_blur(3, 0.5), out, norm_out, nc, bits, wd);
                        else if (!want7) hipLaunchKernelGGL((streak_kernel<true, 1, true>), grid, dim3(kSThreads), 0, s, imgs, H, W, pk, prims, make_blur(3, 1.0), out, norm_out, nc, bits, wd);
                        else hipLaunchKernelGGL((streak_kernel<true, 3, true>), grid, dim3(kSThreads), 0, s, imgs, H, W, pk, prims, make_blur(7, 1.0), out, norm_out, nc, bits, wd);
                    }
#define AWSEG_STRIP(SN, RV, TAPS)                                                                                                       \
    do {                                                                                                                              \
        if ((W & 3) == 0 && W >= 16) hipLaunchKernelGGL((streak_strip_kernel<SN, RV, true>), sgrid, dim3(256), 0, s, imgs, H, W, pk, TAPS, out, norm_out, nc, bits, wd); \
        else hipLaunchKernelGGL((streak_strip_kernel<SN, RV, false>), sgrid, dim3(256), 0, s, imgs, H, W, pk, TAPS, out, norm_out, nc, bits, wd);                          \
    } while (0)
                    else if (!snow) AWSEG_STRIP(false, 1, make_blur(3, 0.5));
                    else if (!want7) AWSEG_STRIP(true, 1, make_blur(3, 1.0));
#undef AWSEG_STRIP
                }
                else if (!snow) hipLaunchKernelGGL((streak_kernel<false, 1, false>), grid, dim3(kSThreads), 0, s, imgs, H, W, pk, prims, make_blur(3, 0.5), out, norm_out, nc, bits, wd);
                else if (!want7) hipLaunchKernelGGL((streak_kernel<true, 1, false>), grid, dim3(kSThreads), 0, s, imgs, H, W, pk, prims, make_blur(3, 1.0), out, norm_out, nc, bits, wd);
                else hipLaunchKernelGGL((streak_kernel<true, 3, false>), grid, dim3(kSThreads), 0, s, imgs, H, W, pk, prims, make_blur(7, 1.0), out, norm_out, nc, bits, wd);
                AWSEG_LAUNCH_CHECK();
                cnt = 0;
            }
        }
    }
    return 0;
}

AWSEG_API int64_t awseg_streak_workspace(int n_jobs, int height, int width)
{
    if (n_jobs < 1 || height < 1 || width < 1) return 0;
    const int64_t group = n_jobs < kMaxJobs ? n_jobs : kMaxJobs;     // frames of one launch group share the buffer's start
    return group * height * ((width + 31) / 32) * (int64_t)sizeof(uint32_t);
}

AWSEG_API int awseg_rain_apply(const uint8_t* imgs, int height, int width, const awseg_prim_job* jobs, int n_jobs,
                               const int32_t* drops, uint8_t* out, float* norm_out, const float* mean_host,
                               const float* std_host, void* workspace, awseg_stream_t stream)
{
    return streak_common(false, imgs, height, width, jobs, n_jobs, drops, out, norm_out, mean_host, std_host, workspace, awseg_s(stream));
}

AWSEG_API int awseg_snow_apply(const uint8_t* imgs, int height, int width, const awseg_prim_job* jobs, int n_jobs,
                               const int32_t* flakes, uint8_t* out, float* norm_out, const float* mean_host,
                               const float* std_host, void* workspace, awseg_stream_t stream)
{
    return streak_common(true, imgs, height, width, jobs, n_jobs, flakes, out, norm_out, mean_host, std_host, workspace, awseg_s(stream));
}

AWSEG_API int awseg_fog_density_field(const float* scale_offset, int batch, int64_t hw, uint64_t seed, const float* uniform,
                                      float* density, awseg_stream_t stream)
{
    if (!scale_offset || !density || batch < 1 || hw < 1) return AWSEG_EINVAL;
    if (batch > 65535) return AWSEG_ERANGE;
    dim3 grid(grid_for((hw + 3) / 4, batch), batch);
    hipLaunchKernelGGL(density_field_kernel, grid, dim3(kThreads), 0, awseg_s(stream), hw, seed, scale_offset, uniform, density);
    AWSEG_LAUNCH_CHECK();
    return 0;
}
