// bntrain.hip — BatchNorm2d (batch statistics) -> ReLU -> Dropout2d of the TRAINING graph as one forward pass and two backward
// passes over an NCHW map, for the full-resolution heads (PKG/models/model.py:152-158: Conv3x3 -> BatchNorm2d -> ReLU ->
// Dropout2d(0.1) -> Conv1x1 of the segmentation head; :42-52 of DepthEstimationHead) inside AdverseWeatherTrainer.train_epoch
// (PKG/training/trainer.py:299-353).  At 1024x2048, batch 8, the maps are 17 GB (256 channels) and 8.6 GB (128): as separate
// modules BatchNorm, ReLU and Dropout2d are ~7 passes forward and ~8 backward and keep three copies for autograd; here
//   statistics   sum, sum of squares per (image, channel) plane chunk in float64 -> mean, biased variance per channel
//   forward      out = max(x a_c + b_c, 0) * noise[b, c]          a = gamma invstd, b = beta - mean a           (1 read, 1 write)
//   backward 1   g' = g noise [x a + b > 0];  sum g', sum g' xhat per plane chunk (float64)                  (2 reads)
//   backward 2   dx = a (g' - mean(g') - xhat mean(g' xhat))                                                (2 reads, 1 write)
// with xhat = (x - mean) invstd recomputed from x — only x, the two per-channel vectors and the [B, C] noise are kept.
// Partial sums are folded in a fixed order on the host side of the launcher's second kernel: deterministic.
#include "awseg_common.h"

namespace {

constexpr int BN_T = 256;
constexpr int BN_CHUNK = 16384;             // floats of a plane per block

__device__ __forceinline__ double bn_block_sum(double v, double* red)
{
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v += __shfl_down(v, o, 64);
    const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
    if (lane == 0) red[wv] = v;
    __syncthreads();
    double s = 0.0;
    if (threadIdx.x == 0) { for (int i = 0; i < BN_T / 64; ++i) s += red[i]; }
    __syncthreads();
    return s;                                                        // valid in thread 0
}

// grid (chunks, C, B): partial[(c * B + b) * chunks + chunk][2] = {sum x, sum x^2}
__global__ __launch_bounds__(BN_T)
void bn_stats_partial_kernel(const float* __restrict__ x, int C, int64_t hw, double* __restrict__ partial)
{
    __shared__ double red[BN_T / 64];
    const int c = blockIdx.y, b = blockIdx.z, nch = gridDim.x;
    const float* p = x + ((int64_t)b * C + c) * hw;
    const int64_t lo = (int64_t)blockIdx.x * BN_CHUNK, hi = lo + BN_CHUNK < hw ? lo + BN_CHUNK : hw;
    double s = 0.0, q = 0.0;
    for (int64_t i = lo + threadIdx.x * 4; i < hi; i += BN_T * 4) {
        const float4 v = *reinterpret_cast<const float4*>(p + i);
        s += ((double)v.x + (double)v.y) + ((double)v.z + (double)v.w);
        q += ((double)v.x * v.x + (double)v.y * v.y) + ((double)v.z * v.z + (double)v.w * v.w);
    }
    const double ts = bn_block_sum(s, red), tq = bn_block_sum(q, red);
    if (threadIdx.x == 0) { double* o = partial + (((int64_t)c * gridDim.z + b) * nch + blockIdx.x) * 2; o[0] = ts; o[1] = tq; }
}

// one thread per channel: fold the partial pairs in index order -> out0[c], out1[c]
// MODE 0: mean, biased variance.  MODE 1: the two sums as they are (backward: sum g', sum g' xhat).
template <int MODE>
__global__ void bn_fold_kernel(const double* __restrict__ partial, int C, int per_channel, double count, float* __restrict__ out0, float* __restrict__ out1)
{
    const int c = blockIdx.x * blockDim.x + threadIdx.x;
    if (c >= C) return;
    double s = 0.0, q = 0.0;
    const double* p = partial + (int64_t)c * per_channel * 2;
    for (int i = 0; i < per_channel; ++i) { s += p[2 * i]; q += p[2 * i + 1]; }
    if (MODE == 0) {
        const double m = s / count;
        double var = q / count - m * m;
        out0[c] = (float)m; out1[c] = (float)(var > 0.0 ? var : 0.0);
    } else { out0[c] = (float)s; out1[c] = (float)q; }
}

// grid (chunks, C, B)
__global__ __launch_bounds__(BN_T)
void bn_relu_drop_fwd_kernel(const float* __restrict__ x, int C, int64_t hw, const float* __restrict__ mean, const float* __restrict__ invstd,
                             const float* __restrict__ gamma, const float* __restrict__ beta, const float* __restrict__ noise, float* __restrict__ out)
{
    const int c = blockIdx.y, b = blockIdx.z;
    const float a = gamma[c] * invstd[c], sh = beta[c] - mean[c] * a;
    const float nz = noise ? noise[(int64_t)b * C + c] : 1.0f;
    const int64_t base = ((int64_t)b * C + c) * hw;
    const int64_t lo = (int64_t)blockIdx.x * BN_CHUNK, hi = lo + BN_CHUNK < hw ? lo + BN_CHUNK : hw;
    for (int64_t i = lo + threadIdx.x * 4; i < hi; i += BN_T * 4) {
        const float4 v = *reinterpret_cast<const float4*>(x + base + i);
        float4 r;
        r.x = fmaxf(fmaf(v.x, a, sh), 0.f) * nz; r.y = fmaxf(fmaf(v.y, a, sh), 0.f) * nz;
        r.z = fmaxf(fmaf(v.z, a, sh), 0.f) * nz; r.w = fmaxf(fmaf(v.w, a, sh), 0.f) * nz;
        *reinterpret_cast<float4*>(out + base + i) = r;
    }
}

// backward pass 1: partial[(c * B + b) * chunks + chunk][2] = {sum g', sum g' xhat}
__global__ __launch_bounds__(BN_T)
void bn_relu_drop_bwd_reduce_kernel(const float* __restrict__ x, const float* __restrict__ g, int C, int64_t hw, const float* __restrict__ mean,
                                    const float* __restrict__ invstd, const float* __restrict__ gamma, const float* __restrict__ beta,
                                    const float* __restrict__ noise, double* __restrict__ partial)
{
    __shared__ double red[BN_T / 64];
    const int c = blockIdx.y, b = blockIdx.z, nch = gridDim.x;
    const float mu = mean[c], is = invstd[c];
    const float a = gamma[c] * is, sh = beta[c] - mu * a;
    const float nz = noise ? noise[(int64_t)b * C + c] : 1.0f;
    const int64_t base = ((int64_t)b * C + c) * hw;
    const int64_t lo = (int64_t)blockIdx.x * BN_CHUNK, hi = lo + BN_CHUNK < hw ? lo + BN_CHUNK : hw;
    double s = 0.0, q = 0.0;
    for (int64_t i = lo + threadIdx.x * 4; i < hi; i += BN_T * 4) {
        const float4 v = *reinterpret_cast<const float4*>(x + base + i);
        const float4 gv = *reinterpret_cast<const float4*>(g + base + i);
        const float xs[4] = {v.x, v.y, v.z, v.w}, gs[4] = {gv.x, gv.y, gv.z, gv.w};
#pragma unroll
        for (int k = 0; k < 4; ++k) {
            const float gp = fmaf(xs[k], a, sh) > 0.f ? gs[k] * nz : 0.f;
            s += (double)gp;
            q += (double)gp * (double)((xs[k] - mu) * is);
        }
    }
    const double ts = bn_block_sum(s, red), tq = bn_block_sum(q, red);
    if (threadIdx.x == 0) { double* o = partial + (((int64_t)c * gridDim.z + b) * nch + blockIdx.x) * 2; o[0] = ts; o[1] = tq; }
}

// backward pass 2: dx = a (g' - dbeta / N - xhat dgamma / N)
__global__ __launch_bounds__(BN_T)
void bn_relu_drop_bwd_dx_kernel(const float* __restrict__ x, const float* __restrict__ g, int C, int64_t hw, const float* __restrict__ mean,
                                const float* __restrict__ invstd, const float* __restrict__ gamma, const float* __restrict__ beta,
                                const float* __restrict__ noise, const float* __restrict__ dbeta, const float* __restrict__ dgamma, float inv_n,
                                float* __restrict__ dx)
{
    const int c = blockIdx.y, b = blockIdx.z;
    const float mu = mean[c], is = invstd[c];
    const float a = gamma[c] * is, sh = beta[c] - mu * a;
    const float nz = noise ? noise[(int64_t)b * C + c] : 1.0f;
    const float mb = dbeta[c] * inv_n, mg = dgamma[c] * inv_n;
    const int64_t base = ((int64_t)b * C + c) * hw;
    const int64_t lo = (int64_t)blockIdx.x * BN_CHUNK, hi = lo + BN_CHUNK < hw ? lo + BN_CHUNK : hw;
    for (int64_t i = lo + threadIdx.x * 4; i < hi; i += BN_T * 4) {
        const float4 v = *reinterpret_cast<const float4*>(x + base + i);
        const float4 gv = *reinterpret_cast<const float4*>(g + base + i);
        const float xs[4] = {v.x, v.y, v.z, v.w}, gs[4] = {gv.x, gv.y, gv.z, gv.w};
        float r[4];
#pragma unroll
        for (int k = 0; k < 4; ++k) {
            const float gp = fmaf(xs[k], a, sh) > 0.f ? gs[k] * nz : 0.f;
            const float xh = (xs[k] - mu) * is;
            r[k] = a * ((gp - mb) - xh * mg);
        }
        *reinterpret_cast<float4*>(dx + base + i) = make_float4(r[0], r[1], r[2], r[3]);
    }
}

// backward pass 2 with the result stored CHANNELS-LAST ([batch, hw, C] memory): what the layer in front of the full-resolution heads'
// BatchNorm wants — awseg_upconv3x3_adjoint reads its gradient as NHWC, and torch's permute + contiguous of the 17 GB map ran at
// 0.7 TB/s (8 x 6.6 ms + 4 x 3.9 ms per step).  Block = (256 pixels, 32 channels, image): loads along the planes, dx through an LDS tile,
// stores with 8 lanes per pixel (128 contiguous bytes).  C % 32 == 0, hw % 4 == 0.
__global__ __launch_bounds__(BN_T)
void bn_relu_drop_bwd_dx_nhwc_kernel(const float* __restrict__ x, const float* __restrict__ g, int C, int64_t hw, const float* __restrict__ mean,
                                     const float* __restrict__ invstd, const float* __restrict__ gamma, const float* __restrict__ beta,
                                     const float* __restrict__ noise, const float* __restrict__ dbeta, const float* __restrict__ dgamma, float inv_n,
                                     float* __restrict__ dx)
{
    __shared__ __attribute__((aligned(16))) float tile[256 * 36];
    const int b = blockIdx.z, c0 = blockIdx.y * 32;
    const int64_t p0 = (int64_t)blockIdx.x * 256;
    {
        const int cl = threadIdx.x >> 3, qg = threadIdx.x & 7;
        const int c = c0 + cl;
        const float mu = mean[c], is = invstd[c];
        const float a = gamma[c] * is, sh = beta[c] - mu * a;
        const float nz = noise ? noise[(int64_t)b * C + c] : 1.0f;
        const float mb = dbeta[c] * inv_n, mg = dgamma[c] * inv_n;
        const int64_t base = ((int64_t)b * C + c) * hw + p0;
#pragma unroll
        for (int j = 0; j < 8; ++j) {
            const int px = (qg + 8 * j) * 4;
            if (p0 + px >= hw) break;                                 // hw % 4 == 0: whole float4s
            const float4 v = *reinterpret_cast<const float4*>(x + base + px);
            const float4 gv = *reinterpret_cast<const float4*>(g + base + px);
            const float xs[4] = {v.x, v.y, v.z, v.w}, gs[4] = {gv.x, gv.y, gv.z, gv.w};
#pragma unroll
            for (int k = 0; k < 4; ++k) {
                const float gp = fmaf(xs[k], a, sh) > 0.f ? gs[k] * nz : 0.f;
                const float xh = (xs[k] - mu) * is;
                tile[(px + k) * 36 + cl] = a * ((gp - mb) - xh * mg);
            }
        }
    }
    __syncthreads();
    const int q = threadIdx.x & 7, pl = threadIdx.x >> 3;
#pragma unroll
    for (int it = 0; it < 8; ++it) {
        const int px = it * 32 + pl;
        if (p0 + px >= hw) break;
        *reinterpret_cast<float4*>(dx + ((int64_t)b * hw + p0 + px) * C + c0 + 4 * q) = *reinterpret_cast<const float4*>(tile + px * 36 + 4 * q);
    }
}

int bn_chunks(int64_t hw) { return (int)((hw + BN_CHUNK - 1) / BN_CHUNK); }

}  // namespace

AWSEG_API int64_t awseg_bn_train_workspace(int batch, int channels, int64_t hw)
{
    if (batch < 1 || channels < 1 || hw < 1) return 0;
    return (int64_t)channels * batch * bn_chunks(hw) * 2 * (int64_t)sizeof(double);
}

static int bn_check(const void* x, int batch, int channels, int64_t hw, const void* ws)
{
    if (!x || !ws || batch < 1 || channels < 1 || hw < 4) return AWSEG_EINVAL;
    if ((hw & 3) || batch > 65535 || channels > 65535) return AWSEG_ERANGE;
    if (((uintptr_t)x & 15) || ((uintptr_t)ws & 15)) return AWSEG_EALIGN;
    return 0;
}

AWSEG_API int awseg_bn_train_stats(const float* x, int batch, int channels, int64_t hw, void* workspace, float* mean, float* var, awseg_stream_t stream)
{
    if (int rc = bn_check(x, batch, channels, hw, workspace)) return rc;
    if (!mean || !var) return AWSEG_EINVAL;
    const int nch = bn_chunks(hw);
    double* partial = reinterpret_cast<double*>(workspace);
    hipLaunchKernelGGL(bn_stats_partial_kernel, dim3(nch, channels, batch), dim3(BN_T), 0, awseg_s(stream), x, channels, hw, partial);
    AWSEG_LAUNCH_CHECK();
    hipLaunchKernelGGL(bn_fold_kernel<0>, dim3((channels + 63) / 64), dim3(64), 0, awseg_s(stream), partial, channels, batch * nch,
                       (double)batch * (double)hw, mean, var);
    AWSEG_LAUNCH_CHECK();
    return 0;
}

AWSEG_API int awseg_bn_relu_dropout_forward(const float* x, int batch, int channels, int64_t hw, const float* mean, const float* invstd,
                                            const float* gamma, const float* beta, const float* noise, float* out, awseg_stream_t stream)
{
    if (!x || !out || !mean || !invstd || !gamma || !beta || batch < 1 || channels < 1 || hw < 4) return AWSEG_EINVAL;
    if ((hw & 3) || batch > 65535 || channels > 65535) return AWSEG_ERANGE;
    if (((uintptr_t)x & 15) || ((uintptr_t)out & 15)) return AWSEG_EALIGN;
    hipLaunchKernelGGL(bn_relu_drop_fwd_kernel, dim3(bn_chunks(hw), channels, batch), dim3(BN_T), 0, awseg_s(stream), x, channels, hw, mean, invstd,
                       gamma, beta, noise, out);
    AWSEG_LAUNCH_CHECK();
    return 0;
}

AWSEG_API int awseg_bn_relu_dropout_backward(const float* x, const float* grad_out, int batch, int channels, int64_t hw, const float* mean,
                                             const float* invstd, const float* gamma, const float* beta, const float* noise, void* workspace,
                                             float* dgamma, float* dbeta, float* dx, int dx_channels_last, awseg_stream_t stream)
{
    if (int rc = bn_check(x, batch, channels, hw, workspace)) return rc;
    if (!grad_out || !mean || !invstd || !gamma || !beta || !dgamma || !dbeta || !dx) return AWSEG_EINVAL;
    if (((uintptr_t)grad_out & 15) || ((uintptr_t)dx & 15)) return AWSEG_EALIGN;
    const int nch = bn_chunks(hw);
    double* partial = reinterpret_cast<double*>(workspace);
    dim3 grid(nch, channels, batch);
    hipLaunchKernelGGL(bn_relu_drop_bwd_reduce_kernel, grid, dim3(BN_T), 0, awseg_s(stream), x, grad_out, channels, hw, mean, invstd, gamma, beta,
                       noise, partial);
    AWSEG_LAUNCH_CHECK();
    hipLaunchKernelGGL(bn_fold_kernel<1>, dim3((channels + 63) / 64), dim3(64), 0, awseg_s(stream), partial, channels, batch * nch, 1.0, dbeta, dgamma);
    AWSEG_LAUNCH_CHECK();
    const float inv_n = (float)(1.0 / ((double)batch * (double)hw));
    if (dx_channels_last) {
        if (channels % 32) return AWSEG_ERANGE;
        if ((hw + 255) / 256 >= ((int64_t)1 << 31)) return AWSEG_ERANGE;
        hipLaunchKernelGGL(bn_relu_drop_bwd_dx_nhwc_kernel, dim3((unsigned)((hw + 255) / 256), channels / 32, batch), dim3(BN_T), 0, awseg_s(stream), x,
                           grad_out, channels, hw, mean, invstd, gamma, beta, noise, dbeta, dgamma, inv_n, dx);
        AWSEG_LAUNCH_CHECK();
        return 0;
    }
    hipLaunchKernelGGL(bn_relu_drop_bwd_dx_kernel, grid, dim3(BN_T), 0, awseg_s(stream), x, grad_out, channels, hw, mean, invstd, gamma, beta, noise,
                       dbeta, dgamma, inv_n, dx);
    AWSEG_LAUNCH_CHECK();
    return 0;
}
