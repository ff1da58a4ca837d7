// smallops.hip — small passes of the eval step that used to run as library calls (hipBLASLt GEMMs of 8 rows, an addmm with one
// output column, separate add / sigmoid passes, a strided torch copy):
//   * awseg_rowdot_sigmoid   : out[r] = sigmoid(x[r, :] . w + b) — the 1x1 convolution to ONE channel + Sigmoid that ends
//                              DepthEstimationHead (PKG/models/model.py:49-51) on the stride-16 map of the DeepLab member (:368);
//   * awseg_aspp_pool_branch : smp's ASPPPooling branch after its global mean — 1x1 conv + BatchNorm + ReLU on one row per image —
//                              followed by that branch's slice of the ASPP projection (the smp model built at model.py:262-268):
//                              out[b, :] = relu(mean[b, :] W1^T + b1) W2^T + b2: two small launches (64 blocks each);
//   * awseg_stem_image       : the planar frames into the zero-padded 4-channel NHWC image both 7x7 stems gather their rows from.
// The first two are a few hundred KB of weights against a few rows: bound by latency, not by any pipe; the third is one pass at the
// HBM rate (torch's strided copy ran at a third of it).
#include "awseg_common.h"

namespace {

constexpr int SO_T = 256;

__global__ __launch_bounds__(SO_T)
void rowdot_sigmoid_kernel(const float* __restrict__ x, int64_t rows, int k, const float* __restrict__ w, const float* __restrict__ bias,
                           int sigmoid, float* __restrict__ out)
{
    // 16 lanes per row, float4 per lane per step: a 128-wide row is two steps
    const int sub = threadIdx.x & 15;
    const int64_t r = ((int64_t)blockIdx.x * SO_T + threadIdx.x) >> 4;
    float acc = 0.f;
    if (r < rows) {
        const float4* xr = reinterpret_cast<const float4*>(x + r * k);
        const float4* w4 = reinterpret_cast<const float4*>(w);
        for (int j = sub; j < k / 4; j += 16) {
            const float4 a = xr[j], b = w4[j];
            acc = fmaf(a.x, b.x, acc); acc = fmaf(a.y, b.y, acc); acc = fmaf(a.z, b.z, acc); acc = fmaf(a.w, b.w, acc);
        }
    }
#pragma unroll
    for (int o = 8; o >= 1; o >>= 1) acc += __shfl_xor(acc, o, 16);
    if (r < rows && sub == 0) {
        const float v = acc + (bias ? bias[0] : 0.f);
        out[r] = sigmoid ? 1.0f / (1.0f + expf(-v)) : v;
    }
}

// planar frames [B, C <= 4, H, W] (any strides whose innermost is 1) -> the interior columns 3 .. 3 + W - 1 of the zero-padded
// 4-channel NHWC image [B, H, Wp, 4] both 7x7 stems gather their rows from; channels >= C of the interior are written as zeros, the
// padding columns are not touched (zero since the buffer was made).  A wave takes 256 consecutive pixels of a row, lane i the pixels
// i, 64 + i, 128 + i, 192 + i: every load instruction reads 256 contiguous bytes of a plane and every store instruction writes 1 KB of
// contiguous pixels (4 consecutive pixels per lane would make each store touch 64 lines a quarter full).
__global__ __launch_bounds__(SO_T)
void stem_image_kernel(const float* __restrict__ x, int B, int C, int H, int W, int64_t sb, int64_t sc, int64_t sy, float* __restrict__ img, int Wp)
{
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int spans = (W + 255) / 256;                             // 256-pixel spans per row
    const int64_t total = (int64_t)B * H * spans;
    for (int64_t u = (int64_t)blockIdx.x * (SO_T / 64) + wave; u < total; u += (int64_t)gridDim.x * (SO_T / 64)) {
        const int sp = (int)(u % spans);
        const int64_t t = u / spans;
        const int y = (int)(t % H), b = (int)(t / H);
        const float* src = x + b * sb + y * sy + sp * 256 + lane;
        float v[4][4];
#pragma unroll
        for (int c = 0; c < 4; ++c)
#pragma unroll
            for (int k = 0; k < 4; ++k) v[c][k] = (c < C && sp * 256 + 64 * k + lane < W) ? src[c * sc + 64 * k] : 0.f;
        float* dst = img + (((int64_t)b * H + y) * Wp + 3 + sp * 256 + lane) * 4;
#pragma unroll
        for (int k = 0; k < 4; ++k)
            if (sp * 256 + 64 * k + lane < W) *reinterpret_cast<float4*>(dst + 256 * k) = make_float4(v[0][k], v[1][k], v[2][k], v[3][k]);
    }
}

constexpr int PB_MC = 4;        // mid channels per block
constexpr int PB_B = 8;         // images per pass

__global__ __launch_bounds__(SO_T)
void aspp_pool_branch_kernel(const float* __restrict__ mean, int batch, int cin, const float* __restrict__ w1, const float* __restrict__ b1,
                             int cmid, float* __restrict__ g_ws)
{
    __shared__ float sRed[SO_T / 64][PB_MC * PB_B];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int c0 = blockIdx.x * PB_MC;
    for (int bb = 0; bb < batch; bb += PB_B) {
        float acc[PB_MC][PB_B];
#pragma unroll
        for (int c = 0; c < PB_MC; ++c)
#pragma unroll
            for (int b = 0; b < PB_B; ++b) acc[c][b] = 0.f;
        for (int j = tid; j < cin; j += SO_T) {
            float wv[PB_MC];
#pragma unroll
            for (int c = 0; c < PB_MC; ++c) wv[c] = c0 + c < cmid ? w1[(int64_t)(c0 + c) * cin + j] : 0.f;
#pragma unroll
            for (int b = 0; b < PB_B; ++b) {
                const float m = bb + b < batch ? mean[(int64_t)(bb + b) * cin + j] : 0.f;
#pragma unroll
                for (int c = 0; c < PB_MC; ++c) acc[c][b] = fmaf(wv[c], m, acc[c][b]);
            }
        }
#pragma unroll
        for (int c = 0; c < PB_MC; ++c)
#pragma unroll
            for (int b = 0; b < PB_B; ++b) {
                float v = acc[c][b];
#pragma unroll
                for (int o = 32; o >= 1; o >>= 1) v += __shfl_xor(v, o, 64);
                if (lane == 0) sRed[wave][c * PB_B + b] = v;
            }
        __syncthreads();
        if (tid < PB_MC * PB_B) {
            const int c = tid / PB_B, b = tid - c * PB_B;
            if (c0 + c < cmid && bb + b < batch) {
                float v = b1[c0 + c];
#pragma unroll
                for (int wv = 0; wv < SO_T / 64; ++wv) v += sRed[wv][tid];
                g_ws[(int64_t)(bb + b) * cmid + c0 + c] = v > 0.f ? v : 0.f;
            }
        }
        __syncthreads();
    }
}

// second half: out[b, o] = g[b, :] . w2[o, :] + b2[o] — a block per PB_MC output channels (one per wave), g through LDS
__global__ __launch_bounds__(SO_T)
void aspp_pool_proj_kernel(const float* __restrict__ g, int batch, int cmid, const float* __restrict__ w2, const float* __restrict__ b2,
                           int cout, float* __restrict__ out)
{
    extern __shared__ float sG[];                                  // [PB_B][cmid]
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int o = blockIdx.x * (SO_T / 64) + wave;
    for (int bb = 0; bb < batch; bb += PB_B) {
        const int nb = batch - bb < PB_B ? batch - bb : PB_B;
        __syncthreads();
        for (int i = tid; i < nb * cmid; i += SO_T) sG[i] = g[(int64_t)bb * cmid + i];
        __syncthreads();
        if (o >= cout) continue;
        float v[PB_B];
#pragma unroll
        for (int b = 0; b < PB_B; ++b) v[b] = 0.f;
        for (int c = lane; c < cmid; c += 64) {
            const float wv = w2[(int64_t)o * cmid + c];
#pragma unroll
            for (int b = 0; b < PB_B; ++b) if (b < nb) v[b] = fmaf(sG[b * cmid + c], wv, v[b]);
        }
#pragma unroll
        for (int b = 0; b < PB_B; ++b) {
#pragma unroll
            for (int s = 32; s >= 1; s >>= 1) v[b] += __shfl_xor(v[b], s, 64);
            if (lane == 0 && b < nb) out[(int64_t)(bb + b) * cout + o] = v[b] + (b2 ? b2[o] : 0.f);
        }
    }
}

}  // namespace

AWSEG_API int awseg_rowdot_sigmoid(const float* x, int64_t rows, int k, const float* w, const float* bias, int sigmoid, float* out,
                                   awseg_stream_t stream)
{
    if (rows == 0) return 0;
    if (!x || !w || !out || rows < 0 || k < 4) return AWSEG_EINVAL;
    if (k % 4) return AWSEG_ERANGE;
    if (((uintptr_t)x & 15) || ((uintptr_t)w & 15)) return AWSEG_EALIGN;
    const int64_t blocks = (rows * 16 + SO_T - 1) / SO_T;
    if (blocks > 0x7fffffff) return AWSEG_ERANGE;
    hipLaunchKernelGGL(rowdot_sigmoid_kernel, dim3((unsigned)blocks), dim3(SO_T), 0, awseg_s(stream), x, rows, k, w, bias, sigmoid ? 1 : 0, out);
    AWSEG_LAUNCH_CHECK();
    return 0;
}

AWSEG_API int64_t awseg_aspp_pool_branch_workspace(int batch, int cmid)
{
    return batch < 0 || cmid < 0 ? 0 : (int64_t)batch * cmid * 4;
}

AWSEG_API int awseg_aspp_pool_branch(const float* mean, int batch, int cin, const float* w1, const float* b1, int cmid, const float* w2,
                                     const float* b2, int cout, void* workspace, float* out, awseg_stream_t stream)
{
    if (batch == 0 || cout == 0) return 0;
    if (!mean || !w1 || !b1 || !w2 || !workspace || !out || batch < 0 || cin < 1 || cmid < 1 || cout < 0) return AWSEG_EINVAL;
    if ((size_t)PB_B * cmid * sizeof(float) > 48 * 1024) return AWSEG_ERANGE;
    float* g = reinterpret_cast<float*>(workspace);               // relu(mean w1^T + b1), float32 [batch][cmid]
    hipLaunchKernelGGL(aspp_pool_branch_kernel, dim3((unsigned)((cmid + PB_MC - 1) / PB_MC)), dim3(SO_T), 0, awseg_s(stream), mean, batch, cin,
                       w1, b1, cmid, g);
    AWSEG_LAUNCH_CHECK();
    hipLaunchKernelGGL(aspp_pool_proj_kernel, dim3((unsigned)((cout + SO_T / 64 - 1) / (SO_T / 64))), dim3(SO_T), (size_t)PB_B * cmid * sizeof(float),
                       awseg_s(stream), g, batch, cmid, w2, b2, cout, out);
    AWSEG_LAUNCH_CHECK();
    return 0;
}

AWSEG_API int awseg_stem_image(const float* x, int batch, int channels, int height, int width, int64_t stride_b, int64_t stride_c,
                               int64_t stride_y, float* image, int padded_width, awseg_stream_t stream)
{
    if (batch == 0) return 0;
    if (!x || !image || batch < 0 || channels < 1 || channels > 4 || height < 1 || width < 1 || padded_width < width + 3) return AWSEG_EINVAL;
    if (stride_b < 0 || stride_c < 0 || stride_y < width) return AWSEG_EINVAL;
    if ((uintptr_t)image & 15) return AWSEG_EALIGN;
    const int64_t total = (int64_t)batch * height * ((width + 255) / 256) * 64;      // a wave per 256-pixel span
    hipLaunchKernelGGL(stem_image_kernel, dim3(awseg_grid_1d(total, SO_T)), dim3(SO_T), 0, awseg_s(stream), x, batch, channels, height, width,
                       stride_b, stride_c, stride_y, image, padded_width);
    AWSEG_LAUNCH_CHECK();
    return 0;
}
