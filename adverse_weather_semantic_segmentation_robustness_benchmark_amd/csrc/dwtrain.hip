// dwtrain.hip — weight (and bias) gradient of a depthwise 3x3 convolution on NHWC tensors, for the TRAINING step (BASELINE
// configs[3]: the MiT Mix-FFN's depthwise convolutions — transformers' SegformerDepthWiseConv behind PKG/models/model.py:120-130 —
// and the separable convolutions of the DeepLabV3+ ASPP / decoder behind :259-265; backward pass of PKG/training/trainer.py:299-353).
//
//     dW[tap][c] = sum over (b, y, x) of dy[b, y, x, c] * x[b, y + (ky - 1) d, x + (kx - 1) d, c]     (zero outside the frame)
//     db[c]      = sum over (b, y, x) of dy[b, y, x, c]
//
// MIOpen's immediate-mode pick for these layers is CK's batched-GEMM weight-gradient kernel: 14 calls of ~37 ms per 1024x2048
// training step — a quarter of the step — for a reduction that reads two maps once (profiles/r03_train_step_kernels.csv).  Here:
// block = (chunk of 2048 pixels, slice of 32 channel quads), 256 threads = 8 pixel lanes x 32 quads (a pixel's 128 channels are
// 512 contiguous bytes across the quad lanes); a thread keeps 9 + 1 float4 sums over its pixels, the eight pixel lanes meet through
// LDS, the block writes its partial sums, and a second small kernel adds the chunks in a fixed order (deterministic, no atomics).
// The forward and the input gradient of these layers are awseg_dwconv3x3_nhwc (backbone.hip) — the latter with the taps flipped.
#include "awseg_common.h"

namespace {

constexpr int DW_CHUNK = 2048;              // pixels per block
constexpr int DW_QS = 32;                   // channel quads per block slice

__global__ __launch_bounds__(256)
void dwconv3x3_wgrad_partial_kernel(const float* __restrict__ x, const float* __restrict__ dy, int64_t npix, int H, int W, int C, int dil,
                                    float* __restrict__ partial)
{
    __shared__ float4 s_acc[8][10][DW_QS];
    const int ql = threadIdx.x & (DW_QS - 1), pl = threadIdx.x / DW_QS;
    const int cq = blockIdx.y * DW_QS + ql;                         // channel quad
    const int cqn = C / 4;
    const bool live = cq < cqn;
    const int64_t p0 = (int64_t)blockIdx.x * DW_CHUNK;
    float4 acc[10];
#pragma unroll
    for (int k = 0; k < 10; ++k) acc[k] = make_float4(0.f, 0.f, 0.f, 0.f);
    if (live) {
        for (int i = pl; i < DW_CHUNK; i += 8) {
            const int64_t p = p0 + i;
            if (p >= npix) break;
            const int xx = (int)(p % W);
            const int64_t t = p / W;
            const int yy = (int)(t % H);
            const float4 g = *reinterpret_cast<const float4*>(dy + p * C + cq * 4);
            acc[9].x += g.x; acc[9].y += g.y; acc[9].z += g.z; acc[9].w += g.w;
#pragma unroll
            for (int ky = 0; ky < 3; ++ky) {
                const int sy = yy + (ky - 1) * dil;
                if (sy < 0 || sy >= H) continue;
#pragma unroll
                for (int kx = 0; kx < 3; ++kx) {
                    const int sx = xx + (kx - 1) * dil;
                    if (sx < 0 || sx >= W) continue;
                    const float4 v = *reinterpret_cast<const float4*>(x + (p + (int64_t)(ky - 1) * dil * W + (kx - 1) * dil) * C + cq * 4);
                    float4& a = acc[ky * 3 + kx];
                    a.x = fmaf(g.x, v.x, a.x); a.y = fmaf(g.y, v.y, a.y); a.z = fmaf(g.z, v.z, a.z); a.w = fmaf(g.w, v.w, a.w);
                }
            }
        }
    }
#pragma unroll
    for (int k = 0; k < 10; ++k) s_acc[pl][k][ql] = acc[k];
    __syncthreads();
    // 10 x 32 quads = 320 sums of 8 pixel lanes, fixed order
    for (int i = threadIdx.x; i < 10 * DW_QS; i += 256) {
        const int k = i / DW_QS, q = i - k * DW_QS;
        const int oq = blockIdx.y * DW_QS + q;
        if (oq >= cqn) continue;
        float4 s = s_acc[0][k][q];
#pragma unroll
        for (int l = 1; l < 8; ++l) { const float4 v = s_acc[l][k][q]; s.x += v.x; s.y += v.y; s.z += v.z; s.w += v.w; }
        *reinterpret_cast<float4*>(partial + ((int64_t)blockIdx.x * 10 + k) * C + oq * 4) = s;
    }
}

// dW9 [9][C], db [C] = sum over chunks of partial [chunks][10][C], in chunk order (float64 running sums)
__global__ __launch_bounds__(256)
void dwconv3x3_wgrad_fold_kernel(const float* __restrict__ partial, int nchunks, int C, float* __restrict__ dw9, float* __restrict__ db)
{
    const int i = blockIdx.x * 256 + threadIdx.x;                   // (k, c)
    if (i >= 10 * C) return;
    double s = 0.0;
    for (int ch = 0; ch < nchunks; ++ch) s += (double)partial[(int64_t)ch * 10 * C + i];
    if (i < 9 * C) dw9[i] = (float)s;
    else if (db) db[i - 9 * C] = (float)s;
}

}  // namespace

AWSEG_API int64_t awseg_dwconv3x3_wgrad_workspace(int64_t batch, int height, int width, int channels)
{
    if (batch < 1 || height < 1 || width < 1 || channels < 4) return 0;
    const int64_t chunks = (batch * height * width + DW_CHUNK - 1) / DW_CHUNK;
    return chunks * 10 * channels * (int64_t)sizeof(float);
}

AWSEG_API int awseg_dwconv3x3_wgrad_nhwc(const float* x, const float* dy, int64_t batch, int height, int width, int channels, int dilation,
                                         void* workspace, float* dw9, float* db, awseg_stream_t stream)
{
    if (batch == 0) return 0;
    if (!x || !dy || !workspace || !dw9 || batch < 0 || height < 1 || width < 1 || dilation < 1) return AWSEG_EINVAL;
    if (channels < 4 || (channels & 3)) return AWSEG_ERANGE;
    if (((uintptr_t)x & 15) || ((uintptr_t)dy & 15) || ((uintptr_t)workspace & 15)) return AWSEG_EALIGN;
    const int64_t npix = batch * height * width;
    const int64_t chunks = (npix + DW_CHUNK - 1) / DW_CHUNK;
    const int slices = (channels / 4 + DW_QS - 1) / DW_QS;
    if (chunks >= ((int64_t)1 << 31) || slices > 65535) return AWSEG_ERANGE;
    float* partial = reinterpret_cast<float*>(workspace);
    hipLaunchKernelGGL(dwconv3x3_wgrad_partial_kernel, dim3((unsigned)chunks, (unsigned)slices), dim3(256), 0, awseg_s(stream), x, dy, npix, height,
                       width, channels, dilation, partial);
    AWSEG_LAUNCH_CHECK();
    hipLaunchKernelGGL(dwconv3x3_wgrad_fold_kernel, dim3((10 * channels + 255) / 256), dim3(256), 0, awseg_s(stream), partial, (int)chunks, channels,
                       dw9, db);
    AWSEG_LAUNCH_CHECK();
    return 0;
}
