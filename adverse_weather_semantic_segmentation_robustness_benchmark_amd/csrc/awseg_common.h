// awseg_common.h — shared device/host helpers for libawseg_hip.so (gfx950 only).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>
#include "../../include/awseg.h"

#define AWSEG_API extern "C" __attribute__((visibility("default")))

#define AWSEG_WAVE 64
#define AWSEG_CUS 256

// Launch check: kernels are enqueued asynchronously, so only launch-time errors surface here.
#define AWSEG_LAUNCH_CHECK()                                  \
    do {                                                      \
        hipError_t e__ = hipGetLastError();                   \
        if (e__ != hipSuccess) return (int)e__;               \
    } while (0)

static inline hipStream_t awseg_s(awseg_stream_t s) { return (hipStream_t)s; }

// Memory-bound grids: cap at 8 resident 256-thread blocks per CU and grid-stride the rest
// (cdna_hip_programming.md Guideline 11).
static inline int awseg_grid_1d(int64_t work_items, int per_block, int max_blocks = AWSEG_CUS * 8)
{
    int64_t b = (work_items + per_block - 1) / per_block;
    if (b < 1) b = 1;
    if (b > max_blocks) b = max_blocks;
    return (int)b;
}

// 64-lane wavefront reductions (shuffle tree, no LDS).
__device__ __forceinline__ double awseg_wave_sum(double v)
{
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v += __shfl_down(v, o, 64);
    return v;
}
__device__ __forceinline__ float awseg_wave_min(float v)
{
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v = fminf(v, __shfl_down(v, o, 64));
    return v;
}
__device__ __forceinline__ float awseg_wave_max(float v)
{
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v = fmaxf(v, __shfl_down(v, o, 64));
    return v;
}

// label load for the two dtypes the reference produces (uint8 from the loader, int64 in its tests)
template <int DT> __device__ __forceinline__ int64_t awseg_ld_label(const void* p, int64_t i)
{
    if (DT == AWSEG_U8) return (int64_t)((const uint8_t*)p)[i];
    return ((const int64_t*)p)[i];
}

// Philox4x32-7 counter-based generator (Salmon et al., Random123: 7 rounds pass BigCrush) for the
// throughput-mode noise; parity mode takes host draws instead.  One call = four uint32.
struct awseg_philox {
    __device__ __forceinline__ static void gen(uint64_t seed, uint64_t ctr, uint32_t stream, uint32_t out[4])
    {
        const uint32_t M0 = 0xD2511F53u, M1 = 0xCD9E8D57u;
        uint32_t c0 = (uint32_t)ctr, c1 = (uint32_t)(ctr >> 32), c2 = stream, c3 = 0x9E3779B9u;
        uint32_t k0 = (uint32_t)seed, k1 = (uint32_t)(seed >> 32);
#pragma unroll
        for (int r = 0; r < 7; ++r) {
            uint32_t hi0 = __umulhi(M0, c0), lo0 = M0 * c0;
            uint32_t hi1 = __umulhi(M1, c2), lo1 = M1 * c2;
            uint32_t n0 = hi1 ^ c1 ^ k0, n2 = hi0 ^ c3 ^ k1;
            c0 = n0; c1 = lo1; c2 = n2; c3 = lo0;
            k0 += 0x9E3779B9u; k1 += 0xBB67AE85u;
        }
        out[0] = c0; out[1] = c1; out[2] = c2; out[3] = c3;
    }
};
// two uint32 -> one N(0,1) pair (Box-Muller in float32 on the hardware transcendental units:
// v_log_f32 is log2, v_sin/v_cos take their argument in revolutions).  Distribution-only parity.
__device__ __forceinline__ void awseg_box_muller(uint32_t a, uint32_t b, float& n0, float& n1)
{
    float u1 = ((float)(a >> 8) + 0.5f) * (1.0f / 16777216.0f);
    float u2 = (float)(b >> 8) * (1.0f / 16777216.0f);
    float r = __builtin_amdgcn_sqrtf(-1.38629436111989061883f * __builtin_amdgcn_logf(u1));   // sqrt(-2 ln u1), ln = log2 * ln2
    n0 = r * __builtin_amdgcn_cosf(u2);
    n1 = r * __builtin_amdgcn_sinf(u2);
}
// One uint32 -> one N(0,1) pair from its two 16-bit halves.  The noise these feed is quantised to
// 8 bits at sigma <= 1 LSB (night) or smoothed by a 17x17 Gaussian (fog depth), so 16-bit uniforms
// (tails truncated at 4.7 sigma) are ample and halve the Philox calls per pixel.
__device__ __forceinline__ void awseg_box_muller16(uint32_t a, float& n0, float& n1)
{
    float u1 = ((float)(a & 0xFFFFu) + 0.5f) * (1.0f / 65536.0f);
    float u2 = (float)(a >> 16) * (1.0f / 65536.0f);
    float r = __builtin_amdgcn_sqrtf(-1.38629436111989061883f * __builtin_amdgcn_logf(u1));
    n0 = r * __builtin_amdgcn_cosf(u2);
    n1 = r * __builtin_amdgcn_sinf(u2);
}
__device__ __forceinline__ float awseg_u01(uint32_t a) { return (float)(a >> 8) * (1.0f / 16777216.0f); }

// gemm_split3.hip (the LDS-DMA split-operand GEMM for N % 256 == 0, K % 32 == 0), called from gemm_split.hip
int awseg_gemm_split3_weights(const float* w, int n, int k, uint16_t* w3, const unsigned* trailer, hipStream_t stream);
bool awseg_gemm_split3_eligible(int64_t m, int n, int k, const void* x, const void* out, const void* residual, const void* bias);
// dual: the A operand continues in a second source behind column k1 (gemm_split3.hip g3_args: x2, K1, x2H, x2W, x2s, x2Ho, x2Wo)
struct awseg_g3_dual { const float* x2; int k1; int h, w, stride, ho, wo; int64_t bytes; const float* x3 = nullptr; const float* x4 = nullptr; };   // bytes: of ONE piece
int awseg_gemm_split3_launch(const float* x, const uint16_t* w3, const unsigned* trailer, const float* bias, const float* residual,
                             int act, float* out, int64_t m, int n, int k, int cus, hipStream_t stream, const int* conv = nullptr,
                             bool bf16 = false, const awseg_g3_dual* dual = nullptr);
int awseg_gemm_split3_bn(int n);
int64_t awseg_gemm_split3_image_rows(int n);
int awseg_gemm_bf16_3_weights(const float* w, int n, int k, uint16_t* w3, hipStream_t stream);
