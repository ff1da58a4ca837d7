// wino.hip — 3x3 stride-1 "same" convolutions of the eval forward as Winograd F(2x2,3x3) on the
// fp32 matrix cores (v_mfma_f32_32x32x2_f32): 16 multiplies per 2x2 outputs instead of 36, i.e.
// 2.25x fewer MFMA flops than the direct (implicit-GEMM) form whose ceiling is the 157 TFLOP/s
// fp32 MFMA peak.  Used for the depth heads (PKG/models/model.py:42-52: Conv3x3 -> BN -> ReLU ->
// Conv1x1 -> Sigmoid, on the SegFormer branch at FULL resolution, :219-221) and the ResNet
// bottleneck 3x3s (incl. the dilated layer4 — a dilation-d convolution is d*d independent
// undilated convolutions on the pixel sub-grids y%d, x%d).
//
//   V_p = (B^T d B)_p   per 4x4 input tile, per input channel          (VALU, into LDS)
//   M_p = V_p @ U_p     16 GEMMs [tiles x Cin] x [Cin x Cout]          (MFMA, U = G g G^T from the host,
//                                                                       BatchNorm scale folded in)
//   Y   = A^T M A       2x2 outputs per tile                           (in-lane: a lane's 16 accumulators
//                                                                       of one (tile, cout) are the 16 p's)
// Block = 8x8 tiles (16x16 outputs) x 64 output channels, 4 waves: wave (mt, nt) owns tiles
// 32*mt.. and couts 32*nt.., all 16 positions -> 16 accumulator tiles of 32x32 (256 registers per
// lane, one wave per SIMD).  Input channels stream through LDS in chunks of 16; the next chunk's
// raw pixels and weights are prefetched into registers while the matrix cores run on the current one.
// Epilogues: FULL  out[b,y,x,n] = act(Y + shift[n] (+ residual))            (NHWC)
//            HEAD1 out[b,y,x]   = sigmoid(b2 + sum_n w2[n] * relu(Y + shift[n]))   (Cout == 64)
#include "awseg_common.h"

namespace {

typedef float f32x16 __attribute__((ext_vector_type(16)));

constexpr int WT = 256;            // threads per block
constexpr int TB = 8;              // 8 x 8 tiles per block
constexpr int NTILE = TB * TB;     // 64
constexpr int KC = 8;              // input channels per chunk
constexpr int NB = 64;             // output channels per block
constexpr int V_FLOATS = 16 * KC * NTILE;   // 32 KB per buffer
constexpr int U_FLOATS = 16 * KC * NB;      // 32 KB per buffer
constexpr int PW = 2 * TB + 2;                // 18 x 18 input pixels feed the block's 8 x 8 tiles
constexpr int P_UNITS = 11 * 64;              // 16-byte units per patch buffer: 18*18 px x (KC*4/16) = 648, rounded up to whole wave instructions
constexpr int P_FLOATS = P_UNITS * 4;         // 11 KB per buffer
constexpr int LDS_FLOATS = 2 * V_FLOATS + 2 * U_FLOATS + 2 * P_FLOATS;   // 150 KB: operands and the raw patch double-buffered

struct wino_args {
    const float* x; const float* U; const float* shift; const float* residual; const float* w2; const float* b2;
    float* out;
    int H, W, Cin, Cout, dil, act, nbx, nby, ngroups;
};

__device__ __forceinline__ float act_apply(float v, int act)
{
    if (act == AWSEG_ACT_RELU) return v > 0.f ? v : 0.f;
    return v;
}

__device__ __forceinline__ float2 f2sub(float2 a, float2 b) { return make_float2(a.x - b.x, a.y - b.y); }
__device__ __forceinline__ float2 f2add(float2 a, float2 b) { return make_float2(a.x + b.x, a.y + b.y); }

// V = B^T d B of one (tile, channel pair) in two stages so the work can be spread between the MFMA
// groups of a step: rows first (t = B^T d, pixels outside the image masked to zero), then one output
// row i of (t B) -> positions 4i..4i+3, written as sV[p][2*pair + comp][tile'] where odd channel
// rows are rotated by 32 tiles: an MFMA A-fragment read takes rows k and k+1 in the two wave halves,
// and the rotation puts them on disjoint LDS banks.
__device__ __forceinline__ void xform_rows(const float* __restrict__ patch, uint32_t mask, float2 (&t)[16])
{
    float2 r[16];
#pragma unroll
    for (int i = 0; i < 4; ++i)
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            const float2 v = *reinterpret_cast<const float2*>(patch + (i * PW + j) * KC);
            const bool ok = (mask >> (i * 4 + j)) & 1u;
            r[i * 4 + j] = make_float2(ok ? v.x : 0.f, ok ? v.y : 0.f);
        }
#pragma unroll
    for (int j = 0; j < 4; ++j) {
        t[0 * 4 + j] = f2sub(r[0 * 4 + j], r[2 * 4 + j]);
        t[1 * 4 + j] = f2add(r[1 * 4 + j], r[2 * 4 + j]);
        t[2 * 4 + j] = f2sub(r[2 * 4 + j], r[1 * 4 + j]);
        t[3 * 4 + j] = f2sub(r[1 * 4 + j], r[3 * 4 + j]);
    }
}
__device__ __forceinline__ void xform_cols_store(const float2 (&t)[16], int i, float* __restrict__ d0, float* __restrict__ d1)
{
    float2 v[4];
    v[0] = f2sub(t[i * 4 + 0], t[i * 4 + 2]);
    v[1] = f2add(t[i * 4 + 1], t[i * 4 + 2]);
    v[2] = f2sub(t[i * 4 + 2], t[i * 4 + 1]);
    v[3] = f2sub(t[i * 4 + 1], t[i * 4 + 3]);
#pragma unroll
    for (int j = 0; j < 4; ++j) {
        d0[(i * 4 + j) * KC * NTILE] = v[j].x;
        d1[(i * 4 + j) * KC * NTILE] = v[j].y;
    }
}

template <int MODE>   // 0 FULL, 1 HEAD1
__global__ __launch_bounds__(WT, 1)
void conv3x3_wino_kernel(wino_args a)
{
    extern __shared__ float smem[];
    float* sV = smem;                        // [2][16][KC][NTILE]
    float* sU = smem + 2 * V_FLOATS;         // [2][16][KC][NB]
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int bx = blockIdx.x % a.nbx, rx = blockIdx.x / a.nbx;
    const int by = blockIdx.y % a.nby, ry = blockIdx.y / a.nby;
    const int ng = blockIdx.z % a.ngroups, b = blockIdx.z / a.ngroups;
    const int Hs = (a.H - ry + a.dil - 1) / a.dil, Ws = (a.W - rx + a.dil - 1) / a.dil;   // sub-grid extent of this residue
    if (by * 2 * TB >= Hs || bx * 2 * TB >= Ws) return;
    const float* xb = a.x + (int64_t)b * a.H * a.W * a.Cin;
    const int n0 = ng * NB;
    const int nchunks = a.Cin / KC;          // even (Cin % 16 == 0)

    float* sP = smem + 2 * V_FLOATS + 2 * U_FLOATS;   // [2][18*18 px][KC] raw input patch of a chunk
    // ---- raw patch: the 18 x 18 pixels x KC channels the block's tiles are cut from, LDS-DMA'd in
    // 16-byte units (2 per pixel), unit q = (py*18 + px)*2 + half -> LDS slot q (lane-linear).  Fetching
    // every pixel once (instead of once per overlapping tile and channel pair straight into registers)
    // is what keeps the texture-address path off the critical path: a scattered per-lane load costs it
    // one request per lane.  Pixels outside the image fetch a clamped address; the mask zeroes them.
    uint32_t punit[3];
#pragma unroll
    for (int i = 0; i < 3; ++i) {
        int q = (wave + 4 * i) * 64 + lane;
        if (q >= PW * PW * 2) q = PW * PW * 2 - 1;                 // tail lanes of the 11th instruction re-fetch the last unit
        const int pix = q >> 1, py = pix / PW, px = pix - py * PW;
        const int sy = by * 2 * TB - 1 + py, sx = bx * 2 * TB - 1 + px;
        int y = ry + a.dil * sy, x = rx + a.dil * sx;
        const bool ok = sy >= 0 && sx >= 0 && y < a.H && x < a.W;
        y = ok ? y : 0; x = ok ? x : 0;
        punit[i] = (uint32_t)((y * a.W + x) * a.Cin + (q & 1) * 4);
    }
    const int n_pinstr = wave < 3 ? 3 : 2;                          // 11 wave instructions over 4 waves
    auto glds_patch = [&](int chunk, float* dstbuf) {
        const float* base = xb + (chunk < nchunks ? chunk : nchunks - 1) * KC;
#pragma unroll
        for (int i = 0; i < 3; ++i)
            if (i < n_pinstr)
                __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)(base + punit[i]),
                                                 (__attribute__((address_space(3))) void*)(dstbuf + (wave + 4 * i) * 256), 16, 0, 0);
    };
    // transform item of this thread: tile = lane, channel pair = wave; validity of its 4 x 4 pixels
    uint32_t pmask = 0;
    {
        const int uy0 = by * 2 * TB + 2 * (lane >> 3), ux0 = bx * 2 * TB + 2 * (lane & 7);
#pragma unroll
        for (int i = 0; i < 4; ++i)
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                const int sy = uy0 - 1 + i, sx = ux0 - 1 + j;
                const bool ok = sy >= 0 && sx >= 0 && ry + a.dil * sy < a.H && rx + a.dil * sx < a.W;
                pmask |= (ok ? 1u : 0u) << (i * 4 + j);
            }
    }
    const int poff = ((2 * (lane >> 3)) * PW + 2 * (lane & 7)) * KC + wave * 2;   // this tile's top-left pixel, this pair
    // ---- weight chunk: 128 rows (p, k) x 64 couts, LDS-DMA'd 4 rows (1 KiB) per wave instruction,
    // 8 instructions per wave.  Odd-k rows are stored rotated by 32 couts (see xform_cols_store):
    // the LDS image is lane-linear, so the rotation goes on the per-lane SOURCE address.
    uint32_t uoff[8];
#pragma unroll
    for (int i = 0; i < 8; ++i) {
        const int row = (wave * 8 + i) * 4 + (lane >> 4);      // p * KC + k
        const int p = row >> 3, k = row & 7;
        uoff[i] = (uint32_t)((p * a.Cin + k) * a.Cout + n0 + 4 * ((lane & 15) ^ ((k & 1) << 3)));
    }
    auto glds_u = [&](int chunk, float* dstbuf) {
        const float* base = a.U + (int64_t)(chunk < nchunks ? chunk : nchunks - 1) * KC * a.Cout;
#pragma unroll
        for (int i = 0; i < 8; ++i)
            __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)(base + uoff[i]),
                                             (__attribute__((address_space(3))) void*)(dstbuf + (wave * 8 + i) * 256), 16, 0, 0);
    };

    f32x16 acc[16];
#pragma unroll
    for (int p = 0; p < 16; ++p)
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[p][r] = 0.f;

    const int mt = wave >> 1, nt = wave & 1, hk = lane >> 5, li = lane & 31;
    const int aoff = hk * NTILE + ((mt * 32 + li) ^ (hk << 5));
    const int boff = hk * NB + ((nt * 32 + li) ^ (hk << 5));
    const int d0off = (wave * 2) * NTILE + lane, d1off = (wave * 2 + 1) * NTILE + (lane ^ 32);

    glds_u(0, sU);
    glds_patch(0, sP);
    glds_patch(1, sP + P_FLOATS);
    __syncthreads();
    {
        float2 t[16];
        xform_rows(sP + poff, pmask, t);
#pragma unroll
        for (int i = 0; i < 4; ++i) xform_cols_store(t, i, sV + d0off, sV + d1off);
    }
    __syncthreads();

    // One step = the MFMAs of chunk c on buffers BUF while chunk c+1 is staged into the other ones:
    // its patch (landed during the previous step) is transformed into V[BUF^1], its weights are
    // LDS-DMA'd into U[BUF^1], and the patch buffer the previous step's transform freed is refilled
    // with chunk c+2.  No conditionals: past the end the loads re-read the last chunk into buffers
    // nobody consumes.  The four k-steps are software-pipelined by hand (operands of k-step s+1 are
    // read while the MFMAs of s issue; a quarter of the transform rides in each group).
#define WINO_LOAD_OPS(S, AV, BV)                                                               \
    _Pragma("unroll") for (int p = 0; p < 16; ++p) {                                           \
        AV[p] = pa[(p * KC + 2 * (S)) * NTILE];                                                \
        BV[p] = pb[(p * KC + 2 * (S)) * NB];                                                   \
    }
#define WINO_MFMAS(AV, BV)                                                                     \
    _Pragma("unroll") for (int p = 0; p < 16; ++p)                                             \
        acc[p] = __builtin_amdgcn_mfma_f32_32x32x2f32(AV[p], BV[p], acc[p], 0, 0, 0);
#define WINO_STEP(C, BUF)                                                                      \
    do {                                                                                       \
        const float* pa = sV + (BUF) * V_FLOATS + aoff;                                        \
        const float* pb = sU + (BUF) * U_FLOATS + boff;                                        \
        float* vd0 = sV + ((BUF) ^ 1) * V_FLOATS + d0off;                                      \
        float* vd1 = sV + ((BUF) ^ 1) * V_FLOATS + d1off;                                      \
        float a0[16], b0[16], a1[16], b1[16];                                                  \
        float2 t[16];                                                                          \
        WINO_LOAD_OPS(0, a0, b0)                                                               \
        __builtin_amdgcn_sched_barrier(0);                                                     \
        glds_u((C) + 1, sU + ((BUF) ^ 1) * U_FLOATS);                                          \
        glds_patch((C) + 2, sP + (BUF) * P_FLOATS);                                            \
        xform_rows(sP + ((BUF) ^ 1) * P_FLOATS + poff, pmask, t);                              \
        WINO_LOAD_OPS(1, a1, b1)                                                               \
        xform_cols_store(t, 0, vd0, vd1);                                                      \
        WINO_MFMAS(a0, b0)                                                                     \
        __builtin_amdgcn_sched_barrier(0);                                                     \
        WINO_LOAD_OPS(2, a0, b0)                                                               \
        xform_cols_store(t, 1, vd0, vd1);                                                      \
        WINO_MFMAS(a1, b1)                                                                     \
        __builtin_amdgcn_sched_barrier(0);                                                     \
        WINO_LOAD_OPS(3, a1, b1)                                                               \
        xform_cols_store(t, 2, vd0, vd1);                                                      \
        WINO_MFMAS(a0, b0)                                                                     \
        __builtin_amdgcn_sched_barrier(0);                                                     \
        xform_cols_store(t, 3, vd0, vd1);                                                      \
        WINO_MFMAS(a1, b1)                                                                     \
        __syncthreads();                                                                       \
    } while (0)
    static_assert(KC == 8, "the step below is written for four k-steps per chunk");
    for (int c = 0; c < nchunks; c += 2) {
        WINO_STEP(c, 0);
        WINO_STEP(c + 1, 1);
    }
#undef WINO_LOAD_OPS
#undef WINO_MFMAS
#undef WINO_STEP

    // ---- output transform + epilogue.  acc[p][r]: tile row (r&3) + 8*(r>>2) + 4*hk of m-tile mt, cout li of n-tile nt
    const int n = n0 + nt * 32 + li;
    const float sh = a.shift[n];
    float w2v = 0.f;
    if (MODE == 1) w2v = a.w2[n];
    float* red = smem;                                   // HEAD1: [2][NTILE][4] partial sums (after the last sync)
#pragma unroll
    for (int r = 0; r < 16; ++r) {
        const int row = (r & 3) + 8 * (r >> 2) + 4 * hk;
        const int tile = mt * 32 + row;
        float m[4][4];
#pragma unroll
        for (int p = 0; p < 16; ++p) m[p >> 2][p & 3] = acc[p][r];
        float t0[4], t1[4];
#pragma unroll
        for (int j = 0; j < 4; ++j) { t0[j] = m[0][j] + m[1][j] + m[2][j]; t1[j] = m[1][j] - m[2][j] - m[3][j]; }
        float y[2][2];
        y[0][0] = t0[0] + t0[1] + t0[2]; y[0][1] = t0[1] - t0[2] - t0[3];
        y[1][0] = t1[0] + t1[1] + t1[2]; y[1][1] = t1[1] - t1[2] - t1[3];
        const int uy = by * 2 * TB + 2 * (tile >> 3), ux = bx * 2 * TB + 2 * (tile & 7);
        if (MODE == 0) {
#pragma unroll
            for (int aa = 0; aa < 2; ++aa)
#pragma unroll
                for (int bb = 0; bb < 2; ++bb) {
                    const int yy = ry + a.dil * (uy + aa), xx = rx + a.dil * (ux + bb);
                    if (yy < a.H && xx < a.W) {
                        const int64_t o = (((int64_t)b * a.H + yy) * a.W + xx) * a.Cout + n;
                        float v = y[aa][bb] + sh;
                        if (a.residual) v += a.residual[o];
                        a.out[o] = act_apply(v, a.act);
                    }
                }
        } else {
            float v[4];
#pragma unroll
            for (int q = 0; q < 4; ++q) {
                float t = y[q >> 1][q & 1] + sh;
                v[q] = (t > 0.f ? t : 0.f) * w2v;
#pragma unroll
                for (int o = 16; o > 0; o >>= 1) v[q] += __shfl_xor(v[q], o, 32);
            }
            if (li == 0) {
#pragma unroll
                for (int q = 0; q < 4; ++q) red[(nt * NTILE + tile) * 4 + q] = v[q];
            }
        }
    }
    if (MODE == 1) {
        __syncthreads();
        const int tile = tid >> 2, q = tid & 3;
        const int uy = by * 2 * TB + 2 * (tile >> 3) + (q >> 1), ux = bx * 2 * TB + 2 * (tile & 7) + (q & 1);
        const int yy = ry + a.dil * uy, xx = rx + a.dil * ux;
        if (yy < a.H && xx < a.W) {
            const float z = red[tile * 4 + q] + red[(NTILE + tile) * 4 + q] + a.b2[0];
            a.out[((int64_t)b * a.H + yy) * a.W + xx] = 1.0f / (1.0f + expf(-z));
        }
    }
}

template <int MODE>
int launch_wino(const wino_args& a, int batch, hipStream_t s)
{
    auto kern = conv3x3_wino_kernel<MODE>;
    const size_t lds = (size_t)LDS_FLOATS * sizeof(float);
    hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
    if (e != hipSuccess) return (int)e;
    dim3 grid((unsigned)(a.nbx * a.dil), (unsigned)(a.nby * a.dil), (unsigned)(batch * a.ngroups));
    hipLaunchKernelGGL(kern, grid, dim3(WT), lds, s, a);
    AWSEG_LAUNCH_CHECK();
    return 0;
}

}  // namespace

AWSEG_API int awseg_conv3x3_winograd_nhwc(const float* x, int batch, int height, int width, int cin, int cout, int dilation,
                                          const float* u, const float* shift, const float* residual, int act,
                                          const float* w2, const float* b2, float* out, awseg_stream_t stream)
{
    if (batch == 0) return 0;
    if (!x || !u || !shift || !out || batch < 0 || height < 1 || width < 1 || dilation < 1) return AWSEG_EINVAL;
    if (cin < 2 * KC || (cin % (2 * KC)) || cout < NB || (cout % NB)) return AWSEG_ERANGE;
    if ((w2 == nullptr) != (b2 == nullptr)) return AWSEG_EINVAL;
    if (w2 && (cout != NB || residual)) return AWSEG_ERANGE;          // the fused 1x1 head reduces over one 64-channel block
    if (act != AWSEG_ACT_NONE && act != AWSEG_ACT_RELU) return AWSEG_ERANGE;
    if (((uintptr_t)x & 15) || ((uintptr_t)u & 15)) return AWSEG_EALIGN;
    if ((int64_t)height * width * cin >= (int64_t)1 << 30 || (int64_t)16 * cin * cout >= (int64_t)1 << 30) return AWSEG_ERANGE;   // 32-bit element offsets
    wino_args a;
    a.x = x; a.U = u; a.shift = shift; a.residual = residual; a.w2 = w2; a.b2 = b2; a.out = out;
    a.H = height; a.W = width; a.Cin = cin; a.Cout = cout; a.dil = dilation; a.act = act;
    const int hs = (height + dilation - 1) / dilation, ws = (width + dilation - 1) / dilation;
    a.nbx = (ws + 2 * TB - 1) / (2 * TB); a.nby = (hs + 2 * TB - 1) / (2 * TB); a.ngroups = cout / NB;
    if ((int64_t)batch * a.ngroups > 65535 || (int64_t)a.nby * dilation > 65535) return AWSEG_ERANGE;
    return w2 ? launch_wino<1>(a, batch, awseg_s(stream)) : launch_wino<0>(a, batch, awseg_s(stream));
}
