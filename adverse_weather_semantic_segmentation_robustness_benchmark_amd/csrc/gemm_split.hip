// gemm_split.hip — out[M,N] = act(x[M,K] . w[N,K]^T + bias[N] (+ residual[M,N])) in float32 grade on the f16 matrix
// cores: the compute-bound 1x1 convolutions of the ResNet-50 encoder (layer3 / layer4), the ASPP projections and the
// decoder (PKG/models/model.py:349) — the same operator as awseg_gemm_bias_act (gemm.hip, hipBLASLt on the
// float32-input MFMA), which stays the path for the HBM-bound shapes.
//
// v_mfma_f32_32x32x16_f16 issues 16x the multiply-adds per cycle of the float32-input MFMA, and a float32 product is
// three f16 products once both operands are SPLIT (attn.hip has the same scheme, measured there against float64):
//     x = xh + xl / 2048,  xh = f16(x),  xl = f16((x - xh) * 2048)            (22 significant bits)
//     x * w = xh*wh + (xh*wl + xl*wh) / 2048                                  (+ O(2^-22 |x w|), dropped)
// Every f16 product is exact in the float32 accumulator and the accumulation itself is float32, so the result differs
// from a float32 GEMM by operand rounding at 2^-22 instead of 2^-24 — the same order as the difference between two
// float32 summation orders (tests/test_gpu_kernels.py prices both against float64).
//   * weights are split once (awseg_gemm_split_weights -> [2][N][K] f16), activations at tile-load time;
//   * block = 128 x 128 outputs, 8 waves as 4 x 2, each 32 x 64 = 1 x 2 MFMA tiles with a main and a correction
//     accumulator (64 accumulator registers, 128 VGPRs: four waves per SIMD); K tiles of 32, double-buffered in LDS,
//     next tile's global loads in flight during the MFMAs; LDS rows = 32 hi | 32 lo | pad halfs (144 B: conflict-free
//     ds_read_b128).  Measured limit: the operand-fetch rate, not the MFMA, LDS or vector pipes;
//   * lanes own output COLUMNS (n), so a store instruction writes 128 contiguous bytes per output row;
//   * persistent blocks (two per CU) walk the tiles XCD-aware (the turns of one XCD sweep the n-tiles of one m-tile
//     band, whose x rows stay in that XCD's L2) and fetch the next tile's first K tile before their own epilogue, so
//     the epilogue's stores and the next prologue's load latency overlap.
// Operand range.  f16 holds |v| < 65504 and the scaled low part needs |v| < 2^15, so:
//   * weights are NORMALISED when they are split: awseg_gemm_split_weights finds max|w| on the device and stores
//     w * 2^-ew (max in [2^13, 2^14)) together with 2^ew, which the GEMM folds into its epilogue — any float32 weight
//     tensor keeps 22 significant bits, tiny or huge;
//   * activations are split OPTIMISTICALLY (scale 1) while every block tracks max|x| of the A tiles it stages; a
//     block that has seen |x| >= 2^15 throws its accumulators away and recomputes the tile with x * 2^-e (e from the
//     observed maximum: exact, power of two), multiplying 2^e back in the epilogue.  LayerNorm / BatchNorm-scaled
//     activations never take the second pass; a tensor at 1e5 or 1e30 costs twice the time and is still float32-grade
//     (tests/test_gpu_kernels.py::test_gemm_split_large_operands).  Inf / NaN propagate as in a float32 GEMM.
#include "awseg_common.h"

namespace {

typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef _Float16 h8 __attribute__((ext_vector_type(8)));
typedef _Float16 h2 __attribute__((ext_vector_type(2)));
typedef unsigned u32x2 __attribute__((ext_vector_type(2)));
typedef unsigned u32x4 __attribute__((ext_vector_type(4)));

constexpr int GT = 512;          // threads of a GEMM block (8 waves)
constexpr int SWT = 256;         // threads of the weight-split kernel
constexpr int GKT = 32;          // K tile
constexpr int GROW = 72;         // halfs per LDS row: 32 hi | 32 lo | 8 pad
constexpr float kLoScale = 2048.0f, kLoInv = 1.0f / 2048.0f;

__device__ __forceinline__ void split_pair(float a, float b, unsigned& hi, unsigned& lo)
{
    const auto hp = __builtin_amdgcn_cvt_pkrtz(a, b);
    const h2 hh = __builtin_bit_cast(h2, hp);
    const float ra = (a - (float)hh.x) * kLoScale, rb = (b - (float)hh.y) * kLoScale;
    hi = __builtin_bit_cast(unsigned, hp);
    lo = __builtin_bit_cast(unsigned, __builtin_amdgcn_cvt_pkrtz(ra, rb));
}

// Unscaled form: hi = f16(x) (round toward zero), lo = f16(x - hi) — three instructions per pair (v_cvt_pkrtz + two v_fma_mix).
// The low part keeps its 11 bits while |lo| >= 2^-14, i.e. |x| >= 2^-3; the single-accumulator kernel stages x * 2^4, which
// moves that bound to 2^-7 (below it the absolute error is 2^-29 of a unit: far inside float32 grade for O(1) tensors).
__device__ __forceinline__ void split_pair_unscaled(float a, float b, unsigned& hi, unsigned& lo)
{
    hi = __builtin_bit_cast(unsigned, __builtin_amdgcn_cvt_pkrtz(a, b));
    asm("v_fma_mixlo_f16 %0, %1, 1.0, -%3 op_sel_hi:[0,0,1]\n\t"
        "v_fma_mixhi_f16 %0, %2, 1.0, -%3 op_sel:[0,0,1] op_sel_hi:[0,0,1]"
        : "=&v"(lo) : "v"(a), "v"(b), "v"(hi));
}

// power-of-two scale that brings a maximum magnitude with float bits `maxbits` into [2^13, 2^14): returns e with
// scaled = v * 2^-e.  Zero / subnormal maxima (and Inf / NaN) leave the tensor unscaled.
__device__ __forceinline__ int norm_exponent(unsigned maxbits)
{
    const int ex = (int)(maxbits >> 23) & 0xff;
    if (ex == 0 || ex == 0xff) return 0;
    return ex - 127 - 13;
}
__device__ __forceinline__ float pow2f(int e) { return __builtin_bit_cast(float, (unsigned)(127 + e) << 23); }   // -126 <= e <= 127

// trailer of a split-weight buffer (8 uint16 = 16 bytes behind the [2][N][K] halfs): {max|w| bits, 2^ew as float, 0, 0}
__global__ __launch_bounds__(SWT)
void weights_absmax_kernel(const float* __restrict__ w, int64_t n_elems, unsigned* __restrict__ trailer)
{
    float m = 0.f;
    for (int64_t i = (int64_t)blockIdx.x * SWT + threadIdx.x; i < n_elems; i += (int64_t)gridDim.x * SWT) m = fmaxf(m, fabsf(w[i]));
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) m = fmaxf(m, __shfl_xor(m, o, 64));
    if ((threadIdx.x & 63) == 0 && m > 0.f) atomicMax(trailer, __builtin_bit_cast(unsigned, m));
}

__global__ __launch_bounds__(SWT)
void split_weights_kernel(const float* __restrict__ w, int64_t n_elems, uint16_t* __restrict__ out, unsigned* __restrict__ trailer)
{
    const int e = norm_exponent(trailer[0]);
    // |e| can exceed the exponent range of one float factor (max|w| = 1e-38 -> e = -139): apply it in two halves
    const int e1 = e / 2, e2 = e - e1;
    const float s1 = pow2f(-e1), s2 = pow2f(-e2);
    if (blockIdx.x == 0 && threadIdx.x == 0) {
        trailer[1] = (unsigned)e;                                   // the GEMM rebuilds 2^e from the integer (same two-factor form)
    }
    const int64_t i = ((int64_t)blockIdx.x * SWT + threadIdx.x) * 2;
    if (i >= n_elems) return;
    const float a = w[i] * s1 * s2, b = (i + 1 < n_elems) ? w[i + 1] * s1 * s2 : 0.f;
    unsigned hi, lo;
    split_pair(a, b, hi, lo);
    out[i] = (uint16_t)hi; out[n_elems + i] = (uint16_t)lo;
    if (i + 1 < n_elems) { out[i + 1] = (uint16_t)(hi >> 16); out[n_elems + i + 1] = (uint16_t)(lo >> 16); }
}

#ifdef AWSEG_GEMM_STAMP
// tools/probe_gemm_stamps.hip: cycle stamps of one wave's K loop (block 0, wave 0): [reads+MFMAs, wait for the next
// tile's global loads, split + LDS writes, load issue, barrier, iterations]
__device__ unsigned long long g_gemm_stamp[8];
#define STAMP(var) const unsigned long long var = __builtin_readcyclecounter()
#endif

struct gemm_args {
    const float* x; const _Float16* wh; const _Float16* wl; const float* bias; const float* residual; float* out;
    const unsigned* trailer;                                     // {max|w| bits, weight exponent ew} (awseg_gemm_split_weights)
    int64_t M; int N, K, act, ntm, ntm8, ntn;
    // conv = 1: x is an NHWC image batch [B, cH, cW, cC] and row m = (b, oy, ox) of the A operand is gathered from it —
    // column k = (ky * ckw + kx) * cC + c is x[b, oy * cs - cp + ky * cd, ox * cs - cp + kx * cd, c], zero outside (the im2col
    // matrix of awseg_im2col_nhwc, never materialised).  cC % 32 == 0: a 32-wide K tile lies inside one tap.
    int conv, cH, cW, cC, cHo, cWo, ckw, cs, cp, cd;
};

constexpr float kSplitLimit = 32768.0f;                          // |x| below this splits without loss (hi < 65504, lo * 2048 < 65504)

__device__ __forceinline__ constexpr int acc_row(int r, int hk) { return (r & 3) + 8 * (r >> 2) + 4 * hk; }

// MT x NT MFMA tiles per wave, WM x WN waves per block (8 waves): block tile (32 MT WM) x (32 NT WN).
//   <1, 2, 4, 2>: 128 x 128, 64 accumulator registers per lane, four waves per SIMD, two blocks per CU
//   <2, 2, 2, 4>: 128 x 256, 128 accumulator registers, two waves per SIMD, one block per CU: 25 % fewer operand bytes
//                 fetched per multiply-add (the measured limit) for the shapes with N >= 256
typedef __bf16 bf8 __attribute__((ext_vector_type(8)));

// float32 pair -> packed bf16 (round to nearest even: v_cvt_pk_bf16_f32)
__device__ __forceinline__ unsigned pack_bf16(float a, float b)
{
    typedef __bf16 bf2 __attribute__((ext_vector_type(2)));
    typedef float f2 __attribute__((ext_vector_type(2)));
    return __builtin_bit_cast(unsigned, __builtin_convertvector(f2{a, b}, bf2));
}

// BF16 = true: the same kernel with ONE v_mfma_f32_32x32x16_bf16 per product tile (BASELINE config 5: the bf16 MFMA
// path): activations are rounded to bf16 when they are staged, weights come as one bf16 plane (awseg_gemm_bf16_weights
// writes it where the split form keeps its high parts), float32 accumulation and epilogue.  bf16 has float32's exponent
// range: no range guard, no second pass.
// ONE = true: ONE accumulator per product tile.  The weights' low plane is brought back to its unscaled value when it is staged
// (x 2^-11, exact: v_pk_mul_f16), the activations are split into unscaled parts, and the three products of a tile add into the
// same registers — half the accumulator registers, which is what lets a wave own 128 x 64 (MT 4, NT 2: 12 KB of LDS fragment
// reads per 24 MFMAs instead of 8 KB per 12) in a 256 x 256 block tile.
template <int MT, int NT, int WM, int WN, bool KTAIL, bool BF16, bool ONE = false>
__global__ __launch_bounds__(GT, (MT * NT <= 2 ? 4 : 2))
void gemm_split_kernel(gemm_args a)
{
    static_assert(!(ONE && BF16), "single-accumulator form is for split operands");
    constexpr float kSx0 = ONE ? 16.0f : 1.0f;                   // optimistic-pass activation scale (ONE: see split_pair_unscaled)
    constexpr int kXe0 = ONE ? -4 : 0;
    static_assert(WM * WN == 8, "eight waves");
    constexpr int BM = 32 * MT * WM, BN = 32 * NT * WN;
    constexpr int NA = BM / 64;                                  // float4 of x per thread per K tile
    constexpr int NB = BN / 64;                                  // 16-byte chunks of w per thread per K tile (hi and lo)
    __shared__ __attribute__((aligned(16))) _Float16 sA[2][BM * GROW];
    __shared__ __attribute__((aligned(16))) _Float16 sB[2][BN * GROW];
    __shared__ unsigned sMax[2];                                 // max|x| bits seen by the block in a pass (if >= 2^15), by pass parity
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int hk = lane >> 5, li = lane & 31;
    const int wm = wave / WN, wn = wave % WN;                    // WM x WN waves, each (32 MT) rows x (32 NT) columns
    const int K = a.K;
    const int ntiles = a.ntm8 * a.ntn;                           // tile slots (m-tiles rounded up to 8 per group)

    // staging roles.  A: 128 rows x 8 float4 (k = 4c .. 4c+3); thread -> rows (tid>>3) + 64 i, column group c = tid & 7
    const int ar = tid >> 3, ac = tid & 7;
    // B: BN rows x (4 hi + 4 lo) 16-byte chunks; thread -> rows (tid >> 2) + 128 (i >> 1), chunk c8 = tid & 3, half i & 1
    const int bdst0 = (tid >> 2) * GROW + 8 * (tid & 3);
    // persistent walk over the tiles, XCD-aware: slot % 8 is the XCD (gridDim.x is a multiple of 8); consecutive turns
    // of one XCD sweep the n-tiles of one m-tile, so the x rows it re-reads are in that XCD's L2
    auto tile_of = [&](int slot, int64_t& m0, int& n0) -> bool {
        const int xcd = slot & 7, jj = slot >> 3;
        const int nt_i = jj % a.ntn, mt_i = (jj / a.ntn) * 8 + xcd;
        m0 = (int64_t)mt_i * BM; n0 = nt_i * BN;
        return mt_i < a.ntm;
    };
    // operand addresses of the tile being fetched: block-uniform bases (scalar registers) + 32-bit lane offsets
    const float* xb = nullptr; const _Float16* whb = nullptr; const _Float16* wlb = nullptr;
    int aoff[NA], boff[NB / 2];
    int cby[NA], cy0[NA], cx0[NA];                               // conv: image row base b * cH, first tap's input row / column of the A rows
    auto point = [&](int64_t m0, int n0) {
        xb = a.x + m0 * K; whb = a.wh + (int64_t)n0 * K; wlb = a.wl + (int64_t)n0 * K;
        const int64_t mleft = a.M - m0;                          // clamped rows are computed and never stored
        const int nleft = a.N - n0;
#pragma unroll
        for (int i = 0; i < NA; ++i) {
            const int row = ar + 64 * i;
            aoff[i] = (int)(row < mleft ? row : mleft - 1) * K + 4 * ac;
            if (a.conv) {                                        // block-uniform
                const int64_t m = m0 + (row < mleft ? row : mleft - 1);
                const int b = (int)(m / ((int64_t)a.cHo * a.cWo));
                const int rem = (int)(m - (int64_t)b * a.cHo * a.cWo);
                const int oy = rem / a.cWo, ox = rem - oy * a.cWo;
                cby[i] = b * a.cH; cy0[i] = oy * a.cs - a.cp; cx0[i] = ox * a.cs - a.cp;
            }
        }
#pragma unroll
        for (int i = 0; i < NB / 2; ++i) {
            const int r = (tid >> 2) + 128 * i;
            boff[i] = (r < nleft ? r : nleft - 1) * K + 8 * (tid & 3);
        }
    };
    float4 areg[NA]; u32x4 breg[NB];
    float amax = 0.f;                                            // max|x| this thread staged in the current pass
    bool scaled = false;                                         // second pass over a tile whose activations left the split range
    float sx = kSx0;                                             // activation scale of the pass (2^-e)
    int xe = kXe0;
    auto fetch = [&](int k0) {
        if (a.conv) {                                            // block-uniform: the K tile's tap and its channel offset
            const int tap = k0 / a.cC, c0 = k0 - tap * a.cC + 4 * ac;
            const int ky = tap / a.ckw, kx = tap - ky * a.ckw;
#pragma unroll
            for (int i = 0; i < NA; ++i) {
                const int iy = cy0[i] + ky * a.cd, ix = cx0[i] + kx * a.cd;
                const bool ok = (unsigned)iy < (unsigned)a.cH && (unsigned)ix < (unsigned)a.cW;
                areg[i] = ok ? *reinterpret_cast<const float4*>(a.x + ((int64_t)(cby[i] + iy) * a.cW + ix) * a.cC + c0)
                             : make_float4(0.f, 0.f, 0.f, 0.f);
            }
        } else {
#pragma unroll
        for (int i = 0; i < NA; ++i) {
            if (KTAIL && k0 + 4 * ac >= K) areg[i] = make_float4(0.f, 0.f, 0.f, 0.f);
            else areg[i] = *reinterpret_cast<const float4*>(xb + aoff[i] + k0);
        }
        }
#pragma unroll
        for (int i = 0; i < NB; ++i) {
            if (BF16 && (i & 1)) continue;                       // no low plane
            if (KTAIL && k0 + 8 * (tid & 3) >= K) breg[i] = u32x4{0u, 0u, 0u, 0u};
            else breg[i] = *reinterpret_cast<const u32x4*>(((i & 1) == 0 ? whb : wlb) + boff[i >> 1] + k0);
        }
    };
    auto stage = [&](int buf) {
#pragma unroll
        for (int i = 0; i < NA; ++i) {
            u32x2 H, L; unsigned hh, ll;
            float4 v = areg[i];
            _Float16* d = &sA[buf][(ar + 64 * i) * GROW + 4 * ac];
            if (BF16) {
                H[0] = pack_bf16(v.x, v.y); H[1] = pack_bf16(v.z, v.w);
                *reinterpret_cast<u32x2*>(d) = H;
                continue;
            }
            if (ONE || scaled) { v.x *= sx; v.y *= sx; v.z *= sx; v.w *= sx; }       // block-uniform branch
            amax = __builtin_fmaxf(__builtin_fmaxf(amax, __builtin_fabsf(v.x)), __builtin_fabsf(v.y));   // v_max3_f32 with |.| modifiers
            amax = __builtin_fmaxf(__builtin_fmaxf(amax, __builtin_fabsf(v.z)), __builtin_fabsf(v.w));
            if (ONE) {
                split_pair_unscaled(v.x, v.y, hh, ll); H[0] = hh; L[0] = ll;
                split_pair_unscaled(v.z, v.w, hh, ll); H[1] = hh; L[1] = ll;
            } else {
                split_pair(v.x, v.y, hh, ll); H[0] = hh; L[0] = ll;
                split_pair(v.z, v.w, hh, ll); H[1] = hh; L[1] = ll;
            }
            *reinterpret_cast<u32x2*>(d) = H;
            *reinterpret_cast<u32x2*>(d + 32) = L;
        }
#pragma unroll
        for (int i = 0; i < NB; ++i) {
            if (BF16 && (i & 1)) continue;
            if (ONE && (i & 1)) {                                // low plane: stored as (w - hi) * 2^11 -> back to w - hi
                // one vector multiply (element-wise writes through breg[i][q] in a loop were compiled into a chain reading the
                // element just written: hipcc / clang-22, checked in the ISA)
                const h8 lo8 = __builtin_bit_cast(h8, breg[i]) * (_Float16)0.00048828125f;
                breg[i] = __builtin_bit_cast(u32x4, lo8);
            }
            *reinterpret_cast<u32x4*>(&sB[buf][bdst0 + (i >> 1) * 128 * GROW + 32 * (i & 1)]) = breg[i];
        }
    };

    const int nkt = (K + GKT - 1) / GKT;
    const int fa = (wm * 32 * MT + li) * GROW + 8 * hk;          // this lane's fragment row in sA (m-tile 0)
    const int fb = (wn * 32 * NT + li) * GROW + 8 * hk;

    // first live tile of this block
    int slot = blockIdx.x;
    int64_t m0 = 0; int n0 = 0;
    while (slot < ntiles && !tile_of(slot, m0, n0)) slot += gridDim.x;
    if (slot >= ntiles) return;                                  // block-uniform, before any barrier
    point(m0, n0);
    fetch(0);
    if (tid < 2) sMax[tid] = 0u;                                 // ordered before its first use by the barrier behind stage(0)
    int par = 0;                                                 // pass parity: which sMax word this pass reports into
    const int we = (int)a.trailer[1];                            // weights were stored as w * 2^-we

    while (true) {
        // next live tile (its first K tile is fetched while this tile's last K tiles compute / its epilogue stores)
        int nslot = slot + gridDim.x;
        int64_t nm0 = 0; int nn0 = 0;
        while (nslot < ntiles && !tile_of(nslot, nm0, nn0)) nslot += gridDim.x;
        const bool has_next = nslot < ntiles;

        f32x16 am[MT][NT], ac2[MT][NT];
#pragma unroll
        for (int i = 0; i < MT; ++i)
#pragma unroll
            for (int j = 0; j < NT; ++j)
#pragma unroll
                for (int r = 0; r < 16; ++r) { am[i][j][r] = 0.f; ac2[i][j][r] = 0.f; }

        amax = 0.f;
        stage(0);
        if (nkt > 1) fetch(GKT);
        else if (has_next) { point(nm0, nn0); fetch(0); }
        __syncthreads();

#ifdef AWSEG_GEMM_STAMP
        unsigned long long st_mma = 0, st_wait = 0, st_stage = 0, st_fetch = 0, st_bar = 0, st_n = 0;
#endif
        for (int t = 0; t < nkt; ++t) {
            const int buf = t & 1;
            // K tile t+1 (in registers since the previous iteration) is split and staged FIRST, and the loads of tile t+2
            // are issued right behind it, so the vector-memory unit works through them during this tile's MFMAs.  Issued
            // after the MFMAs (the obvious order) the 8 waves of a block hit the unit together just before the barrier:
            // cycle stamps (tools/probe_gemm_stamps.hip) put 21 % of a K tile into issuing those loads and 32 % into the
            // barrier behind them.  (Ring slot buf^1 was last read before the previous barrier.)
#ifdef AWSEG_GEMM_STAMP
            STAMP(c0);
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            STAMP(c1);
            if (t + 1 < nkt) stage(buf ^ 1);
            asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
            STAMP(c2);
#else
            if (t + 1 < nkt) stage(buf ^ 1);
#endif
            // range guard bookkeeping (two block-uniform branches per K tile).  First K tile: clear the OTHER parity's word
            // — its readers (previous pass) are behind this pass's first barrier, its writers (next pass) behind this
            // pass's last one.  Last K tile: everything this thread will stage in this pass has been staged; report.
            if (t == 0 && tid == 0) sMax[par ^ 1] = 0u;
            if (!BF16 && t == nkt - 1 && !scaled && amax >= kSplitLimit) atomicMax(&sMax[par], __builtin_bit_cast(unsigned, amax));
            if (t + 1 < nkt) {
                if (t + 2 < nkt) fetch((t + 2) * GKT);
                else if (has_next) { point(nm0, nn0); fetch(0); }
            }
#ifdef AWSEG_GEMM_STAMP
            __builtin_amdgcn_sched_barrier(0);
            STAMP(c3);
#endif
#pragma unroll
            for (int ks = 0; ks < 2; ++ks) {
                h8 Ah[MT], Al[MT], Bh[NT], Bl[NT];
#pragma unroll
                for (int i = 0; i < MT; ++i) {
                    const _Float16* pa = &sA[buf][fa + i * 32 * GROW + 16 * ks];
                    Ah[i] = *reinterpret_cast<const h8*>(pa);
                    if (!BF16) Al[i] = *reinterpret_cast<const h8*>(pa + 32);
                }
#pragma unroll
                for (int j = 0; j < NT; ++j) {
                    const _Float16* pb = &sB[buf][fb + j * 32 * GROW + 16 * ks];
                    Bh[j] = *reinterpret_cast<const h8*>(pb);
                    if (!BF16) Bl[j] = *reinterpret_cast<const h8*>(pb + 32);
                }
#pragma unroll
                for (int i = 0; i < MT; ++i)
#pragma unroll
                    for (int j = 0; j < NT; ++j) {
                        if (BF16) {
                            am[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf8, Ah[i]), __builtin_bit_cast(bf8, Bh[j]), am[i][j], 0, 0, 0);
                            continue;
                        }
                        am[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_f16(Ah[i], Bh[j], am[i][j], 0, 0, 0);
                        if (ONE) {
                            am[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_f16(Ah[i], Bl[j], am[i][j], 0, 0, 0);
                            am[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_f16(Al[i], Bh[j], am[i][j], 0, 0, 0);
                            continue;
                        }
                        ac2[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_f16(Ah[i], Bl[j], ac2[i][j], 0, 0, 0);
                        ac2[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_f16(Al[i], Bh[j], ac2[i][j], 0, 0, 0);
                    }
            }
#ifdef AWSEG_GEMM_STAMP
            __builtin_amdgcn_sched_barrier(0);
            STAMP(c4);
            __syncthreads();
            STAMP(c5);
            st_wait += c1 - c0; st_stage += c2 - c1; st_fetch += c3 - c2; st_mma += c4 - c3; st_bar += c5 - c4; st_n += 1;
        }
        if (blockIdx.x == 0 && tid == 0) {
            g_gemm_stamp[0] += st_mma; g_gemm_stamp[1] += st_wait; g_gemm_stamp[2] += st_stage; g_gemm_stamp[3] += st_fetch;
            g_gemm_stamp[4] += st_bar; g_gemm_stamp[5] += st_n;
        }
#else
            __syncthreads();
        }
#endif

        // ---- range guard: some activation of this tile was too large for the optimistic split -> second pass, scaled
        {
            const unsigned mx = BF16 ? 0u : sMax[par];           // written before the K loop's last barrier; block-uniform
            par ^= 1;
            if (mx != 0u && !scaled) {
                const int ex = (int)(mx >> 23) & 0xff;
                if (ex != 0xff) {                                // Inf / NaN: nothing to rescue, let them propagate
                    xe = ex - 127 - 13 + kXe0;                   // max|x| * 2^-xe in [2^13, 2^14) (the maximum was taken after the x 2^-kXe0 staging scale)
                    sx = pow2f(-xe);                             // xe in [2, 114]: one factor is enough
                    scaled = true;
                    point(m0, n0);                               // the registers hold the NEXT tile's first K tile: fetch this one again
                    fetch(0);
                    continue;
                }
            }
        }
        // result scale: 2^(we + xe), applied as two factors (each a normal float; their product may legitimately overflow
        // to Inf exactly where the float32 GEMM would)
        const int oe = we + xe;                                  // xe = kXe0 on the optimistic pass
        const int oe1 = oe / 2, oe2 = oe - oe1;
        const float os1 = pow2f(oe1 < -126 ? -126 : (oe1 > 127 ? 127 : oe1)), os2 = pow2f(oe2 < -126 ? -126 : (oe2 > 127 ? 127 : oe2));
        const bool rescale = oe != 0;

        // ---- epilogue: lane = output column n, registers = rows m.  Raw-buffer accesses relative to the tile: one lane
        // offset per column group (out-of-range columns get an out-of-range VECTOR offset and are dropped) plus a scalar
        // row offset.  The hardware range check covers the vector offset only, so the one tile band with rows past M
        // takes the guarded form (row folded into the vector offset).  The residual loads of 8 rows are issued before
        // their stores (residual may alias out: element-wise the same thread reads, then writes).
        {
            const int64_t tile_off = m0 * a.N + n0;
            const int64_t rem = ((int64_t)a.M * a.N - tile_off) * 4;
            const int nrec = rem > 0x7fffffff ? 0x7fffffff : (int)rem;
            const __amdgpu_buffer_rsrc_t o_rsrc = __builtin_amdgcn_make_buffer_rsrc((void*)(a.out + tile_off), 0, nrec, 0x00020000);
            const __amdgpu_buffer_rsrc_t r_rsrc = __builtin_amdgcn_make_buffer_rsrc((void*)((a.residual ? a.residual : a.out) + tile_off), 0, nrec, 0x00020000);
            const bool has_res = a.residual != nullptr;
            const bool ragged = m0 + BM > a.M;                   // block-uniform
            const int mleft = ragged ? (int)(a.M - m0) : BM;
            const int vrow = (wm * 32 * MT + 4 * hk) * a.N;
#pragma unroll
            for (int j = 0; j < NT; ++j) {
                const int nl = wn * 32 * NT + j * 32 + li;
                const bool n_ok = n0 + nl < a.N;
                const float bv = (a.bias && n_ok) ? a.bias[n0 + nl] : 0.f;
                const int voff = n_ok ? (vrow + nl) * 4 : (int)0x80000000;
#pragma unroll
                for (int i = 0; i < MT; ++i) {
#pragma unroll
                    for (int half = 0; half < 2; ++half) {
                        float rv[8]; int vo[8], so[8];
#pragma unroll
                        for (int r8 = 0; r8 < 8; ++r8) {
                            const int r = 8 * half + r8;
                            const int rowc = i * 32 + (r & 3) + 8 * (r >> 2);                 // + wm * 32 MT + 4 * hk (in vrow)
                            if (ragged) {
                                vo[r8] = (wm * 32 * MT + 4 * hk + rowc < mleft) ? voff + rowc * a.N * 4 : (int)0x80000000;
                                so[r8] = 0;
                            } else { vo[r8] = voff; so[r8] = rowc * a.N * 4; }
                            rv[r8] = has_res ? __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(r_rsrc, vo[r8], so[r8], 0)) : 0.f;
                        }
#pragma unroll
                        for (int r8 = 0; r8 < 8; ++r8) {
                            const int r = 8 * half + r8;
                            float vv = ONE ? am[i][j][r] : fmaf(ac2[i][j][r], kLoInv, am[i][j][r]);
                            if (rescale) vv = vv * os1 * os2;                                  // block-uniform branch
                            vv = vv + bv + rv[r8];
                            if (a.act == 1) vv = fmaxf(vv, 0.f);
                            __builtin_amdgcn_raw_buffer_store_b32(__builtin_bit_cast(uint32_t, vv), o_rsrc, vo[r8], so[r8], 0);
                        }
                    }
                }
            }
        }
        if (!has_next) break;
        slot = nslot; m0 = nm0; n0 = nn0;
        scaled = false; sx = kSx0; xe = kXe0;
    }
}

}  // namespace

AWSEG_API int awseg_gemm_split_weights(const float* w, int n, int k, uint16_t* w_split, awseg_stream_t stream)
{
    if (n == 0 || k == 0) return 0;
    if (!w || !w_split || n < 0 || k < 0) return AWSEG_EINVAL;
    const int64_t ne = (int64_t)n * k;
    const int64_t blocks = (ne / 2 + SWT) / SWT;
    unsigned* trailer = reinterpret_cast<unsigned*>(w_split + 2 * ne);      // 16 bytes behind the two f16 planes
    if ((uintptr_t)trailer & 3) return AWSEG_EALIGN;
    hipError_t e = hipMemsetAsync(trailer, 0, 16, awseg_s(stream));
    if (e != hipSuccess) return (int)e;
    const int64_t rb = (ne + SWT - 1) / SWT;
    hipLaunchKernelGGL(weights_absmax_kernel, dim3((unsigned)(rb < 1024 ? rb : 1024)), dim3(SWT), 0, awseg_s(stream), w, ne, trailer);
    AWSEG_LAUNCH_CHECK();
    hipLaunchKernelGGL(split_weights_kernel, dim3((unsigned)blocks), dim3(SWT), 0, awseg_s(stream), w, ne, w_split, trailer);
    AWSEG_LAUNCH_CHECK();
    // N % 256 == 0, K % 8 == 0: the k-blocked image of gemm_split3.hip behind the trailer (awseg_gemm_split_weight_halfs says how much room)
    if (awseg_gemm_split3_bn(n) && k % 8 == 0) return awseg_gemm_split3_weights(w, n, k, w_split + 2 * ne + 8, trailer, awseg_s(stream));
    return 0;
}

AWSEG_API int64_t awseg_gemm_split_weight_halfs(int n, int k)
{
    if (n < 1 || k < 1) return -1;
    const int64_t ne = (int64_t)n * k;
    return 2 * ne + 8 + ((awseg_gemm_split3_bn(n) && k % 8 == 0) ? 2 * awseg_gemm_split3_image_rows(n) * ((k + 31) / 32 * 32) : 0);
}

namespace {
struct conv_desc { int H, W, C, Ho, Wo, kw, stride, pad, dil; int64_t batch; int pitch = 0, padx = -1; };   // pitch / padx: the ROWS form of gemm_split3.hip

int gemm_launch(bool bf16, const float* x, const uint16_t* w_split, const float* bias, const float* residual,
                int act, float* out, int64_t m, int n, int k, awseg_stream_t stream, const conv_desc* cv = nullptr)
{
    if (m == 0 || n == 0) return 0;
    if (!x || !w_split || !out || m < 0 || n < 0 || k < 8 || act < 0 || act > 1) return AWSEG_EINVAL;
    if (k % 8) return AWSEG_ERANGE;                              // 16-byte operand chunks
    if (((uintptr_t)x & 15) || ((uintptr_t)w_split & 15)) return AWSEG_EALIGN;
    gemm_args a;
    a.x = x; a.wh = reinterpret_cast<const _Float16*>(w_split); a.wl = a.wh + (int64_t)n * k;
    a.trailer = reinterpret_cast<const unsigned*>(w_split + 2 * (int64_t)n * k);
    a.bias = bias; a.residual = residual; a.out = out; a.M = m; a.N = n; a.K = k; a.act = act;
    a.conv = cv ? 1 : 0;
    if (cv) { a.cH = cv->H; a.cW = cv->W; a.cC = cv->C; a.cHo = cv->Ho; a.cWo = cv->Wo; a.ckw = cv->kw; a.cs = cv->stride; a.cp = cv->pad; a.cd = cv->dil; }
    else { a.cH = a.cW = a.cC = a.cHo = a.cWo = a.ckw = a.cs = 1; a.cp = 0; a.cd = 1; }
    static int cus = 0;                                          // CU count of the (single, per-process) device, read once
    if (cus == 0) {
        int dev = 0, n_cu = 0;
        if (hipGetDevice(&dev) != hipSuccess || hipDeviceGetAttribute(&n_cu, hipDeviceAttributeMultiprocessorCount, dev) != hipSuccess || n_cu < 1) n_cu = 256;
        cus = n_cu;
    }
    static int v3_on = -1;                                       // AWSEG_GEMM_SPLIT_V3=0: keep the register-staged kernels of round 2 (A/B measurements)
    if (v3_on < 0) { const char* e = getenv("AWSEG_GEMM_SPLIT_V3"); v3_on = e ? atoi(e) : 1; }
    // (bf16: only the 256-wide tile — on B5 + R101's N = 64 / 128 / 320 projections the narrow tiles measured slower than the
    // round-2 bf16 kernel: 105.0 against 98.7 ms per step)
    if (v3_on && awseg_gemm_split3_eligible(m, n, k, cv ? nullptr : x, out, residual, bias) && !(bf16 && (cv || n % 256)) &&
        (!cv || (cv->C % 32 == 0 && (int64_t)cv->batch * cv->H * cv->W * (cv->pitch > 0 ? cv->pitch : cv->C) * 4 <= 0x7fffffff)) &&
        (((m + 255) / 256) * (int64_t)((n + awseg_gemm_split3_bn(n) - 1) / awseg_gemm_split3_bn(n)) >= (int64_t)cus / 2 ||
         (!bf16 && ((m + 127) / 128) * (int64_t)((n + 127) / 128) >= (int64_t)cus / 2))) {     // few rows: 128-row tiles (gemm_split3.hip picks them)
        const int cdesc[12] = { cv ? cv->H : 0, cv ? cv->W : 0, cv ? cv->C : 0, cv ? cv->Ho : 0, cv ? cv->Wo : 0, cv ? cv->kw : 0,
                                cv ? cv->stride : 0, cv ? cv->pad : 0, cv ? cv->dil : 0, cv ? (int)cv->batch : 0, cv ? cv->pitch : 0, cv ? cv->padx : -1 };
        return awseg_gemm_split3_launch(x, w_split + 2 * (int64_t)n * k + 8, a.trailer, bias, residual, act, out, m, n, k, cus, awseg_s(stream),
                                        cv ? cdesc : nullptr, bf16);
    }
    if (cv && cv->pitch > 0) return AWSEG_ERANGE;                // the ROWS form exists in the LDS-DMA kernel only
    static int tile_mode = -1;                                   // AWSEG_GEMM_SPLIT_TILE = 128 / 256 / 512 forces the block tile (measurements; 512 = 256 x 256)
    if (tile_mode < 0) { const char* e = getenv("AWSEG_GEMM_SPLIT_TILE"); tile_mode = !e ? 0 : (atoi(e) == 512 ? 3 : (atoi(e) == 256 ? 2 : (atoi(e) == 128 ? 1 : 0))); }
    // 256 x 256 (single accumulator) when N fills it and there is at least one tile per CU (AWSEG_GEMM_SPLIT_HUGE_MIN_TILES per CU); else the 128 x 256 tile when N
    // fills it and there are enough tiles for every CU; else 128 x 128
    static int huge_min = -1;
    if (huge_min < 0) { const char* e = getenv("AWSEG_GEMM_SPLIT_HUGE_MIN_TILES"); huge_min = e ? atoi(e) : 1; }
    const bool huge = (tile_mode == 3 || (tile_mode == 0 && n % 256 == 0 && k >= 128 && ((m + 255) / 256) * (int64_t)(n / 256) >= (int64_t)huge_min * cus));   // K = 64: two K tiles per 256 x 256 epilogue, measured 6 % slower
    const bool wide = !huge && (tile_mode == 2 || (tile_mode == 0 && n % 256 == 0 && ((m + 127) / 128) * (int64_t)(n / 256) >= cus));
    // 256 x 128 (single accumulator, 64 x 64 wave tiles: 8 KB of LDS fragment reads per 12 MFMAs where the 128 x 128 tile's 32 x 64
    // wave tiles read 6 KB per 6) for the N % 128 == 0 shapes that cannot fill 256-wide tiles; AWSEG_GEMM_SPLIT_TALL=0 turns it off
    static int tall_on = -1;
    if (tall_on < 0) { const char* e = getenv("AWSEG_GEMM_SPLIT_TALL"); tall_on = e ? atoi(e) : 1; }
    const bool tall = !huge && !wide && !bf16 && tile_mode == 0 && tall_on && n % 128 == 0 && k >= 128 &&
                      ((m + 255) / 256) * (int64_t)(n / 128) >= cus;
    const int bn = (wide || huge) ? 256 : 128;
    const int bm = (huge || tall) ? 256 : 128;
    const int64_t ntm = (m + bm - 1) / bm;
    a.ntn = (n + bn - 1) / bn;
    const int64_t ntm8 = (ntm + 7) / 8 * 8;                      // 8 m-tiles (one per XCD) x all n-tiles per group
    if (ntm8 * a.ntn > 0x7fffffff || (int64_t)bm * n > 0x7fffffff) return AWSEG_ERANGE;
    a.ntm = (int)ntm; a.ntm8 = (int)ntm8;
    const int64_t slots = ntm8 * a.ntn;
    int64_t blocks = (int64_t)cus * ((wide || huge || tall) ? 1 : 2) / 8 * 8;      // persistent: one (128 x 256 / 256 x 256 / 256 x 128) or two (128 x 128) blocks per CU
    if (blocks < 8) blocks = 8;
    if (blocks > slots) blocks = slots;                          // slots is a multiple of 8
    const dim3 grid((unsigned)blocks), block(GT);
#define GEMM_GO(MT_, NT_, WM_, WN_)                                                                                    \
    do {                                                                                                              \
        if (bf16) {                                                                                                   \
            if (k % GKT) hipLaunchKernelGGL((gemm_split_kernel<MT_, NT_, WM_, WN_, true, true>), grid, block, 0, awseg_s(stream), a);   \
            else hipLaunchKernelGGL((gemm_split_kernel<MT_, NT_, WM_, WN_, false, true>), grid, block, 0, awseg_s(stream), a);         \
        } else {                                                                                                      \
            if (k % GKT) hipLaunchKernelGGL((gemm_split_kernel<MT_, NT_, WM_, WN_, true, false>), grid, block, 0, awseg_s(stream), a);  \
            else hipLaunchKernelGGL((gemm_split_kernel<MT_, NT_, WM_, WN_, false, false>), grid, block, 0, awseg_s(stream), a);        \
        }                                                                                                             \
    } while (0)
    if (huge && bf16) {                                          // one product per tile anyway: the same 128 x 64 wave tiles
        if (k % GKT) hipLaunchKernelGGL((gemm_split_kernel<4, 2, 2, 4, true, true, false>), grid, block, 0, awseg_s(stream), a);
        else hipLaunchKernelGGL((gemm_split_kernel<4, 2, 2, 4, false, true, false>), grid, block, 0, awseg_s(stream), a);
    }
    else if (huge) {
        if (k % GKT) hipLaunchKernelGGL((gemm_split_kernel<4, 2, 2, 4, true, false, true>), grid, block, 0, awseg_s(stream), a);
        else hipLaunchKernelGGL((gemm_split_kernel<4, 2, 2, 4, false, false, true>), grid, block, 0, awseg_s(stream), a);
    }
    else if (tall) {
        if (k % GKT) hipLaunchKernelGGL((gemm_split_kernel<2, 2, 4, 2, true, false, true>), grid, block, 0, awseg_s(stream), a);
        else hipLaunchKernelGGL((gemm_split_kernel<2, 2, 4, 2, false, false, true>), grid, block, 0, awseg_s(stream), a);
    }
    else if (wide) GEMM_GO(2, 2, 2, 4); else GEMM_GO(1, 2, 4, 2);
#undef GEMM_GO
    AWSEG_LAUNCH_CHECK();
    return 0;
}

__global__ __launch_bounds__(SWT)
void bf16_weights_kernel(const float* __restrict__ w, int64_t n_elems, uint16_t* __restrict__ out, unsigned* __restrict__ trailer)
{
    if (blockIdx.x == 0 && threadIdx.x == 0) { trailer[0] = 0u; trailer[1] = 0u; trailer[2] = 0u; trailer[3] = 0u; }   // exponent 0
    const int64_t i = ((int64_t)blockIdx.x * SWT + threadIdx.x) * 2;
    if (i >= n_elems) return;
    const unsigned p = pack_bf16(w[i], (i + 1 < n_elems) ? w[i + 1] : 0.f);
    out[i] = (uint16_t)p;
    if (i + 1 < n_elems) out[i + 1] = (uint16_t)(p >> 16);
}
}  // namespace

AWSEG_API int awseg_gemm_split_bias_act(const float* x, const uint16_t* w_split, const float* bias, const float* residual,
                                        int act, float* out, int64_t m, int n, int k, awseg_stream_t stream)
{
    return gemm_launch(false, x, w_split, bias, residual, act, out, m, n, k, stream);
}

AWSEG_API int awseg_gemm_split_dual_bias_act(const float* x, int k1, const float* x2, int k2, int64_t batch, int x2_height, int x2_width,
                                             int x2_stride, const uint16_t* w_split, const float* bias, const float* residual, int act,
                                             float* out, int64_t m, int n, awseg_stream_t stream)
{
    if (m == 0 || n == 0) return 0;
    if (!x || !x2 || !w_split || !out || m < 0 || n < 0 || k1 < 32 || k2 < 32 || act < 0 || act > 1 || x2_stride < 0) return AWSEG_EINVAL;
    if (k1 % 32 || k2 % 32) return AWSEG_ERANGE;                  // whole 32-deep K tiles from either source
    if (((uintptr_t)x & 15) || ((uintptr_t)x2 & 15) || ((uintptr_t)w_split & 15)) return AWSEG_EALIGN;
    const int k = k1 + k2;
    if (!awseg_gemm_split3_eligible(m, n, k, x, out, residual, bias)) return AWSEG_ERANGE;
    awseg_g3_dual d;
    d.x2 = x2; d.k1 = k1; d.stride = x2_stride; d.h = d.w = d.ho = d.wo = 1;
    if (x2_stride > 0) {
        if (batch < 1 || x2_height < 1 || x2_width < 1) return AWSEG_EINVAL;
        d.h = x2_height; d.w = x2_width; d.ho = (x2_height - 1) / x2_stride + 1; d.wo = (x2_width - 1) / x2_stride + 1;
        if (batch * d.ho * d.wo != m) return AWSEG_EINVAL;
        d.bytes = batch * (int64_t)x2_height * x2_width * k2 * 4;
    } else {
        d.bytes = m * (int64_t)k2 * 4;
    }
    if (d.bytes > 0x7fffffff) return AWSEG_ERANGE;                // 32-bit offsets into the second source
    static int cus = 0;
    if (cus == 0) {
        int dev = 0, n_cu = 0;
        if (hipGetDevice(&dev) != hipSuccess || hipDeviceGetAttribute(&n_cu, hipDeviceAttributeMultiprocessorCount, dev) != hipSuccess || n_cu < 1) n_cu = 256;
        cus = n_cu;
    }
    return awseg_gemm_split3_launch(x, w_split + 2 * (int64_t)n * k + 8, reinterpret_cast<const unsigned*>(w_split + 2 * (int64_t)n * k), bias, residual,
                                    act, out, m, n, k, cus, awseg_s(stream), nullptr, false, &d);
}

AWSEG_API int awseg_gemm_split_pieces_bias_act(const float* const* pieces, int n_pieces, int k_piece, const uint16_t* w_split, const float* bias,
                                               const float* residual, int act, float* out, int64_t m, int n, awseg_stream_t stream)
{
    if (m == 0 || n == 0) return 0;
    if (!pieces || n_pieces < 2 || n_pieces > 4 || !w_split || !out || m < 0 || n < 0 || k_piece < 32 || act < 0 || act > 1) return AWSEG_EINVAL;
    if (k_piece % 32) return AWSEG_ERANGE;
    for (int i = 0; i < n_pieces; ++i) {
        if (!pieces[i]) return AWSEG_EINVAL;
        if ((uintptr_t)pieces[i] & 15) return AWSEG_EALIGN;
    }
    if ((uintptr_t)w_split & 15) return AWSEG_EALIGN;
    const int k = n_pieces * k_piece;
    if (!awseg_gemm_split3_eligible(m, n, k, pieces[0], out, residual, bias)) return AWSEG_ERANGE;
    awseg_g3_dual d;
    d.x2 = pieces[1]; d.k1 = k_piece; d.stride = 0; d.h = d.w = d.ho = d.wo = 1;
    d.bytes = m * (int64_t)k_piece * 4;
    d.x3 = n_pieces > 2 ? pieces[2] : nullptr; d.x4 = n_pieces > 3 ? pieces[3] : nullptr;
    if (d.bytes > 0x7fffffff) return AWSEG_ERANGE;
    static int cus = 0;
    if (cus == 0) {
        int dev = 0, n_cu = 0;
        if (hipGetDevice(&dev) != hipSuccess || hipDeviceGetAttribute(&n_cu, hipDeviceAttributeMultiprocessorCount, dev) != hipSuccess || n_cu < 1) n_cu = 256;
        cus = n_cu;
    }
    return awseg_gemm_split3_launch(pieces[0], w_split + 2 * (int64_t)n * k + 8, reinterpret_cast<const unsigned*>(w_split + 2 * (int64_t)n * k), bias, residual,
                                    act, out, m, n, k, cus, awseg_s(stream), nullptr, false, &d);
}

AWSEG_API int awseg_conv_gemm_split_bias_act(const float* x, int64_t batch, int height, int width, int channels, int kernel_h,
                                             int kernel_w, int stride, int pad, int dilation, const uint16_t* w_split,
                                             const float* bias, const float* residual, int act, float* out, int n,
                                             awseg_stream_t stream)
{
    if (batch == 0) return 0;
    if (batch < 0 || height < 1 || width < 1 || channels < 32 || kernel_h < 1 || kernel_w < 1 || stride < 1 || pad < 0 || dilation < 1) return AWSEG_EINVAL;
    if (channels % 32) return AWSEG_ERANGE;                      // a 32-wide K tile must lie inside one tap
    const int ho = (height + 2 * pad - dilation * (kernel_h - 1) - 1) / stride + 1;
    const int wo = (width + 2 * pad - dilation * (kernel_w - 1) - 1) / stride + 1;
    if (ho < 1 || wo < 1) return AWSEG_ERANGE;
    const int64_t k = (int64_t)kernel_h * kernel_w * channels;
    if (k > 0x7fffffff || batch * height > 0x7fffffff) return AWSEG_ERANGE;
    const conv_desc cv = { height, width, channels, ho, wo, kernel_w, stride, pad, dilation, batch };
    return gemm_launch(false, x, w_split, bias, residual, act, out, batch * ho * wo, n, (int)k, stream, &cv);
}

AWSEG_API int awseg_conv_rows_gemm_split_bias_act(const float* x, int64_t batch, int height, int width_padded, int pixel_floats,
                                                  int kernel_h, int stride, int pad_y, int out_width, const uint16_t* w_split,
                                                  const float* bias, const float* residual, int act, float* out, int n,
                                                  awseg_stream_t stream)
{
    if (batch == 0) return 0;
    if (batch < 0 || height < 1 || width_padded < 8 || kernel_h < 1 || stride < 1 || pad_y < 0 || out_width < 1) return AWSEG_EINVAL;
    if (pixel_floats < 4 || pixel_floats % 4 || 32 % pixel_floats) return AWSEG_ERANGE;     // 16-byte chunks of whole pixels, a run of 32 floats
    // the last output column's run of 32 floats must end inside its (padded) image row
    if ((int64_t)(out_width - 1) * stride * pixel_floats + 32 > (int64_t)width_padded * pixel_floats) return AWSEG_ERANGE;
    const int ho = (height + 2 * pad_y - kernel_h) / stride + 1;
    if (ho < 1 || batch * height > 0x7fffffff) return AWSEG_ERANGE;
    conv_desc cv = { height, width_padded, 32, ho, out_width, 1, stride, pad_y, 1, batch };
    cv.pitch = pixel_floats; cv.padx = 0;
    return gemm_launch(false, x, w_split, bias, residual, act, out, batch * ho * out_width, n, kernel_h * 32, stream, &cv);
}

AWSEG_API int awseg_gemm_bf16_weights(const float* w, int n, int k, uint16_t* w_bf16, awseg_stream_t stream)
{
    if (n == 0 || k == 0) return 0;
    if (!w || !w_bf16 || n < 0 || k < 0) return AWSEG_EINVAL;
    const int64_t ne = (int64_t)n * k;
    unsigned* trailer = reinterpret_cast<unsigned*>(w_bf16 + 2 * ne);
    if ((uintptr_t)trailer & 3) return AWSEG_EALIGN;
    hipLaunchKernelGGL(bf16_weights_kernel, dim3((unsigned)((ne / 2 + SWT) / SWT)), dim3(SWT), 0, awseg_s(stream), w, ne, w_bf16, trailer);
    AWSEG_LAUNCH_CHECK();
    // N % 256 == 0, K % 8 == 0: the k-blocked bf16 image of gemm_split3.hip behind the trailer (awseg_gemm_bf16_weight_halfs)
    if (n % 64 == 0 && k % 8 == 0) return awseg_gemm_bf16_3_weights(w, n, k, w_bf16 + 2 * ne + 8, awseg_s(stream));
    return 0;
}

AWSEG_API int64_t awseg_gemm_bf16_weight_halfs(int n, int k)
{
    if (n < 1 || k < 1) return -1;
    const int64_t ne = (int64_t)n * k;
    return 2 * ne + 8 + ((n % 64 == 0 && k % 8 == 0) ? (int64_t)n * ((k + 31) / 32 * 32) : 0);
}

AWSEG_API int awseg_gemm_bf16_bias_act(const float* x, const uint16_t* w_bf16, const float* bias, const float* residual,
                                       int act, float* out, int64_t m, int n, int k, awseg_stream_t stream)
{
    return gemm_launch(true, x, w_bf16, bias, residual, act, out, m, n, k, stream);
}
