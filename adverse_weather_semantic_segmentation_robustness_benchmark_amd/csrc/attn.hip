// attn.hip — the self-attention of the MiT encoder (transformers SegformerEfficientSelfAttention inside
// PKG/models/model.py:193's SegformerModel call) for head_dim 32 in exact fp32 on the matrix cores:
//     O = softmax(Q K^T * scale) V          Q [B,Nq,heads*32], K/V [B,Nkv,heads*32] (token-major, as the
//                                           q/k/v Linear layers write them), O in the same layout.
// MiT-B0 has head_dim 32 in every stage and, at 1024x2048, 2048 keys per image after the strided reduction;
// the queries are what is large (131072 tokens in stage 1).
//
// Flash-style, one pass over the keys, both GEMMs on v_mfma_f32_32x32x2_f32, written around one fact measured on
// the Winograd kernel: fp32-input MFMA runs at the vector rate and VALU work does not hide behind
// it, so the softmax must cost as few vector instructions as possible:
//   * S^T = K Q^T is computed TRANSPOSED (keys on the accumulator rows, queries on its columns): a lane owns one
//     query, its 16 accumulator registers are 16 keys -> row max / row sum are in-register, plus ONE cross-half
//     shuffle each, instead of a 32-lane butterfly per query;
//   * the exponentiated tile P^T is, register for register, the B operand of the second GEMM O^T += V^T P^T
//     (lane = query column, register r of the two wave halves = keys row(r,0), row(r,1)); V^T is staged in LDS in
//     exactly that key order, so nothing is transposed or round-tripped;
//   * exp is one v_exp_f32 per element (log2 e folded into the query scale), the running maximum starts at a
//     large negative finite value (no inf - inf).
// (Tried: the softmax subtract / sum as inline-asm v_pk_* — wrong results, because hipcc pads the "VALU reads a
// transcendental / MFMA result" hazards only for instructions it emitted itself; written in plain C++ the packed
// forms bring nothing measurable, so the loop stays scalar.)
// Block = 128 queries of one (image, head): 4 waves x 32 queries, all waves sharing the K / V^T tiles (32 keys,
// double-buffered in LDS, register-prefetched one tile ahead).
#include "awseg_common.h"

namespace {

typedef float f32x16 __attribute__((ext_vector_type(16)));

constexpr int AT = 256;          // threads
#ifndef AWSEG_ATTN_SPLIT_WAVES
#define AWSEG_ATTN_SPLIT_WAVES 2      // minimum waves per SIMD asked of the split-operand kernel (4 was measured: see DESIGN.md)
#endif
constexpr int D = 32;            // head dim
constexpr int TK = 32;           // keys per tile
constexpr int LDK = 36;          // padded row length (floats): conflict-free ds_read_b128 at a 144-byte lane stride

// accumulator row of register r in wave half hk
__device__ __forceinline__ constexpr int acc_row(int r, int hk) { return (r & 3) + 8 * (r >> 2) + 4 * hk; }

__global__ __launch_bounds__(AT, 2)
void attention_d32_kernel(const float* __restrict__ q, const float* __restrict__ k, const float* __restrict__ v,
                          float* __restrict__ out, int nq, int nkv, int heads, float scale_log2e, int kvp)
{
    __shared__ float sK[2][TK * LDK];       // [key][hk*16 + s]   = K[key][2s + hk]
    __shared__ float sV[2][D * LDK];        // [d][hk*16 + r]     = V[row(r,hk)][d]
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int hk = lane >> 5, li = lane & 31;
    const int h = blockIdx.y, b = blockIdx.z;
    const int C = heads * D;
    const int q0 = blockIdx.x * 128 + wave * 32;
    const float* kb = k + (size_t)b * nkv * kvp + h * D;
    const float* vb = v + (size_t)b * nkv * kvp + h * D;

    // Q fragment: B operand of S^T = K Q^T, B[k = d][j = query]: lane (j = li, half hk) holds d = 2s + hk
    float qf[16];
    {
        const int qi = q0 + li;
        const float* qp = q + ((size_t)b * nq + (qi < nq ? qi : nq - 1)) * C + h * D;
#pragma unroll
        for (int s4 = 0; s4 < 8; ++s4) {
            const float4 t4 = *reinterpret_cast<const float4*>(qp + 4 * s4);       // d = 4*s4 .. 4*s4+3
            qf[2 * s4] = (hk ? t4.y : t4.x) * scale_log2e;                          // d = 4*s4 + hk
            qf[2 * s4 + 1] = (hk ? t4.w : t4.z) * scale_log2e;                      // d = 4*s4 + 2 + hk
        }
    }
    // tile staging role of this thread: key = tid / 8, 4 consecutive d = 4c..4c+3
    const int lkey = tid >> 3, lc = tid & 7;
    const size_t g_off = (size_t)lkey * kvp + 4 * lc;
    const int wk0 = lkey * LDK + 2 * lc;                 // sK: (d = 4c, 4c+2) -> hk 0, s = 2c, 2c+1 ; (4c+1, 4c+3) -> hk 1
    const int vkk = (lkey >> 2) & 1, vr = (lkey & 3) + 4 * (lkey >> 3);             // inverse of acc_row
    const int wv0 = (4 * lc) * LDK + vkk * 16 + vr;      // sV[d = 4c + e][vkk*16 + vr], e = 0..3
    auto stage = [&](int buf, const float4& kk4, const float4& vv4) {
        *reinterpret_cast<float2*>(&sK[buf][wk0]) = make_float2(kk4.x, kk4.z);
        *reinterpret_cast<float2*>(&sK[buf][wk0 + 16]) = make_float2(kk4.y, kk4.w);
        sV[buf][wv0] = vv4.x; sV[buf][wv0 + LDK] = vv4.y; sV[buf][wv0 + 2 * LDK] = vv4.z; sV[buf][wv0 + 3 * LDK] = vv4.w;
    };
    const int ntiles = nkv / TK;
    float4 kreg = *reinterpret_cast<const float4*>(kb + g_off), vreg = *reinterpret_cast<const float4*>(vb + g_off);
    stage(0, kreg, vreg);
    if (ntiles > 1) {
        kreg = *reinterpret_cast<const float4*>(kb + (size_t)TK * kvp + g_off);
        vreg = *reinterpret_cast<const float4*>(vb + (size_t)TK * kvp + g_off);
    }
    __syncthreads();

    f32x16 oacc;
#pragma unroll
    for (int r = 0; r < 16; ++r) oacc[r] = 0.f;
    float m_run = -1e30f, l_run = 0.f;
    const int a_off = li * LDK + hk * 16;                // this lane's 16 contiguous operand floats in a tile row

    for (int t = 0; t < ntiles; ++t) {
        const int buf = t & 1;
        // ---- S^T = K Q^T for this tile: rows = 32 keys, columns = this wave's 32 queries
        float kf[16];
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            const float4 t4 = *reinterpret_cast<const float4*>(&sK[buf][a_off + 4 * j]);
            kf[4 * j] = t4.x; kf[4 * j + 1] = t4.y; kf[4 * j + 2] = t4.z; kf[4 * j + 3] = t4.w;
        }
        f32x16 sacc;
#pragma unroll
        for (int r = 0; r < 16; ++r) sacc[r] = 0.f;
#pragma unroll
        for (int s = 0; s < 16; ++s) sacc = __builtin_amdgcn_mfma_f32_32x32x2f32(kf[s], qf[s], sacc, 0, 0, 0);
        // ---- the next tile's V^T fragments do not depend on the softmax: fetch them now
        float vf[16];
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            const float4 t4 = *reinterpret_cast<const float4*>(&sV[buf][a_off + 4 * j]);
            vf[4 * j] = t4.x; vf[4 * j + 1] = t4.y; vf[4 * j + 2] = t4.z; vf[4 * j + 3] = t4.w;
        }
        // ---- online softmax, one query per lane (its keys: 16 registers here + 16 in the other wave half)
        float mt = fmaxf(fmaxf(sacc[0], sacc[1]), fmaxf(sacc[2], sacc[3]));
#pragma unroll
        for (int r = 4; r < 16; r += 4) mt = fmaxf(mt, fmaxf(fmaxf(sacc[r], sacc[r + 1]), fmaxf(sacc[r + 2], sacc[r + 3])));
        mt = fmaxf(mt, __shfl_xor(mt, 32, 64));
        const float m_new = fmaxf(m_run, mt);
        const float alpha = __builtin_amdgcn_exp2f(m_run - m_new);
        float p[16], ls = 0.f;
#pragma unroll
        for (int r = 0; r < 16; ++r) { p[r] = __builtin_amdgcn_exp2f(sacc[r] - m_new); ls += p[r]; }
        ls += __shfl_xor(ls, 32, 64);
        l_run = l_run * alpha + ls;
        m_run = m_new;
        // after the first few tiles the running maxima stop moving: skip the 16 rescaling multiplies unless some
        // lane of the wave needs them (wave-uniform branch; VALU instructions are MFMA time here)
        if (__builtin_amdgcn_ballot_w64(alpha != 1.0f) != 0) {
#pragma unroll
            for (int r = 0; r < 16; ++r) oacc[r] *= alpha;
        }
        // ---- O^T += V^T P^T: A = V^T [d][key], B = P^T — the accumulator tile itself, register r = k-step r
#pragma unroll
        for (int r = 0; r < 16; ++r) oacc = __builtin_amdgcn_mfma_f32_32x32x2f32(vf[r], p[r], oacc, 0, 0, 0);
        // ---- stage tile t+1 (already in registers) into the other buffer, prefetch tile t+2
        if (t + 1 < ntiles) {
            stage(buf ^ 1, kreg, vreg);
            if (t + 2 < ntiles) {
                kreg = *reinterpret_cast<const float4*>(kb + (size_t)(t + 2) * TK * kvp + g_off);
                vreg = *reinterpret_cast<const float4*>(vb + (size_t)(t + 2) * TK * kvp + g_off);
            }
        }
        __syncthreads();
    }

    // ---- O = O^T / l: lane = query li, registers = d rows; rows 4g..4g+3 of a half are 4 consecutive d
    const int qi = q0 + li;
    if (qi < nq) {
        const float inv = 1.0f / l_run;
        float* op = out + ((size_t)b * nq + qi) * C + h * D;
#pragma unroll
        for (int g = 0; g < 4; ++g)
            *reinterpret_cast<float4*>(op + 8 * g + 4 * hk) =
                make_float4(oacc[4 * g] * inv, oacc[4 * g + 1] * inv, oacc[4 * g + 2] * inv, oacc[4 * g + 3] * inv);
    }
}

// ------------------------------------------------------------------------------------------------------------------
// The same attention on the f16 matrix cores with SPLIT operands.  v_mfma_f32_32x32x16_f16 issues 16x the multiply-adds
// per cycle of v_mfma_f32_32x32x2_f32, so a float32 product is rebuilt from three f16 products at 5.3x the rate:
//     x = xh + xl / 2048,   xh = f16(x) (11 significant bits),  xl = f16((x - xh) * 2048) (the next 11 bits)
//     x * y = xh*yh + (xh*yl + xl*yh) / 2048 + O(2^-22 |x y|)          (the xl*yl term is below float32 rounding noise)
// with two float32 accumulator tiles (main, correction) and float32 accumulation inside the MFMA throughout: 22-bit
// operands instead of 24-bit, everything else as the kernel above (transposed scores, in-register softmax, the
// probability tile reused as the B operand).  tests/test_gpu_kernels.py measures both kernels against a float64
// reference: their errors are of the same order (a few 1e-7 relative), two orders inside the 1e-4 gate.
// The probabilities are produced pre-scaled by 2^15 (folded into the exponent: exp2(s - m + 15)) so that every one that
// matters is a NORMAL f16 number; the scale cancels in O / l.  Any finite float32 q, k, v: see the range guard below.
typedef _Float16 h8 __attribute__((ext_vector_type(8)));
typedef _Float16 h2 __attribute__((ext_vector_type(2)));
typedef unsigned u32x4 __attribute__((ext_vector_type(4)));
typedef unsigned u32x2 __attribute__((ext_vector_type(2)));

constexpr int SROW = 72;         // halfs per LDS row: 32 hi | 32 lo | 8 pad  (144 B: conflict-free ds_read_b128)
constexpr float kLoScale = 2048.0f, kLoInv = 1.0f / 2048.0f;

// (a, b) -> packed f16 high parts and packed scaled residuals.  Round-toward-zero is fine: the residual is exact.
__device__ __forceinline__ void split_pair(float a, float b, unsigned& hi, unsigned& lo)
{
    const auto hp = __builtin_amdgcn_cvt_pkrtz(a, b);
    const h2 hh = __builtin_bit_cast(h2, hp);
    const float ra = (a - (float)hh.x) * kLoScale, rb = (b - (float)hh.y) * kLoScale;
    hi = __builtin_bit_cast(unsigned, hp);
    lo = __builtin_bit_cast(unsigned, __builtin_amdgcn_cvt_pkrtz(ra, rb));
}
// the same without scaling the residual: for values whose low part is a normal (or harmlessly subnormal) f16 as is
// Three instructions (wino_split.hip): v_cvt_pkrtz_f16_f32, then one mixed-precision FMA per value — x * 1.0 + (-hi) with the f16
// source widened and the sum taken in float32 (exact), rounded to f16 into one half of the destination.  (The plain form is six:
// two v_cvt_f32_f16, two subtracts, a second pack; the loop is bound by its vector instruction count: -1.6 % on stage 1.)
// Only the probabilities use it.  Q, K and V keep the x 2048 low parts: unscaled ones are subnormal below |x| = 2^-3, an ABSOLUTE
// 2^-25 that the tests with a small q against one large k (flat softmax) and with whole tensors at 1e-5 (tiny operands: 5e-6
// RELATIVE) do not allow — both were tried.
__device__ __forceinline__ void split_pair_unscaled(float a, float b, unsigned& hi, unsigned& lo)
{
    hi = __builtin_bit_cast(unsigned, __builtin_amdgcn_cvt_pkrtz(a, b));
    asm("v_fma_mixlo_f16 %0, %1, 1.0, -%3 op_sel_hi:[0,0,1]\n\t"
        "v_fma_mixhi_f16 %0, %2, 1.0, -%3 op_sel:[0,0,1] op_sel_hi:[0,0,1]\n\t"
        "s_nop 1"          // the low parts go straight into an MFMA operand register: two wait states hipcc does not pad behind inline asm
        : "=&v"(lo) : "v"(a), "v"(b), "v"(hi));
}
__device__ __forceinline__ void split8_unscaled(const float* x, h8& hi, h8& lo)
{
    u32x4 H, L;
#pragma unroll
    for (int i = 0; i < 4; ++i) { unsigned h, l; split_pair_unscaled(x[2 * i], x[2 * i + 1], h, l); H[i] = h; L[i] = l; }
    hi = __builtin_bit_cast(h8, H); lo = __builtin_bit_cast(h8, L);
}
__device__ __forceinline__ void split8(const float* x, h8& hi, h8& lo)
{
    u32x4 H, L;
#pragma unroll
    for (int i = 0; i < 4; ++i) { unsigned h, l; split_pair(x[2 * i], x[2 * i + 1], h, l); H[i] = h; L[i] = l; }
    hi = __builtin_bit_cast(h8, H); lo = __builtin_bit_cast(h8, L);
}
// k-slot of tile key `key` in the PV product: element j of lane half h of k-step s is accumulator row
// 16 s + 8 (j >> 2) + 4 h + (j & 3) (the C/D layout read as an operand); slot = 16 s + 8 h + j
__device__ __forceinline__ constexpr int pv_slot(int key)
{
    return (key & 16) | (((key >> 2) & 1) << 3) | (((key >> 3) & 1) << 2) | (key & 3);
}

// Operand range guard (the same scheme as gemm_split.hip).  A pass runs OPTIMISTICALLY with unscaled operands while
// every thread tracks max|.| of the q (after the softmax scale), k and v values it splits; if the block has met a value
// >= 2^15 — where f16(x) or the scaled low part leaves the f16 range — nothing is stored and the block runs the whole
// query tile again with q * 2^-eq, k * 2^-ek, v * 2^-ev (exact powers of two from the observed maxima), multiplying
// 2^(eq+ek) back into the scores and 2^ev into the output.  LayerNorm -> Linear outputs never take the second pass.
constexpr float kSplitLimit = 32768.0f;
struct attn_scales { float sq, sk, sv, bq, bk, bv; };           // forward scales 2^-e and their inverses 2^e

__device__ __forceinline__ float amax4(float m, const float4& t)
{
    m = __builtin_fmaxf(__builtin_fmaxf(m, __builtin_fabsf(t.x)), __builtin_fabsf(t.y));
    return __builtin_fmaxf(__builtin_fmaxf(m, __builtin_fabsf(t.z)), __builtin_fabsf(t.w));
}

typedef __bf16 bf8 __attribute__((ext_vector_type(8)));
__device__ __forceinline__ unsigned pack_bf16(float a, float b)
{
    typedef __bf16 bf2 __attribute__((ext_vector_type(2)));
    typedef float f2 __attribute__((ext_vector_type(2)));
    return __builtin_bit_cast(unsigned, __builtin_convertvector(f2{a, b}, bf2));
}
__device__ __forceinline__ h8 pack8_bf16(const float* x)
{
    u32x4 H;
#pragma unroll
    for (int i = 0; i < 4; ++i) H[i] = pack_bf16(x[2 * i], x[2 * i + 1]);
    return __builtin_bit_cast(h8, H);
}
#define MFMA_BF(A, B, C) __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf8, A), __builtin_bit_cast(bf8, B), C, 0, 0, 0)

// returns true (block-uniform) when the optimistic pass met an out-of-range operand and stored nothing.
// BF16: ONE bf16 MFMA per product tile (q, k, v and the probabilities rounded to bf16, float32 accumulation and softmax):
// BASELINE config 5's bf16 MFMA path; no range guard (bf16 has float32's exponent range).
template <bool SCALED, bool BF16 = false>
__device__ __forceinline__ bool attn_split_pass(const float* __restrict__ q, const float* __restrict__ k, const float* __restrict__ v,
                                                float* __restrict__ out, int nq, int nkv, int heads, float scale_log2e, int kvp,
                                                _Float16 (*sK)[TK * SROW], _Float16 (*sV)[D * SROW], unsigned* sMax, const attn_scales sc)
{
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int hk = lane >> 5, li = lane & 31;
    const int h = blockIdx.y, b = blockIdx.z;
    const int C = heads * D;
    const int q0 = blockIdx.x * 128 + wave * 32;
    const float* kb = k + (size_t)b * nkv * kvp + h * D;
    const float* vb = v + (size_t)b * nkv * kvp + h * D;
    float qmax = 0.f, kmax = 0.f, vmax = 0.f;

    // Q fragments: B operand of S^T = K Q^T, lane (query li, half hk) holds d = 16 s + 8 hk + j
    h8 qh[2], ql[2];
    {
        const int qi = q0 + li;
        const float* qp = q + ((size_t)b * nq + (qi < nq ? qi : nq - 1)) * C + h * D;
        const float qs = SCALED ? scale_log2e * sc.sq : scale_log2e;           // (x * s) * 2^-e == x * (s * 2^-e): exact factor
#pragma unroll
        for (int s = 0; s < 2; ++s) {
            const float4 a4 = *reinterpret_cast<const float4*>(qp + 16 * s + 8 * hk);
            const float4 b4 = *reinterpret_cast<const float4*>(qp + 16 * s + 8 * hk + 4);
            const float x[8] = { a4.x * qs, a4.y * qs, a4.z * qs, a4.w * qs, b4.x * qs, b4.y * qs, b4.z * qs, b4.w * qs };
            if (BF16) { qh[s] = pack8_bf16(x); continue; }
            if (!SCALED) {
#pragma unroll
                for (int i = 0; i < 8; ++i) qmax = __builtin_fmaxf(qmax, __builtin_fabsf(x[i]));
            }
            split8(x, qh[s], ql[s]);
        }
    }
    // tile staging role of this thread: key = tid / 8, 4 consecutive d = 4c .. 4c+3
    const int lkey = tid >> 3, lc = tid & 7;
    const size_t g_off = (size_t)lkey * kvp + 4 * lc;
    const int wk0 = lkey * SROW + 4 * lc;
    const int wv0 = (4 * lc) * SROW + pv_slot(lkey);
    auto stage = [&](int buf, float4 kk4, float4 vv4) {
        if (BF16) {
            u32x2 H;
            H[0] = pack_bf16(kk4.x, kk4.y); H[1] = pack_bf16(kk4.z, kk4.w);
            *reinterpret_cast<u32x2*>(&sK[buf][wk0]) = H;
            const unsigned v01 = pack_bf16(vv4.x, vv4.y), v23 = pack_bf16(vv4.z, vv4.w);
            unsigned short* svb = reinterpret_cast<unsigned short*>(&sV[buf][wv0]);
            svb[0] = (unsigned short)v01; svb[SROW] = (unsigned short)(v01 >> 16);
            svb[2 * SROW] = (unsigned short)v23; svb[3 * SROW] = (unsigned short)(v23 >> 16);
            return;
        }
        if (SCALED) {
            kk4.x *= sc.sk; kk4.y *= sc.sk; kk4.z *= sc.sk; kk4.w *= sc.sk;
            vv4.x *= sc.sv; vv4.y *= sc.sv; vv4.z *= sc.sv; vv4.w *= sc.sv;
        } else {
            kmax = amax4(kmax, kk4); vmax = amax4(vmax, vv4);
        }
        u32x2 H, L; unsigned hh, ll;
        split_pair(kk4.x, kk4.y, hh, ll); H[0] = hh; L[0] = ll;
        split_pair(kk4.z, kk4.w, hh, ll); H[1] = hh; L[1] = ll;
        *reinterpret_cast<u32x2*>(&sK[buf][wk0]) = H;
        *reinterpret_cast<u32x2*>(&sK[buf][wk0 + 32]) = L;
        unsigned vh01, vl01, vh23, vl23;
        split_pair(vv4.x, vv4.y, vh01, vl01);
        split_pair(vv4.z, vv4.w, vh23, vl23);
        unsigned short* sv = reinterpret_cast<unsigned short*>(&sV[buf][wv0]);
        sv[0] = (unsigned short)vh01;            sv[32] = (unsigned short)vl01;
        sv[SROW] = (unsigned short)(vh01 >> 16); sv[SROW + 32] = (unsigned short)(vl01 >> 16);
        sv[2 * SROW] = (unsigned short)vh23;     sv[2 * SROW + 32] = (unsigned short)vl23;
        sv[3 * SROW] = (unsigned short)(vh23 >> 16); sv[3 * SROW + 32] = (unsigned short)(vl23 >> 16);
    };
    const int ntiles = nkv / TK;
    float4 kreg = *reinterpret_cast<const float4*>(kb + g_off), vreg = *reinterpret_cast<const float4*>(vb + g_off);
    stage(0, kreg, vreg);
    if (ntiles > 1) {
        kreg = *reinterpret_cast<const float4*>(kb + (size_t)TK * kvp + g_off);
        vreg = *reinterpret_cast<const float4*>(vb + (size_t)TK * kvp + g_off);
    }
    __syncthreads();

    f32x16 om, oc;                   // O^T: main and correction (x 2048) accumulators
#pragma unroll
    for (int r = 0; r < 16; ++r) { om[r] = 0.f; oc[r] = 0.f; }
    float m_run = -1e30f, l_run = 0.f;
    const int a_off = li * SROW + 8 * hk;

    for (int t = 0; t < ntiles; ++t) {
        const int buf = t & 1;
        const h8* kr = reinterpret_cast<const h8*>(&sK[buf][a_off]);
        const h8 kh0 = kr[0], kh1 = kr[2], kl0 = kr[4], kl1 = kr[6];          // +0, +16, +32, +48 halfs
        f32x16 sm, scx;
#pragma unroll
        for (int r = 0; r < 16; ++r) { sm[r] = 0.f; scx[r] = 0.f; }
        if (BF16) {
            sm = MFMA_BF(kh0, qh[0], sm);
            sm = MFMA_BF(kh1, qh[1], sm);
        } else {
            sm = __builtin_amdgcn_mfma_f32_32x32x16_f16(kh0, qh[0], sm, 0, 0, 0);
            scx = __builtin_amdgcn_mfma_f32_32x32x16_f16(kh0, ql[0], scx, 0, 0, 0);
            sm = __builtin_amdgcn_mfma_f32_32x32x16_f16(kh1, qh[1], sm, 0, 0, 0);
            scx = __builtin_amdgcn_mfma_f32_32x32x16_f16(kh1, ql[1], scx, 0, 0, 0);
            scx = __builtin_amdgcn_mfma_f32_32x32x16_f16(kl0, qh[0], scx, 0, 0, 0);
            scx = __builtin_amdgcn_mfma_f32_32x32x16_f16(kl1, qh[1], scx, 0, 0, 0);
        }
        const h8* vr = reinterpret_cast<const h8*>(&sV[buf][a_off]);
        const h8 vh0 = vr[0], vh1 = vr[2], vl0 = vr[4], vl1 = vr[6];
        // ---- online softmax, one query per lane
        float s[16];
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            s[r] = fmaf(scx[r], kLoInv, sm[r]);
            if (SCALED) s[r] = s[r] * sc.bq * sc.bk;
        }
        float mt = fmaxf(fmaxf(s[0], s[1]), fmaxf(s[2], s[3]));
#pragma unroll
        for (int r = 4; r < 16; r += 4) mt = fmaxf(mt, fmaxf(fmaxf(s[r], s[r + 1]), fmaxf(s[r + 2], s[r + 3])));
        mt = fmaxf(mt, __shfl_xor(mt, 32, 64));
        const float m_new = fmaxf(m_run, mt);
        const float alpha = __builtin_amdgcn_exp2f(m_run - m_new);
        const float m_sh = m_new - 15.0f;                  // probabilities scaled by 2^15 (see the header)
        float p[16], ls = 0.f;
#pragma unroll
        for (int r = 0; r < 16; ++r) { p[r] = __builtin_amdgcn_exp2f(s[r] - m_sh); ls += p[r]; }
        ls += __shfl_xor(ls, 32, 64);
        l_run = l_run * alpha + ls;
        m_run = m_new;
        if (__builtin_amdgcn_ballot_w64(alpha != 1.0f) != 0) {
#pragma unroll
            for (int r = 0; r < 16; ++r) { om[r] *= alpha; oc[r] *= alpha; }
        }
        // ---- O^T += V^T P^T: the probability tile, split, is the B operand: registers 8 s .. 8 s + 7 = k-step s
        // The probabilities are pre-scaled to (0, 2^15], so their low parts need no scaling: p - f16(p) is a normal f16
        // for p >= 2^-3 and below that a subnormal worth < 2^-40 of the row maximum (the f16 matrix cores keep
        // subnormals) -> V_hi P_lo goes to the main accumulator, only V_lo' P_hi to the scaled correction.
        if (BF16) {
            om = MFMA_BF(vh0, pack8_bf16(p), om);
            om = MFMA_BF(vh1, pack8_bf16(p + 8), om);
            if (t + 1 < ntiles) {
                stage(buf ^ 1, kreg, vreg);
                if (t + 2 < ntiles) {
                    kreg = *reinterpret_cast<const float4*>(kb + (size_t)(t + 2) * TK * kvp + g_off);
                    vreg = *reinterpret_cast<const float4*>(vb + (size_t)(t + 2) * TK * kvp + g_off);
                }
            }
            __syncthreads();
            continue;
        }
        h8 ph0, pl0, ph1, pl1;
        split8_unscaled(p, ph0, pl0);
        split8_unscaled(p + 8, ph1, pl1);
        om = __builtin_amdgcn_mfma_f32_32x32x16_f16(vh0, ph0, om, 0, 0, 0);
        oc = __builtin_amdgcn_mfma_f32_32x32x16_f16(vl0, ph0, oc, 0, 0, 0);
        om = __builtin_amdgcn_mfma_f32_32x32x16_f16(vh1, ph1, om, 0, 0, 0);
        oc = __builtin_amdgcn_mfma_f32_32x32x16_f16(vl1, ph1, oc, 0, 0, 0);
        om = __builtin_amdgcn_mfma_f32_32x32x16_f16(vh0, pl0, om, 0, 0, 0);
        om = __builtin_amdgcn_mfma_f32_32x32x16_f16(vh1, pl1, om, 0, 0, 0);
        if (t + 1 < ntiles) {
            stage(buf ^ 1, kreg, vreg);
            if (t + 2 < ntiles) {
                kreg = *reinterpret_cast<const float4*>(kb + (size_t)(t + 2) * TK * kvp + g_off);
                vreg = *reinterpret_cast<const float4*>(vb + (size_t)(t + 2) * TK * kvp + g_off);
            }
        }
        __syncthreads();
    }

    if (!SCALED && !BF16) {
        // report (rare, divergent) -> barrier -> block-uniform verdict
        if (qmax >= kSplitLimit) atomicMax(&sMax[0], __builtin_bit_cast(unsigned, qmax));
        if (kmax >= kSplitLimit) atomicMax(&sMax[1], __builtin_bit_cast(unsigned, kmax));
        if (vmax >= kSplitLimit) atomicMax(&sMax[2], __builtin_bit_cast(unsigned, vmax));
        __syncthreads();
        if ((sMax[0] | sMax[1] | sMax[2]) != 0u) return true;
    }
    const int qi = q0 + li;
    if (qi < nq) {
        const float inv = 1.0f / l_run;
        float* op = out + ((size_t)b * nq + qi) * C + h * D;
#pragma unroll
        for (int g = 0; g < 4; ++g) {
            float4 o4 = make_float4(fmaf(oc[4 * g], kLoInv, om[4 * g]) * inv, fmaf(oc[4 * g + 1], kLoInv, om[4 * g + 1]) * inv,
                                    fmaf(oc[4 * g + 2], kLoInv, om[4 * g + 2]) * inv, fmaf(oc[4 * g + 3], kLoInv, om[4 * g + 3]) * inv);
            if (SCALED) { o4.x *= sc.bv; o4.y *= sc.bv; o4.z *= sc.bv; o4.w *= sc.bv; }
            *reinterpret_cast<float4*>(op + 8 * g + 4 * hk) = o4;
        }
    }
    return false;
}

// exponent e >= 0 with max * 2^-e in [2^13, 2^14) when the maximum (float bits) left the split range, else 0
__device__ __forceinline__ int guard_exponent(unsigned maxbits)
{
    const int ex = (int)(maxbits >> 23) & 0xff;
    if (maxbits == 0u || ex == 0xff) return 0;               // in range, or Inf / NaN (nothing to rescue: they propagate)
    return ex - 127 - 13;
}

__global__ __launch_bounds__(AT, AWSEG_ATTN_SPLIT_WAVES)
void attention_d32_split_kernel(const float* __restrict__ q, const float* __restrict__ k, const float* __restrict__ v,
                                float* __restrict__ out, int nq, int nkv, int heads, float scale_log2e, int kvp)
{
    __shared__ __attribute__((aligned(16))) _Float16 sK[2][TK * SROW];      // [key][hi d 0..31 | lo d 0..31]
    __shared__ __attribute__((aligned(16))) _Float16 sV[2][D * SROW];       // [d][hi slot 0..31 | lo slot 0..31]
    __shared__ unsigned sMax[3];                                             // max |q*scale|, |k|, |v| bits, when >= 2^15
    if (threadIdx.x < 3) sMax[threadIdx.x] = 0u;                             // ordered by the pass's first barrier
    const attn_scales one = {1.f, 1.f, 1.f, 1.f, 1.f, 1.f};
    if (!attn_split_pass<false>(q, k, v, out, nq, nkv, heads, scale_log2e, kvp, sK, sV, sMax, one)) return;
    const int eq = guard_exponent(sMax[0]), ek = guard_exponent(sMax[1]), ev = guard_exponent(sMax[2]);
    auto p2 = [](int e) { return __builtin_bit_cast(float, (unsigned)(127 + e) << 23); };            // |e| <= 114
    const attn_scales sc = {p2(-eq), p2(-ek), p2(-ev), p2(eq), p2(ek), p2(ev)};
    attn_split_pass<true>(q, k, v, out, nq, nkv, heads, scale_log2e, kvp, sK, sV, sMax, sc);
}

// ------------------------------------------------------------------------------------------------------------------
// The split-operand kernel reading a PREPARED key / value image.  attention_d32_split_kernel splits every key tile again in every
// query block — 1 024 blocks per image in MiT stage 1, each spending ~36 of its ~150 vector instructions and 10 of its 19 LDS
// instructions per tile on the same 2 048 keys, in a loop that is bound by its vector instruction count (a software-pipelined
// variant that ran every MFMA beside vector work of another tile was 6 % SLOWER: DESIGN.md 11).  Here one small kernel per launch
// writes, for every (image, head), the tiles exactly as the query blocks want them in LDS — [32 keys][32 hi | 32 lo | pad] then
// V^T [32 d][32 hi | 32 lo | pad] in the PV slot order, f16, scaled low parts, 9 216 bytes a tile — and the query blocks fetch a
// tile with nine LDS-DMA instructions per block (global -> LDS, no registers, no vector work), one tile ahead.
// Operand range: the image kernel sees every key / value of its (image, head) and settles their exponents itself (2^-ek, 2^-ev
// from the maxima when they reach 2^15, exponents left in a table); the query blocks keep the optimistic pass / scaled second pass
// for q alone.  Same arithmetic per element as attention_d32_split_kernel: same values.
constexpr int IMG_TILE_HALFS = (TK + D) * SROW;                    // 4 608 halfs

__device__ __forceinline__ uint32_t at_lds_addr(const void* p) { return (uint32_t)(uintptr_t)(__attribute__((address_space(3))) const void*)p; }
__device__ __forceinline__ void at_dma16(__amdgpu_buffer_rsrc_t rsrc, uint32_t voff, uint32_t soff, uint32_t lds_base)
{
    asm volatile("s_mov_b32 m0, %3\n\ts_nop 0\n\tbuffer_load_dwordx4 %0, %1, %2 offen lds" : : "v"(voff), "s"(rsrc), "s"(soff), "s"(lds_base) : "m0", "memory");
}

__global__ __launch_bounds__(AT)
void attn_kv_image_kernel(const float* __restrict__ k, const float* __restrict__ v, int kvp, int nkv, int heads,
                          _Float16* __restrict__ img, int* __restrict__ exps)
{
    __shared__ unsigned sM[2];
    const int tid = threadIdx.x, h = blockIdx.x, b = blockIdx.y;
    const float* kb = k + (size_t)b * nkv * kvp + h * D;
    const float* vb = v + (size_t)b * nkv * kvp + h * D;
    const int lkey = tid >> 3, lc = tid & 7;
    const size_t g_off = (size_t)lkey * kvp + 4 * lc;
    const int wk0 = lkey * SROW + 4 * lc;
    const int wv0 = TK * SROW + (4 * lc) * SROW + pv_slot(lkey);
    const int ntiles = nkv / TK;
    if (tid < 2) sM[tid] = 0u;
    __syncthreads();
    float kmax = 0.f, vmax = 0.f;
    for (int t = 0; t < ntiles; ++t) {
        kmax = amax4(kmax, *reinterpret_cast<const float4*>(kb + (size_t)t * TK * kvp + g_off));
        vmax = amax4(vmax, *reinterpret_cast<const float4*>(vb + (size_t)t * TK * kvp + g_off));
    }
    if (kmax >= kSplitLimit) atomicMax(&sM[0], __builtin_bit_cast(unsigned, kmax));
    if (vmax >= kSplitLimit) atomicMax(&sM[1], __builtin_bit_cast(unsigned, vmax));
    __syncthreads();
    const int ek = guard_exponent(sM[0]), ev = guard_exponent(sM[1]);
    auto p2 = [](int e) { return __builtin_bit_cast(float, (unsigned)(127 + e) << 23); };
    const float sk = p2(-ek), sv = p2(-ev);
    if (tid == 0) { exps[((size_t)b * heads + h) * 2] = ek; exps[((size_t)b * heads + h) * 2 + 1] = ev; }
    _Float16* base = img + ((size_t)b * heads + h) * ntiles * IMG_TILE_HALFS;
    for (int t = 0; t < ntiles; ++t) {
        float4 kk4 = *reinterpret_cast<const float4*>(kb + (size_t)t * TK * kvp + g_off);
        float4 vv4 = *reinterpret_cast<const float4*>(vb + (size_t)t * TK * kvp + g_off);
        kk4.x *= sk; kk4.y *= sk; kk4.z *= sk; kk4.w *= sk;        // (x 1.0 when in range: exact)
        vv4.x *= sv; vv4.y *= sv; vv4.z *= sv; vv4.w *= sv;
        _Float16* T = base + (size_t)t * IMG_TILE_HALFS;
        u32x2 H, L; unsigned hh, ll;
        split_pair(kk4.x, kk4.y, hh, ll); H[0] = hh; L[0] = ll;
        split_pair(kk4.z, kk4.w, hh, ll); H[1] = hh; L[1] = ll;
        *reinterpret_cast<u32x2*>(&T[wk0]) = H;
        *reinterpret_cast<u32x2*>(&T[wk0 + 32]) = L;
        unsigned vh01, vl01, vh23, vl23;
        split_pair(vv4.x, vv4.y, vh01, vl01);
        split_pair(vv4.z, vv4.w, vh23, vl23);
        unsigned short* sv16 = reinterpret_cast<unsigned short*>(&T[wv0]);
        sv16[0] = (unsigned short)vh01;            sv16[32] = (unsigned short)vl01;
        sv16[SROW] = (unsigned short)(vh01 >> 16); sv16[SROW + 32] = (unsigned short)(vl01 >> 16);
        sv16[2 * SROW] = (unsigned short)vh23;     sv16[2 * SROW + 32] = (unsigned short)vl23;
        sv16[3 * SROW] = (unsigned short)(vh23 >> 16); sv16[3 * SROW + 32] = (unsigned short)(vl23 >> 16);
    }
}

// SCALED: some operand carries a power-of-two scale (sc); QGUARD: track max|q| and report a block whose q left the range
template <bool SCALED, bool QGUARD>
__device__ __forceinline__ bool attn_split_pass_img(const float* __restrict__ q, const _Float16* __restrict__ img_bh, float* __restrict__ out,
                                                    int nq, int nkv, int heads, float scale_log2e, _Float16 (*sKV)[IMG_TILE_HALFS],
                                                    unsigned* sMax, const attn_scales sc)
{
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int hk = lane >> 5, li = lane & 31;
    const int h = blockIdx.y, b = blockIdx.z;
    const int C = heads * D;
    const int q0 = blockIdx.x * 128 + wave * 32;
    float qmax = 0.f;
    h8 qh[2], ql[2];
    {
        const int qi = q0 + li;
        const float* qp = q + ((size_t)b * nq + (qi < nq ? qi : nq - 1)) * C + h * D;
        const float qs = SCALED ? scale_log2e * sc.sq : scale_log2e;
#pragma unroll
        for (int s = 0; s < 2; ++s) {
            const float4 a4 = *reinterpret_cast<const float4*>(qp + 16 * s + 8 * hk);
            const float4 b4 = *reinterpret_cast<const float4*>(qp + 16 * s + 8 * hk + 4);
            const float x[8] = { a4.x * qs, a4.y * qs, a4.z * qs, a4.w * qs, b4.x * qs, b4.y * qs, b4.z * qs, b4.w * qs };
            if (QGUARD) {
#pragma unroll
                for (int i = 0; i < 8; ++i) qmax = __builtin_fmaxf(qmax, __builtin_fabsf(x[i]));
            }
            split8(x, qh[s], ql[s]);
        }
    }
    const int ntiles = nkv / TK;
    const __amdgpu_buffer_rsrc_t rsrc = __builtin_amdgcn_make_buffer_rsrc((void*)img_bh, 0, ntiles * IMG_TILE_HALFS * 2, 0x00020000);
    const uint32_t lds0 = __builtin_amdgcn_readfirstlane(at_lds_addr(&sKV[0][0]));
    const uint32_t wave_u = __builtin_amdgcn_readfirstlane(wave);
    // a tile = nine 1 KB LDS-DMA instructions: wave w issues chunks w and w + 4, wave 0 the ninth
    auto dma_tile = [&](int t, int buf) {
        const uint32_t so = (uint32_t)t * (uint32_t)(IMG_TILE_HALFS * 2), lb = lds0 + (uint32_t)buf * (uint32_t)(IMG_TILE_HALFS * 2);
        at_dma16(rsrc, (uint32_t)(lane * 16) + wave_u * 1024u, so, lb + wave_u * 1024u);
        at_dma16(rsrc, (uint32_t)(lane * 16) + (wave_u + 4u) * 1024u, so, lb + (wave_u + 4u) * 1024u);
        if (wave_u == 0) at_dma16(rsrc, (uint32_t)(lane * 16) + 8192u, so, lb + 8192u);
    };
    dma_tile(0, 0);
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();

    f32x16 om, oc;
#pragma unroll
    for (int r = 0; r < 16; ++r) { om[r] = 0.f; oc[r] = 0.f; }
    float m_run = -1e30f, l_run = 0.f;
    const int a_off = li * SROW + 8 * hk;

    for (int t = 0; t < ntiles; ++t) {
        const int buf = t & 1;
        if (t + 1 < ntiles) dma_tile(t + 1, buf ^ 1);              // (that buffer's readers passed the barrier behind tile t - 1)
        const h8* kr = reinterpret_cast<const h8*>(&sKV[buf][a_off]);
        const h8 kh0 = kr[0], kh1 = kr[2], kl0 = kr[4], kl1 = kr[6];
        f32x16 sm, scx;
#pragma unroll
        for (int r = 0; r < 16; ++r) { sm[r] = 0.f; scx[r] = 0.f; }
        sm = __builtin_amdgcn_mfma_f32_32x32x16_f16(kh0, qh[0], sm, 0, 0, 0);
        scx = __builtin_amdgcn_mfma_f32_32x32x16_f16(kh0, ql[0], scx, 0, 0, 0);
        sm = __builtin_amdgcn_mfma_f32_32x32x16_f16(kh1, qh[1], sm, 0, 0, 0);
        scx = __builtin_amdgcn_mfma_f32_32x32x16_f16(kh1, ql[1], scx, 0, 0, 0);
        scx = __builtin_amdgcn_mfma_f32_32x32x16_f16(kl0, qh[0], scx, 0, 0, 0);
        scx = __builtin_amdgcn_mfma_f32_32x32x16_f16(kl1, qh[1], scx, 0, 0, 0);
        const h8* vr = reinterpret_cast<const h8*>(&sKV[buf][TK * SROW + a_off]);
        const h8 vh0 = vr[0], vh1 = vr[2], vl0 = vr[4], vl1 = vr[6];
        float s[16];
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            s[r] = fmaf(scx[r], kLoInv, sm[r]);
            if (SCALED) s[r] = s[r] * sc.bq * sc.bk;
        }
        float mt = fmaxf(fmaxf(s[0], s[1]), fmaxf(s[2], s[3]));
#pragma unroll
        for (int r = 4; r < 16; r += 4) mt = fmaxf(mt, fmaxf(fmaxf(s[r], s[r + 1]), fmaxf(s[r + 2], s[r + 3])));
        mt = fmaxf(mt, __shfl_xor(mt, 32, 64));
        const float m_new = fmaxf(m_run, mt);
        const float alpha = __builtin_amdgcn_exp2f(m_run - m_new);
        const float m_sh = m_new - 15.0f;
        float p[16], ls = 0.f;
#pragma unroll
        for (int r = 0; r < 16; ++r) { p[r] = __builtin_amdgcn_exp2f(s[r] - m_sh); ls += p[r]; }
        ls += __shfl_xor(ls, 32, 64);
        l_run = l_run * alpha + ls;
        m_run = m_new;
        if (__builtin_amdgcn_ballot_w64(alpha != 1.0f) != 0) {
#pragma unroll
            for (int r = 0; r < 16; ++r) { om[r] *= alpha; oc[r] *= alpha; }
        }
        h8 ph0, pl0, ph1, pl1;
        split8_unscaled(p, ph0, pl0);
        split8_unscaled(p + 8, ph1, pl1);
        om = __builtin_amdgcn_mfma_f32_32x32x16_f16(vh0, ph0, om, 0, 0, 0);
        oc = __builtin_amdgcn_mfma_f32_32x32x16_f16(vl0, ph0, oc, 0, 0, 0);
        om = __builtin_amdgcn_mfma_f32_32x32x16_f16(vh1, ph1, om, 0, 0, 0);
        oc = __builtin_amdgcn_mfma_f32_32x32x16_f16(vl1, ph1, oc, 0, 0, 0);
        om = __builtin_amdgcn_mfma_f32_32x32x16_f16(vh0, pl0, om, 0, 0, 0);
        om = __builtin_amdgcn_mfma_f32_32x32x16_f16(vh1, pl1, om, 0, 0, 0);
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");          // this wave's share of tile t + 1 has landed
        __syncthreads();
    }

    if (QGUARD) {
        if (qmax >= kSplitLimit) atomicMax(&sMax[0], __builtin_bit_cast(unsigned, qmax));
        __syncthreads();
        if (sMax[0] != 0u) return true;
    }
    const int qi = q0 + li;
    if (qi < nq) {
        const float inv = 1.0f / l_run;
        float* op = out + ((size_t)b * nq + qi) * C + h * D;
#pragma unroll
        for (int g = 0; g < 4; ++g) {
            float4 o4 = make_float4(fmaf(oc[4 * g], kLoInv, om[4 * g]) * inv, fmaf(oc[4 * g + 1], kLoInv, om[4 * g + 1]) * inv,
                                    fmaf(oc[4 * g + 2], kLoInv, om[4 * g + 2]) * inv, fmaf(oc[4 * g + 3], kLoInv, om[4 * g + 3]) * inv);
            if (SCALED) { o4.x *= sc.bv; o4.y *= sc.bv; o4.z *= sc.bv; o4.w *= sc.bv; }
            *reinterpret_cast<float4*>(op + 8 * g + 4 * hk) = o4;
        }
    }
    return false;
}

__global__ __launch_bounds__(AT, AWSEG_ATTN_SPLIT_WAVES)
void attention_d32_split_img_kernel(const float* __restrict__ q, const _Float16* __restrict__ img, const int* __restrict__ exps,
                                    float* __restrict__ out, int nq, int nkv, int heads, float scale_log2e)
{
    __shared__ __attribute__((aligned(16))) _Float16 sKV[2][IMG_TILE_HALFS];
    __shared__ unsigned sMax[1];
    const int h = blockIdx.y, b = blockIdx.z;
    if (threadIdx.x == 0) sMax[0] = 0u;                              // ordered by the pass's first barrier
    const _Float16* img_bh = img + ((size_t)b * heads + h) * (size_t)(nkv / TK) * IMG_TILE_HALFS;
    const int ek = exps[((size_t)b * heads + h) * 2], ev = exps[((size_t)b * heads + h) * 2 + 1];
    auto p2 = [](int e) { return __builtin_bit_cast(float, (unsigned)(127 + e) << 23); };
    attn_scales sc = {1.f, p2(-ek), p2(-ev), 1.f, p2(ek), p2(ev)};
    bool redo;
    if ((ek | ev) == 0) redo = attn_split_pass_img<false, true>(q, img_bh, out, nq, nkv, heads, scale_log2e, sKV, sMax, sc);
    else redo = attn_split_pass_img<true, true>(q, img_bh, out, nq, nkv, heads, scale_log2e, sKV, sMax, sc);
    if (!redo) return;
    const int eq = guard_exponent(sMax[0]);
    sc.sq = p2(-eq); sc.bq = p2(eq);
    __syncthreads();
    attn_split_pass_img<true, false>(q, img_bh, out, nq, nkv, heads, scale_log2e, sKV, sMax, sc);
}

__global__ __launch_bounds__(AT, 2)
void attention_d32_bf16_kernel(const float* __restrict__ q, const float* __restrict__ k, const float* __restrict__ v,
                               float* __restrict__ out, int nq, int nkv, int heads, float scale_log2e, int kvp)
{
    __shared__ __attribute__((aligned(16))) _Float16 sK[2][TK * SROW];
    __shared__ __attribute__((aligned(16))) _Float16 sV[2][D * SROW];
    __shared__ unsigned sMax[3];
    const attn_scales one = {1.f, 1.f, 1.f, 1.f, 1.f, 1.f};
    attn_split_pass<false, true>(q, k, v, out, nq, nkv, heads, scale_log2e, kvp, sK, sV, sMax, one);
}

}  // namespace

namespace {
template <typename K>
int launch_attention(K kernel, const float* q, const float* k, const float* v, float* out, int batch, int heads,
                     int n_queries, int n_keys, float scale, awseg_stream_t stream, int kv_pitch = 0)
{
    if (batch == 0 || n_queries == 0) return 0;
    if (!q || !k || !v || !out || batch < 0 || heads < 1 || n_queries < 0 || n_keys < TK) return AWSEG_EINVAL;
    if (n_keys % TK) return AWSEG_ERANGE;                 // whole key tiles only (MiT: the key grid is (H/32) x (W/32) ... x sr^-2)
    if (heads > 65535 || batch > 65535) return AWSEG_ERANGE;
    if (((uintptr_t)q & 15) || ((uintptr_t)k & 15) || ((uintptr_t)v & 15) || ((uintptr_t)out & 15)) return AWSEG_EALIGN;
    if (kv_pitch == 0) kv_pitch = heads * D;
    if (kv_pitch < heads * D || (kv_pitch & 3)) return AWSEG_EINVAL;
    dim3 grid((unsigned)((n_queries + 127) / 128), (unsigned)heads, (unsigned)batch);
    hipLaunchKernelGGL(kernel, grid, dim3(AT), 0, awseg_s(stream), q, k, v, out, n_queries, n_keys, heads,
                       scale * 1.4426950408889634f, kv_pitch);
    AWSEG_LAUNCH_CHECK();
    return 0;
}
}  // namespace

AWSEG_API int awseg_attention_d32(const float* q, const float* k, const float* v, float* out, int batch, int heads,
                                  int n_queries, int n_keys, float scale, awseg_stream_t stream)
{
    return launch_attention(attention_d32_kernel, q, k, v, out, batch, heads, n_queries, n_keys, scale, stream);
}

AWSEG_API int awseg_attention_d32_split(const float* q, const float* k, const float* v, float* out, int batch, int heads,
                                        int n_queries, int n_keys, float scale, awseg_stream_t stream)
{
    return launch_attention(attention_d32_split_kernel, q, k, v, out, batch, heads, n_queries, n_keys, scale, stream);
}

AWSEG_API int awseg_attention_d32_bf16(const float* q, const float* k, const float* v, float* out, int batch, int heads,
                                       int n_queries, int n_keys, float scale, awseg_stream_t stream)
{
    return launch_attention(attention_d32_bf16_kernel, q, k, v, out, batch, heads, n_queries, n_keys, scale, stream);
}

// workspace of the image form: [exponent table: 2 ints per (image, head), padded to 256 bytes][tiles: 9 216 bytes each]
AWSEG_API int64_t awseg_attention_d32_split_workspace(int batch, int heads, int n_keys)
{
    if (batch < 0 || heads < 0 || n_keys < 0) return 0;
    const int64_t tab = ((int64_t)batch * heads * 2 * 4 + 255) / 256 * 256;
    return tab + (int64_t)batch * heads * (n_keys / TK) * IMG_TILE_HALFS * 2;
}

AWSEG_API int awseg_attention_d32_split_ws(const float* q, const float* k, const float* v, int kv_pitch, float* out, int batch, int heads,
                                           int n_queries, int n_keys, float scale, void* workspace, awseg_stream_t stream)
{
    if (batch == 0 || n_queries == 0) return 0;
    if (!q || !k || !v || !out || !workspace || batch < 0 || heads < 1 || n_queries < 0 || n_keys < TK) return AWSEG_EINVAL;
    if (n_keys % TK) return AWSEG_ERANGE;
    if (heads > 65535 || batch > 65535) return AWSEG_ERANGE;
    if (((uintptr_t)q & 15) || ((uintptr_t)k & 15) || ((uintptr_t)v & 15) || ((uintptr_t)out & 15) || ((uintptr_t)workspace & 255)) return AWSEG_EALIGN;
    if (kv_pitch == 0) kv_pitch = heads * D;
    if (kv_pitch < heads * D || (kv_pitch & 3)) return AWSEG_EINVAL;
    if ((int64_t)(n_keys / TK) * IMG_TILE_HALFS * 2 > 0x7fffffff) return AWSEG_ERANGE;     // one (image, head)'s tiles behind a 32-bit descriptor
    int* exps = reinterpret_cast<int*>(workspace);
    const int64_t tab = ((int64_t)batch * heads * 2 * 4 + 255) / 256 * 256;
    _Float16* img = reinterpret_cast<_Float16*>(reinterpret_cast<char*>(workspace) + tab);
    hipLaunchKernelGGL(attn_kv_image_kernel, dim3((unsigned)heads, (unsigned)batch), dim3(AT), 0, awseg_s(stream), k, v, kv_pitch, n_keys, heads, img, exps);
    AWSEG_LAUNCH_CHECK();
    dim3 grid((unsigned)((n_queries + 127) / 128), (unsigned)heads, (unsigned)batch);
    hipLaunchKernelGGL(attention_d32_split_img_kernel, grid, dim3(AT), 0, awseg_s(stream), q, img, exps, out, n_queries, n_keys, heads,
                       scale * 1.4426950408889634f);
    AWSEG_LAUNCH_CHECK();
    return 0;
}

AWSEG_API int awseg_attention_d32_packed_kv(const float* q, const float* kv, float* out, int batch, int heads, int n_queries,
                                            int n_keys, float scale, int mode, awseg_stream_t stream)
{
    if (!kv) return AWSEG_EINVAL;
    const int c = heads * D;
    switch (mode) {
    case 0: return launch_attention(attention_d32_kernel, q, kv, kv + c, out, batch, heads, n_queries, n_keys, scale, stream, 2 * c);
    case 1: return launch_attention(attention_d32_split_kernel, q, kv, kv + c, out, batch, heads, n_queries, n_keys, scale, stream, 2 * c);
    case 2: return launch_attention(attention_d32_bf16_kernel, q, kv, kv + c, out, batch, heads, n_queries, n_keys, scale, stream, 2 * c);
    default: return AWSEG_EINVAL;
    }
}
