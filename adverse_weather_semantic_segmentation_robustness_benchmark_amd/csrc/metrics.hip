// metrics.hip — A11/A12/A13 (+ECE) kernels: ensemble combine -> /T -> argmax -> confusion.
//
// Reference arithmetic (PKG = adverse_weather_semantic_segmentation_robustness_benchmark):
//   combine      PKG/models/model.py:443-462
//   argmax       REF/scripts/evaluate.py:179
//   confusion    PKG/evaluation/metrics.py:54-71   (incl. the uint8 `targets*C` wrap)
//   ECE bins     PKG/evaluation/metrics.py:161-194
//
// All of these are HBM-bound scans: 153-229 B/px for the fused kernel, 2 B/px
// for confusion-from-predictions.  Layout is torch's NCHW: a pixel's C logits sit HW floats
// apart, so each lane owns 4 consecutive pixels and walks the channels with 16-byte loads
// (64 lanes x 16 B = one 1 KiB wave access per channel plane).  Confusion counts go to a
// per-block LDS histogram (ds_add_u32), are written once per block as uint32 partials and
// folded into the caller's int64 counters by a tiny second launch — no same-address global
// atomics on the streaming path, and the result is order-independent (integers).
#include "awseg_common.h"

namespace {

constexpr int kThreads = 256;
constexpr int kMaxBins = AWSEG_MAX_CLASSES * AWSEG_MAX_CLASSES;

__device__ __forceinline__ bool is_nan(float v) { return v != v; }

// torch argmax update rule: take v when v > best, or v is NaN and best is not.
__device__ __forceinline__ void amax_step(float v, int c, float& best, int& bi)
{
    if (!(v <= best) && !is_nan(best)) { best = v; bi = c; }
}

template <int LDT>
__device__ __forceinline__ void hist_add(uint32_t* hist, const void* label, int64_t li, int pred,
                                         int C, int ignore_index, int wrap, int64_t* oob)
{
    int64_t t = awseg_ld_label<LDT>(label, li);
    if (t == ignore_index) return;
    int64_t base = wrap ? (int64_t)(uint8_t)(t * C) : t * (int64_t)C;
    int64_t idx = base + pred;
    if (idx < 0 || idx >= (int64_t)C * C || pred < 0 || pred >= C) {
        atomicAdd((unsigned long long*)oob, 1ull);
        return;
    }
    atomicAdd(&hist[idx], 1u);
}

template <int PDT> __device__ __forceinline__ void st_pred(void* pred, int64_t i, int v)
{
    if (PDT == AWSEG_U8) ((uint8_t*)pred)[i] = (uint8_t)v;
    else ((int64_t*)pred)[i] = (int64_t)v;
}

// ---------------------------------------------------------------------------------------
// Fused kernel.  MODE: 0 weighted, 1 max-confidence, 2 mean, 3 single model (seg2 unused).
// VEC: pixels per lane per step (4 -> float4 path, 1 -> scalar fallback for odd shapes).
// grid = (blocks_per_image, B); block b of image y writes partial[(y*gridDim.x + b)][C*C].
// ---------------------------------------------------------------------------------------
template <int MODE, int VEC, int LDT, int PDT, int CT>
__global__ __launch_bounds__(kThreads)
void combine_argmax_confusion_kernel(const float* __restrict__ seg1, const float* __restrict__ seg2,
                                     int C, int64_t hw,
                                     const float* __restrict__ weights, const float* __restrict__ temperature,
                                     float* __restrict__ out_logits, void* __restrict__ pred,
                                     const void* __restrict__ label, int ignore_index, int wrap,
                                     uint32_t* __restrict__ partial, int64_t* __restrict__ oob)
{
    __shared__ uint32_t hist[kMaxBins];
    if (CT > 0) C = CT;                      // compile-time class count: the channel walk fully unrolls
    const int bins = C * C;
    const bool do_hist = (label != nullptr);
    if (do_hist) {
        for (int i = threadIdx.x; i < bins; i += kThreads) hist[i] = 0u;
        __syncthreads();
    }
    const int64_t img = blockIdx.y;
    const float* a = seg1 + img * C * hw;
    const float* d = (MODE == 3) ? nullptr : seg2 + img * C * hw;
    float* o = out_logits ? out_logits + img * C * hw : nullptr;

    float w0 = 0.f, w1 = 0.f, T = 1.f;
    const bool has_t = (temperature != nullptr);
    if (MODE == 0) { w0 = weights[0]; w1 = weights[1]; }
    if (has_t) T = temperature[0];

    const int64_t nvec = hw / VEC;
    for (int64_t v = (int64_t)blockIdx.x * kThreads + threadIdx.x; v < nvec; v += (int64_t)gridDim.x * kThreads) {
        const int64_t p = v * VEC;
        float best[VEC];
        int bi[VEC];
        float use[VEC];
        if (MODE == 1) {
            // max softmax probability of each member = 1 / sum(exp(x - max)); compare (model.py:449-453)
            float m1[VEC], m2[VEC], s1[VEC], s2[VEC];
#pragma unroll
            for (int k = 0; k < VEC; ++k) { m1[k] = -INFINITY; m2[k] = -INFINITY; s1[k] = 0.f; s2[k] = 0.f; }
            for (int c = 0; c < C; ++c)
#pragma unroll
                for (int k = 0; k < VEC; ++k) {
                    m1[k] = fmaxf(m1[k], a[(int64_t)c * hw + p + k]);
                    m2[k] = fmaxf(m2[k], d[(int64_t)c * hw + p + k]);
                }
            for (int c = 0; c < C; ++c)
#pragma unroll
                for (int k = 0; k < VEC; ++k) {
                    s1[k] += expf(a[(int64_t)c * hw + p + k] - m1[k]);
                    s2[k] += expf(d[(int64_t)c * hw + p + k] - m2[k]);
                }
#pragma unroll
            for (int k = 0; k < VEC; ++k) use[k] = (1.0f / s1[k] > 1.0f / s2[k]) ? 1.f : 0.f;
        }
        constexpr int kUnroll = CT > 0 ? CT : 4;
#pragma unroll kUnroll
        for (int c = 0; c < C; ++c) {
            float x[VEC], y[VEC], r[VEC];
            if constexpr (VEC == 4) {
                float4 xv = *reinterpret_cast<const float4*>(a + (int64_t)c * hw + p);
                x[0] = xv.x; x[1] = xv.y; x[2] = xv.z; x[3] = xv.w;
                if (MODE != 3) {
                    float4 yv = *reinterpret_cast<const float4*>(d + (int64_t)c * hw + p);
                    y[0] = yv.x; y[1] = yv.y; y[2] = yv.z; y[3] = yv.w;
                }
            } else {
                x[0] = a[(int64_t)c * hw + p];
                if (MODE != 3) y[0] = d[(int64_t)c * hw + p];
            }
#pragma unroll
            for (int k = 0; k < VEC; ++k) {
                // four separately rounded float32 operations (built with -ffp-contract=off)
                if (MODE == 0) { float u = w0 * x[k]; float t = w1 * y[k]; r[k] = u + t; }
                else if (MODE == 1) { float u = use[k] * x[k]; float t = (1.f - use[k]) * y[k]; r[k] = u + t; }
                else if (MODE == 2) { float u = x[k] + y[k]; r[k] = u / 2.f; }
                else r[k] = x[k];
                if (MODE != 3 && has_t) r[k] = r[k] / T;
                if (c == 0) { best[k] = r[k]; bi[k] = 0; }
                else amax_step(r[k], c, best[k], bi[k]);
            }
            if (o) {
                if constexpr (VEC == 4) *reinterpret_cast<float4*>(o + (int64_t)c * hw + p) = make_float4(r[0], r[1], r[2], r[3]);
                else o[(int64_t)c * hw + p] = r[0];
            }
        }
        if (pred) {
            if constexpr (VEC == 4 && PDT == AWSEG_U8) {
                uint32_t pk = (uint32_t)bi[0] | ((uint32_t)bi[1] << 8) | ((uint32_t)bi[2] << 16) | ((uint32_t)bi[3] << 24);
                *reinterpret_cast<uint32_t*>((uint8_t*)pred + img * hw + p) = pk;
            } else {
#pragma unroll
                for (int k = 0; k < VEC; ++k) st_pred<PDT>(pred, img * hw + p + k, bi[k]);
            }
        }
        if (do_hist) {
#pragma unroll
            for (int k = 0; k < VEC; ++k)
                hist_add<LDT>(hist, label, img * hw + p + k, bi[k], C, ignore_index, wrap, oob);
        }
    }
    if (do_hist) {
        __syncthreads();
        uint32_t* dst = partial + ((int64_t)blockIdx.y * gridDim.x + blockIdx.x) * bins;
        for (int i = threadIdx.x; i < bins; i += kThreads) dst[i] = hist[i];
    }
}

// Fold the per-block uint32 partial histograms of one image into slot 0 and slot 1+cond[img].
// grid = (images, ceil(bins/64)), 1024 threads: a block owns 64 bins and its 16 waves each sum a
// sixteenth of the partials (independent, unrolled loads), then one LDS step combines them — the
// dependent-load chain is blocks_per_image/16 long and the bins run in parallel across blocks.
constexpr int kFoldSlices = 16;
__global__ __launch_bounds__(kFoldSlices * 64)
void fold_partials_kernel(const uint32_t* __restrict__ partial, int blocks_per_image, int bins,
                          const int32_t* __restrict__ cond, int n_slots, int64_t* __restrict__ counts)
{
    __shared__ unsigned long long s_sum[kFoldSlices][64];
    const int img = blockIdx.x;
    const int kl = threadIdx.x & 63, slice = threadIdx.x >> 6;
    const int k = blockIdx.y * 64 + kl;
    const uint32_t* src = partial + (int64_t)img * blocks_per_image * bins;
    unsigned long long s = 0;
    if (k < bins) {
#pragma unroll 8
        for (int b = slice; b < blocks_per_image; b += kFoldSlices) s += src[(int64_t)b * bins + k];
    }
    s_sum[slice][kl] = s;
    __syncthreads();
    if (slice == 0 && k < bins) {
        s = 0;
#pragma unroll
        for (int j = 0; j < kFoldSlices; ++j) s += s_sum[j][kl];
        if (s) {
            int slot = -1;
            if (cond) { int c = cond[img]; if (c >= 0 && c + 1 < n_slots) slot = c + 1; }
            atomicAdd((unsigned long long*)&counts[k], s);
            if (slot > 0) atomicAdd((unsigned long long*)&counts[(int64_t)slot * bins + k], s);
        }
    }
}

// ---------------------------------------------------------------------------------------
// confusion from predictions (2 B/px at u8/u8): 16 pixels per lane, runs of equal index are
// merged in registers before touching LDS (segmentation maps are piecewise constant, so the
// common case is one ds_add per 16 pixels; random maps degrade to one per pixel).
// ---------------------------------------------------------------------------------------
template <int PDT, int LDT>
__global__ __launch_bounds__(kThreads)
void confusion_kernel(const void* __restrict__ pred, const void* __restrict__ label, int64_t n, int C,
                      int ignore_index, int wrap, uint32_t* __restrict__ partial, int64_t* __restrict__ oob)
{
    __shared__ uint32_t hist[kMaxBins];
    const int bins = C * C;
    for (int i = threadIdx.x; i < bins; i += kThreads) hist[i] = 0u;
    __syncthreads();
    constexpr int PER = 16;
    const int64_t nchunk = (n + PER - 1) / PER;
    for (int64_t ch = (int64_t)blockIdx.x * kThreads + threadIdx.x; ch < nchunk; ch += (int64_t)gridDim.x * kThreads) {
        const int64_t base = ch * PER;
        int64_t pv[PER], lv[PER];
        if (base + PER <= n) {
            if (PDT == AWSEG_U8) {
                uint4 q = *reinterpret_cast<const uint4*>((const uint8_t*)pred + base);
                uint32_t w[4] = { q.x, q.y, q.z, q.w };
#pragma unroll
                for (int k = 0; k < PER; ++k) pv[k] = (w[k >> 2] >> ((k & 3) * 8)) & 0xFF;
            } else {
#pragma unroll
                for (int k = 0; k < PER; ++k) pv[k] = ((const int64_t*)pred)[base + k];
            }
            if (LDT == AWSEG_U8) {
                uint4 q = *reinterpret_cast<const uint4*>((const uint8_t*)label + base);
                uint32_t w[4] = { q.x, q.y, q.z, q.w };
#pragma unroll
                for (int k = 0; k < PER; ++k) lv[k] = (w[k >> 2] >> ((k & 3) * 8)) & 0xFF;
            } else {
#pragma unroll
                for (int k = 0; k < PER; ++k) lv[k] = ((const int64_t*)label)[base + k];
            }
        } else {
#pragma unroll
            for (int k = 0; k < PER; ++k) {
                bool in = base + k < n;
                pv[k] = in ? (PDT == AWSEG_U8 ? (int64_t)((const uint8_t*)pred)[base + k] : ((const int64_t*)pred)[base + k]) : 0;
                lv[k] = in ? awseg_ld_label<LDT>(label, base + k) : (int64_t)ignore_index;
            }
        }
        int64_t run_idx = -1; uint32_t run = 0;
#pragma unroll
        for (int k = 0; k < PER; ++k) {
            if (lv[k] == ignore_index) continue;
            int64_t b = wrap ? (int64_t)(uint8_t)(lv[k] * C) : lv[k] * (int64_t)C;
            int64_t idx = b + pv[k];
            if (idx < 0 || idx >= bins || pv[k] < 0 || pv[k] >= C) { atomicAdd((unsigned long long*)oob, 1ull); continue; }
            if (idx == run_idx) { ++run; }
            else { if (run) atomicAdd(&hist[run_idx], run); run_idx = idx; run = 1; }
        }
        if (run) atomicAdd(&hist[run_idx], run);
    }
    __syncthreads();
    uint32_t* dst = partial + (int64_t)blockIdx.x * bins;
    for (int i = threadIdx.x; i < bins; i += kThreads) dst[i] = hist[i];
}

// ---------------------------------------------------------------------------------------
// ECE accumulators: per block LDS bins {count u32, correct u32, sum of confidences}.
// The confidence sum is kept in FIXED POINT, units of 2^-30, in 64-bit integers: a float32 confidence in [2^-7, 1] is a
// multiple of 2^-30, so conf * 2^30 is an exact integer and integer sums do not depend on the order of the atomics, the
// block schedule or the number of ranks the counters are later all-reduced over (results identical at any
// GPU count).  2^33 pixels fit.  Block partials: [nblk][n_bins] x {uint32 cnt, uint32 correct, uint64 sum_conf_q30}.
// ---------------------------------------------------------------------------------------
struct ece_cell { uint32_t cnt; uint32_t correct; unsigned long long sum_conf; };
constexpr float kConfQ = 1073741824.0f;                          // 2^30
__device__ __forceinline__ unsigned long long conf_q30(float conf) { return (unsigned long long)(conf * kConfQ); }

// Bin of a confidence on the reference's float32 linspace, bins (lo, hi] (metrics.py:179-188): the uniform-grid guess is
// checked against the staged edges with the reference's own comparisons and moved by one where rounding put it next door;
// anything else (edges that are not a uniform grid) falls back to the linear scan.  -1 = in no bin.
__device__ __forceinline__ int ece_find_bin(float conf, const float* s_edges, int n_bins)
{
    int b = (int)ceilf(conf * (float)n_bins) - 1;
    b = b < 0 ? 0 : (b > n_bins - 1 ? n_bins - 1 : b);
    if (!(conf > s_edges[b])) b = b > 0 ? b - 1 : 0;
    else if (!(conf <= s_edges[b + 1])) b = b < n_bins - 1 ? b + 1 : b;
    if (conf > s_edges[b] && conf <= s_edges[b + 1]) return b;
    for (int k = 0; k < n_bins; ++k)
        if (conf > s_edges[k] && conf <= s_edges[k + 1]) return k;
    return -1;
}

// C = 19, hw % 4 == 0: four pixels per lane, the 19 x 4 logits of a lane live in registers (one 16-byte load per class
// plane instead of two 4-byte passes), fast exponentials.
template <int LDT>
__global__ __launch_bounds__(kThreads)
void ece19_kernel(const float* __restrict__ logits, int64_t hw, const void* __restrict__ label,
                  const float* __restrict__ edges, int n_bins, ece_cell* __restrict__ partial)
{
    constexpr int C = 19;
    __shared__ uint32_t s_cnt[64], s_cor[64];
    __shared__ unsigned long long s_sum[64];
    __shared__ float s_edges[65];
    for (int i = threadIdx.x; i < n_bins; i += kThreads) { s_cnt[i] = 0; s_cor[i] = 0; s_sum[i] = 0ull; }
    for (int i = threadIdx.x; i <= n_bins; i += kThreads) s_edges[i] = edges[i];
    __syncthreads();
    const int64_t img = blockIdx.y;
    const float* x = logits + img * C * hw;
    const int64_t nvec = hw / 4;
    for (int64_t v = (int64_t)blockIdx.x * kThreads + threadIdx.x; v < nvec; v += (int64_t)gridDim.x * kThreads) {
        const int64_t p = v * 4;
        float4 xv[C];
#pragma unroll
        for (int c = 0; c < C; ++c) xv[c] = *reinterpret_cast<const float4*>(x + (int64_t)c * hw + p);
        int64_t t[4];
#pragma unroll
        for (int k = 0; k < 4; ++k) t[k] = awseg_ld_label<LDT>(label, img * hw + p + k);
#pragma unroll
        for (int k = 0; k < 4; ++k) {
            if (t[k] == 255) continue;                    // metrics.py:170 hard-codes 255
            auto at = [&](int c) { return k == 0 ? xv[c].x : (k == 1 ? xv[c].y : (k == 2 ? xv[c].z : xv[c].w)); };
            float m = at(0); int bi = 0;
#pragma unroll
            for (int c = 1; c < C; ++c) { const float q = at(c); if (q > m) { m = q; bi = c; } }
            float sum = 0.f;
#pragma unroll
            for (int c = 0; c < C; ++c) sum += __expf(at(c) - m);
            const float conf = 1.0f / sum;
            const int b = ece_find_bin(conf, s_edges, n_bins);
            if (b >= 0) {
                atomicAdd(&s_cnt[b], 1u);
                if (bi == (int)t[k]) atomicAdd(&s_cor[b], 1u);
                atomicAdd(&s_sum[b], conf_q30(conf));
            }
        }
    }
    __syncthreads();
    ece_cell* dst = partial + ((int64_t)blockIdx.y * gridDim.x + blockIdx.x) * n_bins;
    for (int i = threadIdx.x; i < n_bins; i += kThreads) { dst[i].cnt = s_cnt[i]; dst[i].correct = s_cor[i]; dst[i].sum_conf = s_sum[i]; }
}

template <int LDT>
__global__ __launch_bounds__(kThreads)
void ece_kernel(const float* __restrict__ logits, int C, int64_t hw, const void* __restrict__ label,
                const float* __restrict__ edges, int n_bins, ece_cell* __restrict__ partial)
{
    __shared__ uint32_t s_cnt[64], s_cor[64];
    __shared__ unsigned long long s_sum[64];
    __shared__ float s_edges[65];
    for (int i = threadIdx.x; i < n_bins; i += kThreads) { s_cnt[i] = 0; s_cor[i] = 0; s_sum[i] = 0ull; }
    for (int i = threadIdx.x; i <= n_bins; i += kThreads) s_edges[i] = edges[i];
    __syncthreads();
    const int64_t img = blockIdx.y;
    const float* x = logits + img * C * hw;
    for (int64_t p = (int64_t)blockIdx.x * kThreads + threadIdx.x; p < hw; p += (int64_t)gridDim.x * kThreads) {
        int64_t t = awseg_ld_label<LDT>(label, img * hw + p);
        if (t == 255) continue;                       // metrics.py:170 hard-codes 255
        float m = x[p]; int bi = 0;
        for (int c = 1; c < C; ++c) { float v = x[(int64_t)c * hw + p]; if (v > m) { m = v; bi = c; } }
        float s = 0.f;
        // the SAME exponential and summation order as ece19_kernel: which of the two kernels runs depends on hw % 4 and on the
        // pointer's alignment, and a view of the same logits must land in the same bins
        for (int c = 0; c < C; ++c) s += __expf(x[(int64_t)c * hw + p] - m);
        float conf = 1.0f / s;
        // bins are (lo, hi] on a float32 linspace (metrics.py:179-188); linear scan keeps the
        // reference's comparison semantics exactly.
        const int k = ece_find_bin(conf, s_edges, n_bins);
        if (k >= 0) {
            atomicAdd(&s_cnt[k], 1u);
            if (bi == t) atomicAdd(&s_cor[k], 1u);
            atomicAdd(&s_sum[k], conf_q30(conf));
        }
    }
    __syncthreads();
    ece_cell* dst = partial + ((int64_t)blockIdx.y * gridDim.x + blockIdx.x) * n_bins;
    for (int i = threadIdx.x; i < n_bins; i += kThreads) { dst[i].cnt = s_cnt[i]; dst[i].correct = s_cor[i]; dst[i].sum_conf = s_sum[i]; }
}

// bins out: per slot, n_bins x {int64 count, int64 sum_conf in units of 2^-30, int64 sum_correct}
struct ece_out { long long cnt; long long sum_conf; long long correct; };

__global__ void ece_fold_kernel(const ece_cell* __restrict__ partial, int blocks_per_image, int n_bins,
                                const int32_t* __restrict__ cond, int n_slots, ece_out* __restrict__ bins)
{
    const int img = blockIdx.x;
    const int k = threadIdx.x;
    if (k >= n_bins) return;
    const ece_cell* src = partial + (int64_t)img * blocks_per_image * n_bins;
    unsigned long long cnt = 0, cor = 0, sum = 0;
    for (int b = 0; b < blocks_per_image; ++b) { cnt += src[b * n_bins + k].cnt; cor += src[b * n_bins + k].correct; sum += src[b * n_bins + k].sum_conf; }
    int slot = -1;
    if (cond) { int c = cond[img]; if (c >= 0 && c + 1 < n_slots) slot = c + 1; }
    if (cnt) {
        atomicAdd((unsigned long long*)&bins[k].cnt, cnt);
        atomicAdd((unsigned long long*)&bins[k].correct, cor);
        atomicAdd((unsigned long long*)&bins[k].sum_conf, sum);
        if (slot > 0) {
            ece_out* b = bins + (int64_t)slot * n_bins;
            atomicAdd((unsigned long long*)&b[k].cnt, cnt);
            atomicAdd((unsigned long long*)&b[k].correct, cor);
            atomicAdd((unsigned long long*)&b[k].sum_conf, sum);
        }
    }
}

// ---------------------------------------------------------------------------------------
// Ensemble calibration + disagreement statistics in ONE pass over the two member logit maps
// (next #1: REF/scripts/evaluate.py:230-255).  Per pixel with label != 255:
//   ECE        r = combine(s1, s2)/T (same four roundings as the combine kernel), conf = max softmax(r),
//              pred = argmax r, bin (lo, hi] -> {count, correct, sum conf}      (metrics.py:161-194)
//   AUROC      p_i = softmax(s_i), m = (p1 + p2)/2, score = H(m) - (H(p1) + H(p2))/2 with the reference's
//              +1e-8 inside the logs (metrics.py:353-367), error = argmax(m) != label (:414-419);
//              the score is histogrammed by error flag (2 x n_hist int64 bins, global atomics) — AUROC is
//              then a rank statistic of the two histograms, additive over batches and ranks.
// C = 19 only (2 x 19 x 4 logits live in registers); 4 pixels per lane.
// ---------------------------------------------------------------------------------------
// CONF = true: the same pass ALSO counts the 19 x 19 confusion matrix of argmax(r) against the labels, with the combine kernel's
// rules (torch's argmax update, ignore_index, the reference's uint8 index wrap, out-of-range count) into per-block partials that
// fold_partials_kernel folds — the separate awseg_combine_argmax_confusion pass over the two logit maps (2.55 GB per batch of 8
// at 1024 x 2048) is not needed when neither the ensemble logits nor the prediction map are asked for.
// TH threads x PX pixels per lane: the block's 64 KB score histogram allows two blocks per CU, so 256 threads are two waves per SIMD
// whatever the register count; 512 threads with 2 pixels per lane (2 x 19 x 2 logits: < 128 registers) are four.
template <int MODE, int LDT, bool CONF, int TH, int PX>
__global__ __launch_bounds__(TH, TH / 128)   // waves per SIMD of two resident blocks
void ensemble_stats_kernel(const float* __restrict__ seg1, const float* __restrict__ seg2, int64_t hw,
                           const float* __restrict__ weights, const float* __restrict__ temperature,
                           const void* __restrict__ label, const float* __restrict__ edges, int n_bins,
                           ece_cell* __restrict__ partial, unsigned long long* __restrict__ hist, int n_hist,
                           float h_lo, float h_scale, int ignore_index, int wrap, uint32_t* __restrict__ conf_partial,
                           int64_t* __restrict__ oob)
{
    constexpr int C = 19;
    __shared__ uint32_t s_conf[CONF ? C * C : 1];
    if (CONF) { for (int i = threadIdx.x; i < C * C; i += TH) s_conf[i] = 0u; }
    __shared__ uint32_t s_cnt[64], s_cor[64];
    __shared__ unsigned long long s_sum[64];
    __shared__ float s_edges[65];
    // Score histogram of this block in LDS (2 x n_hist uint32, n_hist <= kHistMax): pixels of one frame
    // have similar scores, so counting straight into global memory is same-address atomic traffic
    // (measured 45 ms per batch); the block flushes only its non-zero bins at the end.
    extern __shared__ uint32_t s_hist[];
    for (int i = threadIdx.x; i < 2 * n_hist; i += TH) s_hist[i] = 0u;
    for (int i = threadIdx.x; i < n_bins; i += TH) { s_cnt[i] = 0; s_cor[i] = 0; s_sum[i] = 0ull; }
    for (int i = threadIdx.x; i <= n_bins; i += TH) s_edges[i] = edges[i];
    __syncthreads();
    const int64_t img = blockIdx.y;
    const float* a = seg1 + img * C * hw;
    const float* d = seg2 + img * C * hw;
    float w0 = 0.f, w1 = 0.f, T = 1.f;
    const bool has_t = (temperature != nullptr);
    if (MODE == 0) { w0 = weights[0]; w1 = weights[1]; }
    if (has_t) T = temperature[0];
    const int64_t nvec = hw / PX;
    typedef float lvec __attribute__((ext_vector_type(PX)));
    for (int64_t v = (int64_t)blockIdx.x * TH + threadIdx.x; v < nvec; v += (int64_t)gridDim.x * TH) {
        const int64_t p = v * PX;
        float x[C][PX], y[C][PX];
#pragma unroll
        for (int c = 0; c < C; ++c) {
            const lvec xv = *reinterpret_cast<const lvec*>(a + (int64_t)c * hw + p);
            const lvec yv = *reinterpret_cast<const lvec*>(d + (int64_t)c * hw + p);
#pragma unroll
            for (int k = 0; k < PX; ++k) { x[c][k] = xv[k]; y[c][k] = yv[k]; }
        }
#pragma unroll
        for (int k = 0; k < PX; ++k) {
            const int64_t t = awseg_ld_label<LDT>(label, img * hw + p + k);
            if (!CONF && t == 255) continue;                      // metrics.py:170, :426
            // ensemble logits r, their max / argmax / sum-exp for the calibration part
            float rmax = -INFINITY, rsum = 0.f; int rarg = 0;
            float r[C];
            float m1 = x[0][k], m2 = y[0][k];
#pragma unroll
            for (int c = 0; c < C; ++c) {
                float rv;
                if (MODE == 0) { float u = w0 * x[c][k]; float q = w1 * y[c][k]; rv = u + q; }
                else { float u = x[c][k] + y[c][k]; rv = u / 2.f; }
                if (has_t) rv = rv / T;
                r[c] = rv;
                if (CONF) { if (c == 0) { rmax = rv; rarg = 0; } else amax_step(rv, c, rmax, rarg); }   // torch's argmax rule, as the combine kernel
                else if (c == 0 || rv > rmax) { rmax = rv; rarg = c; }
                m1 = fmaxf(m1, x[c][k]); m2 = fmaxf(m2, y[c][k]);
            }
            if (CONF) {
                hist_add<LDT>(s_conf, label, img * hw + p + k, rarg, C, ignore_index, wrap, oob);
                if (t == 255) continue;                           // the calibration / disagreement statistics skip 255 (metrics.py:170, :426)
            }
#pragma unroll
            for (int c = 0; c < C; ++c) rsum += __expf(r[c] - rmax);
            const float conf = 1.0f / rsum;
            const int eb = ece_find_bin(conf, s_edges, n_bins);
            if (eb >= 0) {
                atomicAdd(&s_cnt[eb], 1u);
                if (rarg == (int)t) atomicAdd(&s_cor[eb], 1u);
                atomicAdd(&s_sum[eb], conf_q30(conf));
            }
            // member softmaxes: the raw logits are dead from here on, their registers take the exponentials.
            // Member entropies without logarithms: p = e / z with e = exp(x - m), so log p = (x - m) - log z and
            //   H(p) = -sum p log p = log z - (sum e (x - m)) / z;
            // the reference's -sum p log(p + 1e-8) differs from it by sum p log(1 + 1e-8/p) <= 19e-8 (metrics.py:353-367),
            // below the float32 rounding of the sum itself.  The mixture entropy keeps its per-class logarithm.
            float z1 = 0.f, z2 = 0.f, t1 = 0.f, t2 = 0.f;
#pragma unroll
            for (int c = 0; c < C; ++c) {
                const float d1 = x[c][k] - m1, d2 = y[c][k] - m2;
                x[c][k] = __expf(d1); y[c][k] = __expf(d2);
                z1 += x[c][k]; z2 += y[c][k];
                t1 = fmaf(x[c][k], d1, t1); t2 = fmaf(y[c][k], d2, t2);
            }
            const float i1 = 1.0f / z1, i2 = 1.0f / z2;
            const float h1 = __logf(z1) - t1 * i1, h2 = __logf(z2) - t2 * i2;
            // disagreement (mutual information) and the error flag of the mean-probability prediction
            float hm = 0.f, mbest = -1.f; int marg = 0;
#pragma unroll
            for (int c = 0; c < C; ++c) {
                const float p1 = x[c][k] * i1, p2 = y[c][k] * i2;
                const float mp = (p1 + p2) * 0.5f;
                hm -= mp * __logf(mp + 1e-8f);
                if (mp > mbest) { mbest = mp; marg = c; }
            }
            const float score = hm - (h1 + h2) * 0.5f;
            int hb = (int)((score - h_lo) * h_scale);
            hb = hb < 0 ? 0 : (hb >= n_hist ? n_hist - 1 : hb);
            atomicAdd(&s_hist[(marg != (int)t ? n_hist : 0) + hb], 1u);
        }
    }
    __syncthreads();
    ece_cell* dst = partial + ((int64_t)blockIdx.y * gridDim.x + blockIdx.x) * n_bins;
    for (int i = threadIdx.x; i < n_bins; i += TH) { dst[i].cnt = s_cnt[i]; dst[i].correct = s_cor[i]; dst[i].sum_conf = s_sum[i]; }
    for (int i = threadIdx.x; i < 2 * n_hist; i += TH) {
        const uint32_t v = s_hist[i];
        if (v) atomicAdd(&hist[i], (unsigned long long)v);
    }
    if (CONF) {
        uint32_t* cd = conf_partial + ((int64_t)blockIdx.y * gridDim.x + blockIdx.x) * (C * C);
        for (int i = threadIdx.x; i < C * C; i += TH) cd[i] = s_conf[i];
    }
}

constexpr int kHistMax = 8192;

int blocks_per_image(int64_t hw, int64_t batch, int vec)
{
    // enough blocks to fill 256 CUs x 8 resident blocks across the whole batch, grid-stride beyond
    int64_t want = (hw / vec + kThreads - 1) / kThreads;
    int64_t cap = (AWSEG_CUS * 8 + batch - 1) / batch;
    if (cap < 1) cap = 1;
    if (want > cap) want = cap;
    if (want < 1) want = 1;
    return (int)want;
}

template <int MODE, int VEC>
int launch_fused(const float* seg1, const float* seg2, int64_t batch, int C, int64_t hw,
                 const float* weights, const float* temperature, float* out_logits, void* pred, int pdt,
                 const void* label, int ldt, int ignore_index, int wrap, uint32_t* partial, int64_t* oob,
                 int bpi, hipStream_t s)
{
    dim3 grid(bpi, (unsigned)batch), block(kThreads);
#define AWSEG_FUSED(L, P, CTV) \
    hipLaunchKernelGGL((combine_argmax_confusion_kernel<MODE, VEC, L, P, CTV>), grid, block, 0, s, seg1, seg2, C, hw, \
                       weights, temperature, out_logits, pred, label, ignore_index, wrap, partial, oob)
    if (C == 19 && VEC == 4 && MODE != 1) {          // Cityscapes: fully unrolled channel walk
        if (ldt == AWSEG_U8 && pdt == AWSEG_U8) AWSEG_FUSED(AWSEG_U8, AWSEG_U8, 19);
        else if (ldt == AWSEG_U8) AWSEG_FUSED(AWSEG_U8, AWSEG_I64, 19);
        else if (pdt == AWSEG_U8) AWSEG_FUSED(AWSEG_I64, AWSEG_U8, 19);
        else AWSEG_FUSED(AWSEG_I64, AWSEG_I64, 19);
        return 0;
    }
    if (ldt == AWSEG_U8 && pdt == AWSEG_U8) AWSEG_FUSED(AWSEG_U8, AWSEG_U8, 0);
    else if (ldt == AWSEG_U8) AWSEG_FUSED(AWSEG_U8, AWSEG_I64, 0);
    else if (pdt == AWSEG_U8) AWSEG_FUSED(AWSEG_I64, AWSEG_U8, 0);
    else AWSEG_FUSED(AWSEG_I64, AWSEG_I64, 0);
#undef AWSEG_FUSED
    return 0;
}

bool aligned16(const void* p) { return ((uintptr_t)p & 15) == 0; }

}  // namespace

AWSEG_API int64_t awseg_metrics_workspace(int64_t batch, int num_classes, int64_t hw)
{
    // uint32 partials: one C*C histogram per block; also covers the ECE partials (24 B x 64 bins)
    int64_t bpi = blocks_per_image(hw, batch < 1 ? 1 : batch, 1);
    // + the ECE partials BEHIND the histogram partials when one launch produces both (awseg_combine_confusion_stats)
    int64_t per_block = (int64_t)num_classes * num_classes * 4 + 64 * 24;
    return bpi * (batch < 1 ? 1 : batch) * per_block;
}

static int fused_impl(int mode, const float* seg1, const float* seg2, int64_t batch, int C, int64_t hw,
                      const float* weights, const float* temperature, float* out_logits, void* pred, int pdt,
                      const void* label, int ldt, int ignore_index, int wrap, const int32_t* cond,
                      int64_t* counts, int n_slots, int64_t* oob, void* workspace, hipStream_t s)
{
    if (!seg1 || batch < 1 || hw < 1 || C < 1 || C > AWSEG_MAX_CLASSES) return AWSEG_EINVAL;
    if (mode != 3 && !seg2) return AWSEG_EINVAL;
    if (mode == 0 && !weights) return AWSEG_EINVAL;
    if (batch > 65535) return AWSEG_ERANGE;
    if ((pdt != AWSEG_U8 && pdt != AWSEG_I64) || (ldt != AWSEG_U8 && ldt != AWSEG_I64)) return AWSEG_EINVAL;
    if (label && (!counts || !oob || !workspace || n_slots < 1)) return AWSEG_EINVAL;
    const bool vec4 = (hw % 4 == 0) && aligned16(seg1) && (mode == 3 || aligned16(seg2)) &&
                      (!out_logits || aligned16(out_logits)) && (!pred || pdt != AWSEG_U8 || ((uintptr_t)pred & 3) == 0);
    const int bpi = blocks_per_image(hw, batch, 1);   // same count the workspace query assumed
    uint32_t* partial = (uint32_t*)workspace;
#define AWSEG_MODE(M)                                                                                           \
    (vec4 ? launch_fused<M, 4>(seg1, seg2, batch, C, hw, weights, temperature, out_logits, pred, pdt, label,    \
                               ldt, ignore_index, wrap, partial, oob, bpi, s)                                   \
          : launch_fused<M, 1>(seg1, seg2, batch, C, hw, weights, temperature, out_logits, pred, pdt, label,    \
                               ldt, ignore_index, wrap, partial, oob, bpi, s))
    switch (mode) {
        case 0: AWSEG_MODE(0); break;
        case 1: AWSEG_MODE(1); break;
        case 2: AWSEG_MODE(2); break;
        case 3: AWSEG_MODE(3); break;
        default: return AWSEG_EINVAL;
    }
#undef AWSEG_MODE
    AWSEG_LAUNCH_CHECK();
    if (label) {
        hipLaunchKernelGGL(fold_partials_kernel, dim3((unsigned)batch, (C * C + 63) / 64), dim3(kFoldSlices * 64), 0, s, partial, bpi,
                           C * C, cond, n_slots, counts);
        AWSEG_LAUNCH_CHECK();
    }
    return 0;
}

AWSEG_API int awseg_combine_argmax_confusion(const float* seg1, const float* seg2, int64_t batch, int num_classes,
                                             int64_t hw, int mode, const float* weights, const float* temperature,
                                             float* out_logits, void* pred, int pred_dtype, const void* label,
                                             int label_dtype, int ignore_index, int label_wrap_u8,
                                             const int32_t* cond, int64_t* counts, int n_slots, int64_t* oob,
                                             void* workspace, awseg_stream_t stream)
{
    if (mode < 0 || mode > 2) return AWSEG_EINVAL;
    return fused_impl(mode, seg1, seg2, batch, num_classes, hw, weights, temperature, out_logits, pred, pred_dtype,
                      label, label_dtype, ignore_index, label_wrap_u8, cond, counts, n_slots, oob, workspace,
                      awseg_s(stream));
}

AWSEG_API int awseg_argmax_confusion(const float* logits, int64_t batch, int num_classes, int64_t hw, void* pred,
                                     int pred_dtype, const void* label, int label_dtype, int ignore_index,
                                     int label_wrap_u8, const int32_t* cond, int64_t* counts, int n_slots,
                                     int64_t* oob, void* workspace, awseg_stream_t stream)
{
    return fused_impl(3, logits, nullptr, batch, num_classes, hw, nullptr, nullptr, nullptr, pred, pred_dtype, label,
                      label_dtype, ignore_index, label_wrap_u8, cond, counts, n_slots, oob, workspace, awseg_s(stream));
}

AWSEG_API int awseg_argmax(const float* logits, int64_t batch, int num_classes, int64_t hw, void* pred, int pred_dtype,
                           awseg_stream_t stream)
{
    if (!pred) return AWSEG_EINVAL;
    return fused_impl(3, logits, nullptr, batch, num_classes, hw, nullptr, nullptr, nullptr, pred, pred_dtype, nullptr,
                      AWSEG_U8, 255, 0, nullptr, nullptr, 0, nullptr, nullptr, awseg_s(stream));
}

AWSEG_API int awseg_confusion_accumulate(const void* pred, int pred_dtype, const void* label, int label_dtype,
                                         int64_t n, int num_classes, int ignore_index, int label_wrap_u8,
                                         int64_t* counts, int64_t* oob, void* workspace, awseg_stream_t stream)
{
    if (n == 0) return 0;                            // empty input (all pointers may be NULL): nothing to count
    if (!pred || !label || !counts || !oob || !workspace || n < 0) return AWSEG_EINVAL;
    if (num_classes < 1 || num_classes > AWSEG_MAX_CLASSES) return AWSEG_EINVAL;
    hipStream_t s = awseg_s(stream);
    // never more blocks than the workspace query assumed (blocks_per_image(n, 1, 1)); one block
    // covers 4096 pixels per sweep, 1024 blocks keep every CU busy
    int nblk = blocks_per_image((n + 15) / 16, 1, 1);
    if (nblk > 1024) nblk = 1024;
    uint32_t* partial = (uint32_t*)workspace;
    const bool al = aligned16(pred) && aligned16(label);
    (void)al;  // unaligned byte maps still work: the uint4 path requires 16-B alignment
    if ((pred_dtype == AWSEG_U8 && !aligned16(pred)) || (label_dtype == AWSEG_U8 && !aligned16(label))) return AWSEG_EALIGN;
    dim3 grid(nblk), block(kThreads);
#define AWSEG_CONF(P, L) \
    hipLaunchKernelGGL((confusion_kernel<P, L>), grid, block, 0, s, pred, label, n, num_classes, ignore_index, label_wrap_u8, partial, oob)
    if (pred_dtype == AWSEG_U8 && label_dtype == AWSEG_U8) AWSEG_CONF(AWSEG_U8, AWSEG_U8);
    else if (pred_dtype == AWSEG_U8 && label_dtype == AWSEG_I64) AWSEG_CONF(AWSEG_U8, AWSEG_I64);
    else if (pred_dtype == AWSEG_I64 && label_dtype == AWSEG_U8) AWSEG_CONF(AWSEG_I64, AWSEG_U8);
    else if (pred_dtype == AWSEG_I64 && label_dtype == AWSEG_I64) AWSEG_CONF(AWSEG_I64, AWSEG_I64);
    else return AWSEG_EINVAL;
#undef AWSEG_CONF
    AWSEG_LAUNCH_CHECK();
    hipLaunchKernelGGL(fold_partials_kernel, dim3(1, (num_classes * num_classes + 63) / 64), dim3(kFoldSlices * 64), 0, s, partial, nblk,
                       num_classes * num_classes, (const int32_t*)nullptr, 1, counts);
    AWSEG_LAUNCH_CHECK();
    return 0;
}

AWSEG_API int awseg_ece_accumulate(const float* logits, int64_t batch, int num_classes, int64_t hw, const void* label,
                                   int label_dtype, const int32_t* cond, const float* edges, int n_bins, void* bins,
                                   int n_slots, void* workspace, awseg_stream_t stream)
{
    if (!logits || !label || !edges || !bins || !workspace) return AWSEG_EINVAL;
    if (n_bins < 1 || n_bins > 64 || n_slots < 1 || batch < 1 || batch > 65535 || hw < 1) return AWSEG_EINVAL;
    if (num_classes < 1 || num_classes > AWSEG_MAX_CLASSES) return AWSEG_EINVAL;
    hipStream_t s = awseg_s(stream);
    const bool vec19 = (num_classes == 19) && !(hw & 3) && !((uintptr_t)logits & 15);
    const int bpi = blocks_per_image(hw, batch, vec19 ? 4 : 1);
    dim3 grid(bpi, (unsigned)batch), block(kThreads);
    if (label_dtype != AWSEG_U8 && label_dtype != AWSEG_I64) return AWSEG_EINVAL;
    if (vec19 && label_dtype == AWSEG_U8)
        hipLaunchKernelGGL((ece19_kernel<AWSEG_U8>), grid, block, 0, s, logits, hw, label, edges, n_bins, (ece_cell*)workspace);
    else if (vec19)
        hipLaunchKernelGGL((ece19_kernel<AWSEG_I64>), grid, block, 0, s, logits, hw, label, edges, n_bins, (ece_cell*)workspace);
    else if (label_dtype == AWSEG_U8)
        hipLaunchKernelGGL((ece_kernel<AWSEG_U8>), grid, block, 0, s, logits, num_classes, hw, label, edges, n_bins, (ece_cell*)workspace);
    else if (label_dtype == AWSEG_I64)
        hipLaunchKernelGGL((ece_kernel<AWSEG_I64>), grid, block, 0, s, logits, num_classes, hw, label, edges, n_bins, (ece_cell*)workspace);
    else return AWSEG_EINVAL;
    AWSEG_LAUNCH_CHECK();
    hipLaunchKernelGGL(ece_fold_kernel, dim3((unsigned)batch), dim3(64), 0, s, (const ece_cell*)workspace, bpi, n_bins, cond,
                       n_slots, (ece_out*)bins);
    AWSEG_LAUNCH_CHECK();
    return 0;
}

static int stats_impl(const float* seg1, const float* seg2, int64_t batch, int num_classes, int64_t hw,
                      int mode, const float* weights, const float* temperature, const void* label,
                      int label_dtype, const int32_t* cond, const float* edges, int n_bins, void* ece_bins,
                      int n_slots, int64_t* auroc_hist, int n_hist, float hist_lo, float hist_hi,
                      void* workspace, awseg_stream_t stream,
                      bool conf, int ignore_index, int wrap, int64_t* counts, int count_slots, int64_t* oob)
{
    if (!seg1 || !seg2 || !label || !edges || !ece_bins || !auroc_hist || !workspace) return AWSEG_EINVAL;
    if (conf && (!counts || !oob || count_slots < 1)) return AWSEG_EINVAL;
    if (num_classes != 19) return AWSEG_ERANGE;                  // register-resident 2 x 19 x 4 logits
    if (mode != AWSEG_COMBINE_WEIGHTED && mode != AWSEG_COMBINE_MEAN) return AWSEG_ERANGE;
    if (mode == AWSEG_COMBINE_WEIGHTED && !weights) return AWSEG_EINVAL;
    if (n_bins < 1 || n_bins > 64 || n_slots < 1 || n_hist < 2 || n_hist > kHistMax || batch < 1 || batch > 65535 || hw < 4 || !(hist_hi > hist_lo)) return AWSEG_EINVAL;
    if ((hw & 3) || ((uintptr_t)seg1 & 15) || ((uintptr_t)seg2 & 15)) return AWSEG_EALIGN;
    hipStream_t s = awseg_s(stream);
    int bpi = blocks_per_image(hw, batch, 1);
    const int cap = (int)((AWSEG_CUS * 2 + batch - 1) / batch);   // 64 KB of LDS per block: two blocks per CU
    if (bpi > cap) bpi = cap < 1 ? 1 : cap;
    dim3 grid(bpi, (unsigned)batch);
    static int wide = -1;                                           // AWSEG_STATS_WIDE=0: 256 threads x 4 pixels per lane (A/B measurements)
    if (wide < 0) { const char* e = getenv("AWSEG_STATS_WIDE"); wide = e ? atoi(e) : 1; }
    const float scale = (float)n_hist / (hist_hi - hist_lo);
    const size_t lds = (size_t)2 * n_hist * sizeof(uint32_t);
    unsigned long long* hist = (unsigned long long*)auroc_hist;
    // workspace: [batch][bpi][n_bins] ECE cells, then (conf) [batch][bpi][19 x 19] uint32 histogram partials
    uint32_t* conf_partial = reinterpret_cast<uint32_t*>(reinterpret_cast<char*>(workspace) + (size_t)batch * bpi * 64 * sizeof(ece_cell));
#define AWSEG_ES_K(M, L, CF, TH, PX) { \
        if (hipFuncSetAttribute(reinterpret_cast<const void*>(ensemble_stats_kernel<M, L, CF, TH, PX>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds) != hipSuccess) return AWSEG_EINVAL; \
        hipLaunchKernelGGL((ensemble_stats_kernel<M, L, CF, TH, PX>), grid, dim3(TH), lds, s, seg1, seg2, hw, weights, temperature, label, \
                           edges, n_bins, (ece_cell*)workspace, hist, n_hist, hist_lo, scale, ignore_index, wrap, conf_partial, oob); }
#define AWSEG_ES(M, L, CF) { if (wide == 1) AWSEG_ES_K(M, L, CF, 512, 2) else if (wide == 2) AWSEG_ES_K(M, L, CF, 384, 2) else AWSEG_ES_K(M, L, CF, kThreads, 4) }
#define AWSEG_ES2(M, L) { if (conf) AWSEG_ES(M, L, true) else AWSEG_ES(M, L, false) }
    if (mode == AWSEG_COMBINE_WEIGHTED) { if (label_dtype == AWSEG_U8) AWSEG_ES2(0, AWSEG_U8) else if (label_dtype == AWSEG_I64) AWSEG_ES2(0, AWSEG_I64) else return AWSEG_EINVAL; }
    else { if (label_dtype == AWSEG_U8) AWSEG_ES2(2, AWSEG_U8) else if (label_dtype == AWSEG_I64) AWSEG_ES2(2, AWSEG_I64) else return AWSEG_EINVAL; }
#undef AWSEG_ES2
#undef AWSEG_ES
#undef AWSEG_ES_K
    AWSEG_LAUNCH_CHECK();
    hipLaunchKernelGGL(ece_fold_kernel, dim3((unsigned)batch), dim3(64), 0, s, (const ece_cell*)workspace, bpi, n_bins, cond,
                       n_slots, (ece_out*)ece_bins);
    AWSEG_LAUNCH_CHECK();
    if (conf) {
        hipLaunchKernelGGL(fold_partials_kernel, dim3((unsigned)batch, (19 * 19 + 63) / 64), dim3(kFoldSlices * 64), 0, s, conf_partial, bpi,
                           19 * 19, cond, count_slots, counts);
        AWSEG_LAUNCH_CHECK();
    }
    return 0;
}

AWSEG_API int awseg_ensemble_eval_stats(const float* seg1, const float* seg2, int64_t batch, int num_classes, int64_t hw,
                                        int mode, const float* weights, const float* temperature, const void* label,
                                        int label_dtype, const int32_t* cond, const float* edges, int n_bins, void* ece_bins,
                                        int n_slots, int64_t* auroc_hist, int n_hist, float hist_lo, float hist_hi,
                                        void* workspace, awseg_stream_t stream)
{
    return stats_impl(seg1, seg2, batch, num_classes, hw, mode, weights, temperature, label, label_dtype, cond, edges, n_bins,
                      ece_bins, n_slots, auroc_hist, n_hist, hist_lo, hist_hi, workspace, stream, false, 255, 0, nullptr, 0, nullptr);
}

AWSEG_API int awseg_combine_confusion_stats(const float* seg1, const float* seg2, int64_t batch, int num_classes, int64_t hw,
                                            int mode, const float* weights, const float* temperature, const void* label,
                                            int label_dtype, int ignore_index, int label_wrap_u8, const int32_t* cond,
                                            int64_t* counts, int count_slots, int64_t* oob, const float* edges, int n_bins,
                                            void* ece_bins, int ece_slots, int64_t* auroc_hist, int n_hist, float hist_lo,
                                            float hist_hi, void* workspace, awseg_stream_t stream)
{
    return stats_impl(seg1, seg2, batch, num_classes, hw, mode, weights, temperature, label, label_dtype, cond, edges, n_bins,
                      ece_bins, ece_slots, auroc_hist, n_hist, hist_lo, hist_hi, workspace, stream, true, ignore_index,
                      label_wrap_u8, counts, count_slots, oob);
}
