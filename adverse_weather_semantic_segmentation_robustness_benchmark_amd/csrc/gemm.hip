// gemm.hip — 1x1 stride-1 convolutions of the eval forward as ONE hipBLASLt call each, with the whole
// Conv -> BatchNorm(folded) -> [+identity] -> ReLU epilogue riding on the GEMM:
//     out[M,N] = act(x[M,K] . w[N,K]^T + bias[N] (+ residual[M,N]))        (row-major, fp32)
// On NHWC activations a pointwise convolution IS this GEMM (M = B*H*W pixels, K = Cin, N = Cout).  The
// plain library GEMM is hipBLASLt's job (it runs these shapes at 100-145 TFLOP/s of the 157 peak); what
// this wrapper adds over calling it through torch is the residual as the beta*C operand TOGETHER with the
// bias + ReLU epilogue, which removes the separate bias/add/ReLU pass over the largest activations of the
// ResNet bottlenecks (conv3 of PKG/models/model.py:349's encoder: 3 x 4*planes*pixels*4 B per block).
//
// hipBLASLt is column-major: row-major out[M,N] is the column-major N x M matrix D = W'^T . X' with
// W' = w seen as K x N (ld K) and X' = x seen as K x M (ld K); the bias is per row of D = per output channel.
// The library handle and the per-shape algorithm choice (top heuristic) are created on first use and cached
// (host state only); device workspace comes from the caller.
#include "awseg_common.h"
#include <hipblaslt/hipblaslt.h>
#include <map>
#include <mutex>
#include <tuple>
#include <vector>

namespace {

struct plan {
    hipblasLtMatmulDesc_t desc = nullptr;
    hipblasLtMatrixLayout_t a = nullptr, b = nullptr, c = nullptr;
    hipblasLtMatmulAlgo_t algo;
    std::vector<hipblasLtMatmulAlgo_t> candidates;     // the heuristic's ranked list (awseg_gemm_tune times them)
    size_t ws = 0;
    bool ok = false, tuned = false;
};

std::mutex g_mu;
hipblasLtHandle_t g_handle = nullptr;
std::map<std::tuple<int64_t, int, int, int, int, size_t>, plan> g_plans;

plan* get_plan(int64_t M, int N, int K, int has_res, int act, size_t ws_bytes)
{
    std::lock_guard<std::mutex> lk(g_mu);
    if (!g_handle && hipblasLtCreate(&g_handle) != HIPBLAS_STATUS_SUCCESS) return nullptr;
    auto key = std::make_tuple(M, N, K, has_res, act, ws_bytes);
    auto it = g_plans.find(key);
    if (it != g_plans.end()) return it->second.ok ? &it->second : nullptr;
    plan p;
    hipblasOperation_t opT = HIPBLAS_OP_T, opN = HIPBLAS_OP_N;
    hipblasLtEpilogue_t epi = act == AWSEG_ACT_RELU ? HIPBLASLT_EPILOGUE_RELU_BIAS : HIPBLASLT_EPILOGUE_BIAS;
    hipblasLtMatmulPreference_t pref = nullptr;
    constexpr int kMaxCand = 24;
    hipblasLtMatmulHeuristicResult_t res[kMaxCand];
    int found = 0;
    bool ok = hipblasLtMatmulDescCreate(&p.desc, HIPBLAS_COMPUTE_32F, HIP_R_32F) == HIPBLAS_STATUS_SUCCESS
           && hipblasLtMatmulDescSetAttribute(p.desc, HIPBLASLT_MATMUL_DESC_TRANSA, &opT, sizeof(opT)) == HIPBLAS_STATUS_SUCCESS
           && hipblasLtMatmulDescSetAttribute(p.desc, HIPBLASLT_MATMUL_DESC_TRANSB, &opN, sizeof(opN)) == HIPBLAS_STATUS_SUCCESS
           && hipblasLtMatmulDescSetAttribute(p.desc, HIPBLASLT_MATMUL_DESC_EPILOGUE, &epi, sizeof(epi)) == HIPBLAS_STATUS_SUCCESS
           && hipblasLtMatrixLayoutCreate(&p.a, HIP_R_32F, (uint64_t)K, (uint64_t)N, K) == HIPBLAS_STATUS_SUCCESS      // W' : K x N
           && hipblasLtMatrixLayoutCreate(&p.b, HIP_R_32F, (uint64_t)K, (uint64_t)M, K) == HIPBLAS_STATUS_SUCCESS      // X' : K x M
           && hipblasLtMatrixLayoutCreate(&p.c, HIP_R_32F, (uint64_t)N, (uint64_t)M, N) == HIPBLAS_STATUS_SUCCESS      // C, D : N x M
           && hipblasLtMatmulPreferenceCreate(&pref) == HIPBLAS_STATUS_SUCCESS
           && hipblasLtMatmulPreferenceSetAttribute(pref, HIPBLASLT_MATMUL_PREF_MAX_WORKSPACE_BYTES, &ws_bytes, sizeof(ws_bytes)) == HIPBLAS_STATUS_SUCCESS;
    if (ok) {
        // the bias pointer is part of the heuristic query's problem description: any non-null value will do here
        const void* dummy = reinterpret_cast<const void*>(uintptr_t(16));
        ok = hipblasLtMatmulDescSetAttribute(p.desc, HIPBLASLT_MATMUL_DESC_BIAS_POINTER, &dummy, sizeof(dummy)) == HIPBLAS_STATUS_SUCCESS
          && hipblasLtMatmulAlgoGetHeuristic(g_handle, p.desc, p.a, p.b, p.c, p.c, pref, kMaxCand, res, &found) == HIPBLAS_STATUS_SUCCESS
          && found > 0;
    }
    if (pref) hipblasLtMatmulPreferenceDestroy(pref);
    if (ok) {
        p.algo = res[0].algo; p.ws = res[0].workspaceSize; p.ok = true;
        for (int i = 0; i < found; ++i)
            if (res[i].state == HIPBLAS_STATUS_SUCCESS && res[i].workspaceSize <= ws_bytes) p.candidates.push_back(res[i].algo);
    }
    auto ins = g_plans.emplace(key, p);
    return ins.first->second.ok ? &ins.first->second : nullptr;
}

}  // namespace

AWSEG_API int awseg_gemm_bias_act(const float* x, const float* w, const float* bias, const float* residual, int act,
                                  float* out, int64_t m, int n, int k, void* workspace, size_t workspace_bytes,
                                  awseg_stream_t stream)
{
    if (m == 0) return 0;
    if (!x || !w || !bias || !out || m < 0 || n < 1 || k < 1) return AWSEG_EINVAL;
    if (act != AWSEG_ACT_NONE && act != AWSEG_ACT_RELU) return AWSEG_ERANGE;
    if (workspace_bytes && !workspace) return AWSEG_EINVAL;
    plan* p = get_plan(m, n, k, residual ? 1 : 0, act, workspace_bytes);
    if (!p) return AWSEG_ERANGE;                 // no hipBLASLt solution for this problem with this workspace
    // The descriptor is shared by every call of this shape; the bias pointer is per call, so the set + launch
    // pair is serialised (launches are asynchronous: the lock is held for microseconds).
    std::lock_guard<std::mutex> lk(g_mu);
    const void* bp = bias;
    if (hipblasLtMatmulDescSetAttribute(p->desc, HIPBLASLT_MATMUL_DESC_BIAS_POINTER, &bp, sizeof(bp)) != HIPBLAS_STATUS_SUCCESS) return AWSEG_EINVAL;
    const float alpha = 1.0f, beta = residual ? 1.0f : 0.0f;
    const hipblasStatus_t st = hipblasLtMatmul(g_handle, p->desc, &alpha, w, p->a, x, p->b, &beta, residual ? residual : out, p->c,
                                               out, p->c, &p->algo, workspace, workspace_bytes, awseg_s(stream));
    return st == HIPBLAS_STATUS_SUCCESS ? 0 : AWSEG_EINVAL;
}

// Optional, explicit, SYNCHRONISING: time the library's ranked candidates for one problem on the caller's
// buffers and keep the fastest for later awseg_gemm_bias_act calls of the same (M,N,K,residual,act,workspace).
// `scratch_out` [M,N] is overwritten (it also stands in for the residual).  Returns the number of candidates
// timed (>= 1), or a negative error.  Meant for warm-up; never called implicitly.
AWSEG_API int awseg_gemm_tune(const float* x, const float* w, const float* bias, int has_residual, int act,
                              float* scratch_out, int64_t m, int n, int k, void* workspace, size_t workspace_bytes,
                              awseg_stream_t stream)
{
    if (!x || !w || !bias || !scratch_out || m < 1 || n < 1 || k < 1) return AWSEG_EINVAL;
    plan* p = get_plan(m, n, k, has_residual ? 1 : 0, act, workspace_bytes);
    if (!p) return AWSEG_ERANGE;
    std::lock_guard<std::mutex> lk(g_mu);
    if (p->tuned) return (int)p->candidates.size();
    const void* bp = bias;
    if (hipblasLtMatmulDescSetAttribute(p->desc, HIPBLASLT_MATMUL_DESC_BIAS_POINTER, &bp, sizeof(bp)) != HIPBLAS_STATUS_SUCCESS) return AWSEG_EINVAL;
    const float alpha = 1.0f, beta = has_residual ? 1.0f : 0.0f;
    hipStream_t s = awseg_s(stream);
    hipEvent_t e0, e1;
    if (hipEventCreate(&e0) != hipSuccess || hipEventCreate(&e1) != hipSuccess) return AWSEG_EINVAL;
    float best = 1e30f;
    int timed = 0;
    for (const hipblasLtMatmulAlgo_t& cand : p->candidates) {
        bool ok = true;
        for (int rep = 0; rep < 4 && ok; ++rep) {                       // 1 warm + 3 timed
            if (rep == 1) (void)hipEventRecord(e0, s);
            ok = hipblasLtMatmul(g_handle, p->desc, &alpha, w, p->a, x, p->b, &beta, scratch_out, p->c, scratch_out, p->c, &cand,
                                 workspace, workspace_bytes, s) == HIPBLAS_STATUS_SUCCESS;
        }
        (void)hipEventRecord(e1, s);
        if (hipEventSynchronize(e1) != hipSuccess || !ok) continue;
        float ms = 0.f;
        (void)hipEventElapsedTime(&ms, e0, e1);
        ++timed;
        if (ms < best) { best = ms; p->algo = cand; }
    }
    (void)hipEventDestroy(e0); (void)hipEventDestroy(e1);
    p->tuned = true;
    return timed > 0 ? timed : AWSEG_ERANGE;
}
