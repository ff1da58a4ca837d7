// Cycle stamps of the split-operand GEMM's K loop: where do a wave's cycles per K tile go?
// Build and run on the GPU box:
//   hipcc --offload-arch=gfx950 -O3 -std=c++17 -ffp-contract=off -DAWSEG_GEMM_STAMP -Iinclude \
//         -Iadverse_weather_semantic_segmentation_robustness_benchmark_amd/csrc tools/probe_gemm_stamps.hip -o /tmp/probe_gemm_stamps && /tmp/probe_gemm_stamps
#include "gemm_split.hip"
#include "gemm_split3.hip"      // awseg_gemm_split3_* (the dispatcher in gemm_split.hip references them)
#include <cstdio>
#include <vector>

int main()
{
    const int64_t M = 65536; const int N = 512, K = 2048;
    float *x, *w, *o; uint16_t* ws;
    hipMalloc(&x, M * K * 4); hipMalloc(&w, (size_t)N * K * 4); hipMalloc(&o, M * N * 4); hipMalloc(&ws, (size_t)awseg_gemm_split_weight_halfs(N, K) * 2);   // [2][N][K] halves + trailer + the k-blocked image: the library's own size
    std::vector<float> hx(M * K), hw((size_t)N * K);
    for (size_t i = 0; i < hx.size(); ++i) hx[i] = (float)((i * 2654435761u) % 2001) / 1000.f - 1.f;
    for (size_t i = 0; i < hw.size(); ++i) hw[i] = ((float)((i * 40503u) % 2001) / 1000.f - 1.f) * 0.05f;
    hipMemcpy(x, hx.data(), hx.size() * 4, hipMemcpyHostToDevice); hipMemcpy(w, hw.data(), hw.size() * 4, hipMemcpyHostToDevice);
    awseg_gemm_split_weights(w, N, K, ws, nullptr);
    for (int rep = 0; rep < 3; ++rep) awseg_gemm_split_bias_act(x, ws, nullptr, nullptr, 1, o, M, N, K, nullptr);
    hipDeviceSynchronize();
    unsigned long long z[8] = {0, 0, 0, 0, 0, 0, 0, 0}, r[8];
    hipMemcpyToSymbol(HIP_SYMBOL(g_gemm_stamp), z, sizeof z);
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    hipEventRecord(e0, nullptr);
    awseg_gemm_split_bias_act(x, ws, nullptr, nullptr, 1, o, M, N, K, nullptr);
    hipEventRecord(e1, nullptr); hipDeviceSynchronize();
    float ms = 0; hipEventElapsedTime(&ms, e0, e1);
    hipMemcpyFromSymbol(r, HIP_SYMBOL(g_gemm_stamp), sizeof r);
    const double n = (double)r[5];
    printf("M=%lld N=%d K=%d: %.3f ms (stamped build); block 0 wave 0: %llu K tiles\n", (long long)M, N, K, ms, r[5]);
    printf("per K tile (s_memtime ticks): reads+MFMAs %.0f | wait for next tile's global loads %.0f | split + LDS writes %.0f | load issue %.0f | barrier %.0f | total %.0f\n",
           r[0] / n, r[1] / n, r[2] / n, r[3] / n, r[4] / n, (r[0] + r[1] + r[2] + r[3] + r[4]) / n);
    return 0;
}
