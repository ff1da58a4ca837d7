// Does the memory-side cache (256 MB) keep what a kernel has just WRITTEN, and does the order in which the next kernel reads it back
// matter?  Kernel W writes S bytes in ascending block order; kernel R reads them back ascending (the start of the buffer was
// written longest ago: gone under LRU if S exceeds the cache) or descending (the end was written last).  HIP events per launch.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e)); exit(1); } } while (0)
__global__ __launch_bounds__(256) void wr(float4* p, long n4, float v)
{
    for (long i = (long)blockIdx.x * 256 + threadIdx.x; i < n4; i += (long)gridDim.x * 256) p[i] = make_float4(v, v, v, v);
}
// chunked: block b owns a contiguous chunk; rev: chunks taken from the end
__global__ __launch_bounds__(256) void wr_chunk(float4* p, long chunk4, int nchunks, float v, int rev)
{
    for (int c = blockIdx.x; c < nchunks; c += gridDim.x) {
        const int cc = rev ? nchunks - 1 - c : c;
        float4* q = p + (long)cc * chunk4;
        for (long i = threadIdx.x; i < chunk4; i += 256) q[i] = make_float4(v, v, v, v);
    }
}
__global__ __launch_bounds__(256) void rd_chunk(const float4* p, long chunk4, int nchunks, float* sink, int rev)
{
    float acc = 0.f;
    for (int c = blockIdx.x; c < nchunks; c += gridDim.x) {
        const int cc = rev ? nchunks - 1 - c : c;
        const float4* q = p + (long)cc * chunk4;
        for (long i = threadIdx.x; i < chunk4; i += 256) { const float4 t = q[i]; acc += t.x + t.y + t.z + t.w; }
    }
    if (acc == 123.456f) sink[0] = acc;
}
int main()
{
    const long MB = 1 << 20;
    float4* buf; float* sink;
    CK(hipMalloc(&buf, 2048 * MB)); CK(hipMalloc(&sink, 4));
    hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    const long sizes[] = {32, 64, 128, 192, 256, 384, 512, 1024};
    const long chunk = 256 * 1024;                                 // bytes per chunk
    for (long s : sizes) {
        const long bytes = s * MB; const int nch = (int)(bytes / chunk); const long c4 = chunk / 16;
        for (int rev = 0; rev < 2; ++rev) {
            float tw = 0, tr = 0; const int reps = 5;
            for (int it = 0; it < reps + 1; ++it) {
                float a, b;
                CK(hipEventRecord(e0)); hipLaunchKernelGGL(wr_chunk, dim3(2048), dim3(256), 0, 0, buf, c4, nch, (float)it, 0); CK(hipEventRecord(e1));
                CK(hipEventSynchronize(e1)); CK(hipEventElapsedTime(&a, e0, e1));
                CK(hipEventRecord(e0)); hipLaunchKernelGGL(rd_chunk, dim3(2048), dim3(256), 0, 0, buf, c4, nch, sink, rev); CK(hipEventRecord(e1));
                CK(hipEventSynchronize(e1)); CK(hipEventElapsedTime(&b, e0, e1));
                if (it) { tw += a; tr += b; }
            }
            printf("%5ld MB: write %7.1f us (%6.0f GB/s), read back %s %7.1f us (%6.0f GB/s)\n", s, tw / reps * 1e3, bytes / (tw / reps) / 1e6,
                   rev ? "descending" : "ascending ", tr / reps * 1e3, bytes / (tr / reps) / 1e6);
        }
    }
    return 0;
}
