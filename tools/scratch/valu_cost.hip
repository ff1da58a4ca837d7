// Issue cost of the vector instructions the split-operand kernels lean on: N dependent-free instructions per loop iteration,
// one or two waves per SIMD; cycles per instruction = (s_memtime delta) / (iterations * N).
#include <hip/hip_runtime.h>
#include <cstdio>
#define REP16(x) x x x x x x x x x x x x x x x x
template <int KIND>
__global__ __launch_bounds__(512) void k(unsigned long long* out, float seed)
{
    float a0 = seed + threadIdx.x, a1 = a0 * 1.5f, a2 = a0 + 2.f, a3 = a0 - 3.f, b0 = 1.0001f, b1 = 0.5f;
    unsigned u0 = threadIdx.x, u1 = 77u, u2 = 0u, u3 = 0u;
    typedef float v2f __attribute__((ext_vector_type(2)));
    v2f p0 = {a0, a1}, p1 = {a2, a3}, p2 = {b0, b1}, p3 = {a1, a2};
    unsigned long long t0 = __builtin_readcyclecounter();
    for (int it = 0; it < 256; ++it) {
        if (KIND == 0) { REP16(asm volatile("v_add_f32 %0, %0, %4\n\tv_add_f32 %1, %1, %4\n\tv_add_f32 %2, %2, %4\n\tv_add_f32 %3, %3, %4" : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3) : "v"(b0));) }
        if (KIND == 1) { REP16(asm volatile("v_pk_add_f32 %0, %0, %4\n\tv_pk_add_f32 %1, %1, %4\n\tv_pk_add_f32 %2, %2, %4\n\tv_pk_add_f32 %3, %3, %4" : "+v"(p0), "+v"(p1), "+v"(p2), "+v"(p3) : "v"(p2));) }
        if (KIND == 2) { REP16(asm volatile("v_fma_mixlo_f16 %0, %4, 1.0, %5\n\tv_fma_mixlo_f16 %1, %4, 1.0, %5\n\tv_fma_mixlo_f16 %2, %5, 1.0, %4\n\tv_fma_mixlo_f16 %3, %5, 1.0, %4" : "+v"(u0), "+v"(u1), "+v"(u2), "+v"(u3) : "v"(a0), "v"(a1));) }
        if (KIND == 3) { REP16(asm volatile("v_cvt_pkrtz_f16_f32 %0, %4, %5\n\tv_cvt_pkrtz_f16_f32 %1, %4, %5\n\tv_cvt_pkrtz_f16_f32 %2, %5, %4\n\tv_cvt_pkrtz_f16_f32 %3, %5, %4" : "+v"(u0), "+v"(u1), "+v"(u2), "+v"(u3) : "v"(a0), "v"(a1));) }
        if (KIND == 4) { REP16(asm volatile("v_max3_f32 %0, %0, |%4|, |%5|\n\tv_max3_f32 %1, %1, |%4|, |%5|\n\tv_max3_f32 %2, %2, |%5|, |%4|\n\tv_max3_f32 %3, %3, |%5|, |%4|" : "+v"(a2), "+v"(a3), "+v"(b0), "+v"(b1) : "v"(a0), "v"(a1));) }
        if (KIND == 5) { REP16(asm volatile("v_fma_f32 %0, %0, %4, %5\n\tv_fma_f32 %1, %1, %4, %5\n\tv_fma_f32 %2, %2, %4, %5\n\tv_fma_f32 %3, %3, %4, %5" : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3) : "v"(b0), "v"(b1));) }
        if (KIND == 6) { REP16(asm volatile("v_pk_fma_f32 %0, %0, %4, %4\n\tv_pk_fma_f32 %1, %1, %4, %4\n\tv_pk_fma_f32 %2, %2, %4, %4\n\tv_pk_fma_f32 %3, %3, %4, %4" : "+v"(p0), "+v"(p1), "+v"(p3), "+v"(p2) : "v"(p2));) }
        if (KIND == 7) { REP16(asm volatile("v_cvt_f32_ubyte0 %0, %4\n\tv_cvt_f32_ubyte1 %1, %4\n\tv_cvt_f32_ubyte2 %2, %4\n\tv_cvt_f32_ubyte3 %3, %4" : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3) : "v"(u0));) }
        if (KIND == 8) { REP16(asm volatile("v_exp_f32 %0, %4\n\tv_exp_f32 %1, %4\n\tv_exp_f32 %2, %5\n\tv_exp_f32 %3, %5" : "+v"(a2), "+v"(a3), "+v"(b0), "+v"(b1) : "v"(a0), "v"(a1));) }
        if (KIND == 9) { REP16(asm volatile("v_mul_hi_u32 %0, %4, %5\n\tv_mul_hi_u32 %1, %4, %5\n\tv_mul_lo_u32 %2, %5, %4\n\tv_mul_lo_u32 %3, %5, %4" : "+v"(u0), "+v"(u1), "+v"(u2), "+v"(u3) : "v"(u0), "v"(u1));) }
        if (KIND == 10) { REP16(asm volatile("v_med3_f32 %0, %0, %4, %5\n\tv_med3_f32 %1, %1, %4, %5\n\tv_med3_f32 %2, %2, %4, %5\n\tv_med3_f32 %3, %3, %4, %5" : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3) : "v"(b0), "v"(b1));) }
        if (KIND == 11) { REP16(asm volatile("v_pk_mul_f16 %0, %0, %4\n\tv_pk_mul_f16 %1, %1, %4\n\tv_pk_add_f16 %2, %2, %4\n\tv_pk_add_f16 %3, %3, %4" : "+v"(u0), "+v"(u1), "+v"(u2), "+v"(u3) : "v"(u1));) }
    }
    unsigned long long t1 = __builtin_readcyclecounter();
    if (threadIdx.x % 64 == 0 && blockIdx.x == 0) out[threadIdx.x / 64] = t1 - t0;
    if (a0 + a1 + a2 + a3 + b0 + b1 + p0.x + p1.x + p2.x + p3.y == 12345.f && u0 + u1 + u2 + u3 == 7u) out[63] = 1;
}
template <int KIND> void run(const char* name, unsigned long long* d)
{
    for (int threads : {256, 512, 1024}) {
        hipLaunchKernelGGL(k<KIND>, dim3(256), dim3(threads), 0, 0, d, 1.0f);
        hipDeviceSynchronize();
        unsigned long long h[16]; hipMemcpy(h, d, sizeof h, hipMemcpyDeviceToHost);
        printf("%-24s %d waves/SIMD: %.2f cycles per instruction (wave 0)\n", name, threads / 256, (double)h[0] / (256.0 * 64));
    }
}
int main()
{
    unsigned long long* d; hipMalloc(&d, 64 * 8);
    run<0>("v_add_f32", d); run<5>("v_fma_f32", d); run<1>("v_pk_add_f32", d); run<6>("v_pk_fma_f32", d); run<2>("v_fma_mixlo_f16", d); run<3>("v_cvt_pkrtz_f16_f32", d);
    run<4>("v_max3_f32 |abs|", d); run<10>("v_med3_f32", d); run<7>("v_cvt_f32_ubyteN", d); run<8>("v_exp_f32", d); run<9>("v_mul_hi/lo_u32", d); run<11>("v_pk_mul/add_f16", d);
    return 0;
}
