// Do the MFMA, VALU and LDS pipes of one SIMD run side by side when DIFFERENT waves feed them?  A 512-thread block (8 waves);
// each wave gets a role (MFMA chain / VALU chain / LDS-read chain) from its index by one of two maps — by parity (wave & 1) or by
// half (wave >> 2) — and reports its cycle count and the SIMD it ran on (HW_ID).  hipcc --offload-arch=gfx950 -O3 pipe_overlap.hip
#include <hip/hip_runtime.h>
#include <cstdio>
typedef _Float16 h8 __attribute__((ext_vector_type(8)));
typedef float f16v __attribute__((ext_vector_type(16)));
#define REP8(x) x x x x x x x x

__device__ __forceinline__ void mfma_chain(f16v (&acc)[4], h8 a, h8 b)
{
    for (int it = 0; it < 128; ++it) {
        REP8(asm volatile("v_mfma_f32_32x32x16_f16 %0, %4, %5, %0\n\tv_mfma_f32_32x32x16_f16 %1, %4, %5, %1\n\t"
                          "v_mfma_f32_32x32x16_f16 %2, %4, %5, %2\n\tv_mfma_f32_32x32x16_f16 %3, %4, %5, %3"
                          : "+v"(acc[0]), "+v"(acc[1]), "+v"(acc[2]), "+v"(acc[3]) : "v"(a), "v"(b));)
    }
}
__device__ __forceinline__ void valu_chain(float (&r)[4], float c, int iters)
{
    for (int it = 0; it < iters; ++it) {
        REP8(REP8(asm volatile("v_fma_f32 %0, %0, %4, %4\n\tv_fma_f32 %1, %1, %4, %4\n\tv_fma_f32 %2, %2, %4, %4\n\tv_fma_f32 %3, %3, %4, %4"
                               : "+v"(r[0]), "+v"(r[1]), "+v"(r[2]), "+v"(r[3]) : "v"(c));))
    }
}
__device__ __forceinline__ void lds_chain(float4 (&q)[4], unsigned addr, int iters)
{
    for (int it = 0; it < iters; ++it) {
        REP8(asm volatile("ds_read_b128 %0, %4\n\tds_read_b128 %1, %4 offset:1024\n\tds_read_b128 %2, %4 offset:2048\n\tds_read_b128 %3, %4 offset:3072\n\ts_waitcnt lgkmcnt(0)"
                          : "=v"(q[0]), "=v"(q[1]), "=v"(q[2]), "=v"(q[3]) : "v"(addr));)
    }
}

// one vector instruction type per chain (the Winograd transform's mix): which of them shares a pipe with the MFMAs?
template <int KIND>
__device__ __forceinline__ void valu_kind_chain(float (&r)[4], float c, int iters)
{
    typedef float v2 __attribute__((ext_vector_type(2)));
    v2 p0 = {r[0], r[1]}, p1 = {r[2], r[3]}, p2 = {c, c}, p3 = {r[1], r[2]};
    unsigned u0 = 1, u1 = 2, u2 = 3, u3 = 4;
    for (int it = 0; it < iters; ++it) {
        if (KIND == 0) { REP8(REP8(asm volatile("v_pk_add_f32 %0, %0, %4\n\tv_pk_add_f32 %1, %1, %4\n\tv_pk_add_f32 %2, %2, %4 neg_lo:[0,1] neg_hi:[0,1]\n\tv_pk_add_f32 %3, %3, %4" : "+v"(p0), "+v"(p1), "+v"(p3), "+v"(p2) : "v"(p2));)) }
        if (KIND == 1) { REP8(REP8(asm volatile("v_fma_mixlo_f16 %0, %4, 1.0, -%5 op_sel_hi:[0,0,1]\n\tv_fma_mixhi_f16 %1, %4, 1.0, -%5 op_sel:[0,0,1] op_sel_hi:[0,0,1]\n\tv_fma_mixlo_f16 %2, %5, 1.0, -%4 op_sel_hi:[0,0,1]\n\tv_fma_mixhi_f16 %3, %5, 1.0, -%4 op_sel:[0,0,1] op_sel_hi:[0,0,1]" : "+v"(u0), "+v"(u1), "+v"(u2), "+v"(u3) : "v"(r[0]), "v"(u0));)) }
        if (KIND == 2) { REP8(REP8(asm volatile("v_cvt_pkrtz_f16_f32 %0, %4, %5\n\tv_cvt_pkrtz_f16_f32 %1, %4, %5\n\tv_cvt_pkrtz_f16_f32 %2, %5, %4\n\tv_cvt_pkrtz_f16_f32 %3, %5, %4" : "+v"(u0), "+v"(u1), "+v"(u2), "+v"(u3) : "v"(r[0]), "v"(r[1]));)) }
        if (KIND == 3) { REP8(REP8(asm volatile("v_max3_f32 %0, %0, |%1|, |%2|\n\tv_max3_f32 %0, %0, |%2|, |%1|\n\tv_max3_f32 %0, %0, |%1|, |%2|\n\tv_max3_f32 %0, %0, |%2|, |%1|" : "+v"(r[2]) : "v"(r[0]), "v"(r[1]));)) }
    }
    r[3] += p0.x + p1.y + p2.x + p3.y + (float)(u0 + u1 + u2 + u3);
}

// ONE wave doing both jobs in one stream: an MFMA (rotating over four accumulators), then a transform piece — 4 plain vector ops,
// a v_cvt_pkrtz and 2 v_fma_mix — with an operand ds_read_b128 every third and a ds_write_b64 every second step
__device__ __forceinline__ void fused_like_chain(f16v (&acc)[4], float (&r)[4], unsigned addr, h8 b, int iters)
{
    h8 v0 = b, v1 = b;
    unsigned u0 = 1, u1 = 2;
    for (int it = 0; it < iters; ++it) {
        REP8(asm volatile("v_mfma_f32_32x32x16_f16 %0, %4, %6, %0\n\tv_fma_f32 %7, %7, %11, %11\n\tv_fma_f32 %8, %8, %11, %11\n\tv_fma_f32 %9, %9, %11, %11\n\tv_fma_f32 %10, %10, %11, %11\n\t"
                          "v_cvt_pkrtz_f16_f32 %12, %7, %8\n\tv_fma_mixlo_f16 %13, %7, 1.0, -%12 op_sel_hi:[0,0,1]\n\tv_fma_mixhi_f16 %13, %8, 1.0, -%12 op_sel:[0,0,1] op_sel_hi:[0,0,1]\n\t"
                          "ds_read_b128 %5, %14 offset:4096\n\t"
                          "v_mfma_f32_32x32x16_f16 %1, %4, %6, %1\n\tv_fma_f32 %7, %7, %11, %11\n\tv_fma_f32 %8, %8, %11, %11\n\tv_fma_f32 %9, %9, %11, %11\n\tv_fma_f32 %10, %10, %11, %11\n\t"
                          "v_cvt_pkrtz_f16_f32 %12, %9, %10\n\tv_fma_mixlo_f16 %13, %9, 1.0, -%12 op_sel_hi:[0,0,1]\n\tv_fma_mixhi_f16 %13, %10, 1.0, -%12 op_sel:[0,0,1] op_sel_hi:[0,0,1]\n\t"
                          "ds_write_b64 %14, %15 offset:16384\n\t"
                          "s_waitcnt lgkmcnt(1)\n\t"
                          "v_mfma_f32_32x32x16_f16 %2, %5, %6, %2\n\tv_fma_f32 %7, %7, %11, %11\n\tv_fma_f32 %8, %8, %11, %11\n\tv_fma_f32 %9, %9, %11, %11\n\tv_fma_f32 %10, %10, %11, %11\n\t"
                          "v_cvt_pkrtz_f16_f32 %12, %7, %8\n\tv_fma_mixlo_f16 %13, %7, 1.0, -%12 op_sel_hi:[0,0,1]\n\tv_fma_mixhi_f16 %13, %8, 1.0, -%12 op_sel:[0,0,1] op_sel_hi:[0,0,1]\n\t"
                          "ds_read_b128 %4, %14\n\t"
                          "v_mfma_f32_32x32x16_f16 %3, %5, %6, %3\n\tv_fma_f32 %7, %7, %11, %11\n\tv_fma_f32 %8, %8, %11, %11\n\tv_fma_f32 %9, %9, %11, %11\n\tv_fma_f32 %10, %10, %11, %11\n\t"
                          "v_cvt_pkrtz_f16_f32 %12, %9, %10\n\tv_fma_mixlo_f16 %13, %9, 1.0, -%12 op_sel_hi:[0,0,1]\n\tv_fma_mixhi_f16 %13, %10, 1.0, -%12 op_sel:[0,0,1] op_sel_hi:[0,0,1]\n\t"
                          "ds_write_b64 %14, %15 offset:20480\n\t"
                          "s_waitcnt lgkmcnt(1)"
                          : "+v"(acc[0]), "+v"(acc[1]), "+v"(acc[2]), "+v"(acc[3]), "+v"(v0), "+v"(v1)
                          : "v"(b), "v"(r[0]), "v"(r[1]), "v"(r[2]), "v"(r[3]), "v"(1.0001f), "v"(u0), "v"(u1), "v"(addr), "v"(*reinterpret_cast<unsigned long long*>(&r[0])) : "memory");)
    }
}

// MFMA chain on ONE accumulator: every MFMA waits for its predecessor's result (SrcC)
__device__ __forceinline__ void mfma_dep_chain(f16v& acc, h8 a, h8 b)
{
    for (int it = 0; it < 128; ++it) {
        REP8(asm volatile("v_mfma_f32_32x32x16_f16 %0, %1, %2, %0\n\tv_mfma_f32_32x32x16_f16 %0, %1, %2, %0\n\t"
                          "v_mfma_f32_32x32x16_f16 %0, %1, %2, %0\n\tv_mfma_f32_32x32x16_f16 %0, %1, %2, %0" : "+v"(acc) : "v"(a), "v"(b));)
    }
}
// the multiplying wave of the Winograd kernel: two operand reads from LDS, then 3 + 3 MFMAs on two accumulators (dependent triples)
__device__ __forceinline__ void mfma_like_chain(f16v (&acc)[4], unsigned addr, h8 b)
{
    h8 v0, v1, v2, v3;
    for (int it = 0; it < 170; ++it) {
        REP8(asm volatile("ds_read_b128 %2, %6\n\tds_read_b128 %3, %6 offset:4096\n\tds_read_b128 %4, %6 offset:2048\n\tds_read_b128 %5, %6 offset:6144\n\t"
                          "s_waitcnt lgkmcnt(2)\n\tv_mfma_f32_32x32x16_f16 %0, %2, %7, %0\n\tv_mfma_f32_32x32x16_f16 %0, %2, %7, %0\n\tv_mfma_f32_32x32x16_f16 %0, %3, %7, %0\n\t"
                          "s_waitcnt lgkmcnt(0)\n\tv_mfma_f32_32x32x16_f16 %1, %4, %7, %1\n\tv_mfma_f32_32x32x16_f16 %1, %4, %7, %1\n\tv_mfma_f32_32x32x16_f16 %1, %5, %7, %1"
                          : "+v"(acc[0]), "+v"(acc[1]), "=&v"(v0), "=&v"(v1), "=&v"(v2), "=&v"(v3) : "v"(addr), "v"(b));)
    }
}
// same with the six MFMAs alternating between the two accumulators (dependency distance 2)
__device__ __forceinline__ void mfma_alt_chain(f16v (&acc)[4], unsigned addr, h8 b)
{
    h8 v0, v1, v2, v3;
    for (int it = 0; it < 170; ++it) {
        REP8(asm volatile("ds_read_b128 %2, %6\n\tds_read_b128 %3, %6 offset:4096\n\tds_read_b128 %4, %6 offset:2048\n\tds_read_b128 %5, %6 offset:6144\n\t"
                          "s_waitcnt lgkmcnt(0)\n\tv_mfma_f32_32x32x16_f16 %0, %2, %7, %0\n\tv_mfma_f32_32x32x16_f16 %1, %4, %7, %1\n\tv_mfma_f32_32x32x16_f16 %0, %2, %7, %0\n\t"
                          "v_mfma_f32_32x32x16_f16 %1, %4, %7, %1\n\tv_mfma_f32_32x32x16_f16 %0, %3, %7, %0\n\tv_mfma_f32_32x32x16_f16 %1, %5, %7, %1"
                          : "+v"(acc[0]), "+v"(acc[1]), "=&v"(v0), "=&v"(v1), "=&v"(v2), "=&v"(v3) : "v"(addr), "v"(b));)
    }
}
// the transforming wave: 4 x 16-byte LDS reads, ~36 vector ops, 6 x 8-byte LDS writes, per step; optional vector-memory loads
__device__ __forceinline__ void xform_like_chain(float (&r)[4], float4 (&q)[4], unsigned addr, const float4* g, int iters, bool with_vmem)
{
    float4 u0 = {}, u1 = {};
    typedef float v2 __attribute__((ext_vector_type(2)));
    v2 w0 = {r[0], r[1]}, w1 = {r[2], r[3]};
    for (int it = 0; it < iters; ++it) {
        if (with_vmem) asm volatile("global_load_dwordx4 %0, %2, off\n\tglobal_load_dwordx4 %1, %2, off offset:1024" : "=&v"(u0), "=&v"(u1) : "v"(g + (threadIdx.x & 63)) : "memory");
        REP8(asm volatile("ds_read_b128 %0, %8\n\tds_read_b128 %1, %8 offset:1152\n\tds_read_b128 %2, %8 offset:2304\n\tds_read_b128 %3, %8 offset:3456\n\ts_waitcnt lgkmcnt(0)\n\t"
                          "v_pk_add_f32 %10, %10, %11\n\tv_pk_add_f32 %11, %11, %10\n\tv_fma_f32 %4, %4, %9, %9\n\tv_fma_f32 %5, %5, %9, %9\n\tv_fma_f32 %6, %6, %9, %9\n\tv_fma_f32 %7, %7, %9, %9\n\t"
                          "v_fma_f32 %4, %4, %9, %9\n\tv_fma_f32 %5, %5, %9, %9\n\tv_fma_f32 %6, %6, %9, %9\n\tv_fma_f32 %7, %7, %9, %9\n\t"
                          "v_fma_f32 %4, %4, %9, %9\n\tv_fma_f32 %5, %5, %9, %9\n\tv_fma_f32 %6, %6, %9, %9\n\tv_fma_f32 %7, %7, %9, %9\n\t"
                          "v_fma_f32 %4, %4, %9, %9\n\tv_fma_f32 %5, %5, %9, %9\n\tv_fma_f32 %6, %6, %9, %9\n\tv_fma_f32 %7, %7, %9, %9\n\t"
                          "v_fma_f32 %4, %4, %9, %9\n\tv_fma_f32 %5, %5, %9, %9\n\tv_fma_f32 %6, %6, %9, %9\n\tv_fma_f32 %7, %7, %9, %9\n\t"
                          "v_fma_f32 %4, %4, %9, %9\n\tv_fma_f32 %5, %5, %9, %9\n\tv_fma_f32 %6, %6, %9, %9\n\tv_fma_f32 %7, %7, %9, %9\n\t"
                          "v_fma_f32 %4, %4, %9, %9\n\tv_fma_f32 %5, %5, %9, %9\n\tv_fma_f32 %6, %6, %9, %9\n\tv_fma_f32 %7, %7, %9, %9\n\t"
                          "v_fma_f32 %4, %4, %9, %9\n\tv_fma_f32 %5, %5, %9, %9\n\tv_fma_f32 %6, %6, %9, %9\n\tv_fma_f32 %7, %7, %9, %9\n\t"
                          "ds_write_b64 %8, %10 offset:16384\n\tds_write_b64 %8, %11 offset:20480\n\tds_write_b64 %8, %10 offset:24576\n\tds_write_b64 %8, %11 offset:28672\n\t"
                          "ds_write_b64 %8, %10 offset:32768\n\tds_write_b64 %8, %11 offset:36864"
                          : "=&v"(q[0]), "=&v"(q[1]), "=&v"(q[2]), "=&v"(q[3]), "+v"(r[0]), "+v"(r[1]), "+v"(r[2]), "+v"(r[3]) : "v"(addr), "v"(1.0001f), "v"(w0), "v"(w1) : "memory");)
        if (with_vmem) { asm volatile("s_waitcnt vmcnt(0)" ::: "memory"); r[0] += u0.x + u1.y; }
    }
}

// roles: 0 = MFMA, 1 = VALU, 2 = LDS, 3 = idle, 4 = MFMA on one accumulator, 5 = LDS reads + dependent MFMA triples, 6 = same, alternating accumulators,
//        7 = transform-like (LDS reads, vector ops, LDS writes), 8 = 7 + vector-memory loads.  map: role_a for waves where sel(wave) == 0, role_b otherwise.
__global__ __launch_bounds__(512) void k(unsigned long long* out, const float4* g, int role_a, int role_b, int by_half, int valu_iters, int lds_iters, float seed)
{
    __shared__ float4 lds[4096];
    for (int i = threadIdx.x; i < 4096; i += 512) lds[i] = make_float4(seed, seed, seed, seed);
    __syncthreads();
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int sel = by_half ? (wave >> 2) : (wave & 1);
    const int role = sel ? role_b : role_a;
    f16v acc[4] = {};
    h8 a, b;
    for (int i = 0; i < 8; ++i) { a[i] = (_Float16)(seed + i); b[i] = (_Float16)(0.5f * seed); }
    float r[4] = {seed, seed + 1, seed + 2, seed + 3};
    float4 q[4] = {};
    unsigned hwid;
    asm volatile("s_getreg_b32 %0, hwreg(HW_REG_HW_ID)" : "=s"(hwid));
    __syncthreads();
    unsigned long long t0 = __builtin_readcyclecounter();
    if (role == 0) mfma_chain(acc, a, b);
    else if (role == 1) valu_chain(r, 1.0001f, valu_iters);
    else if (role == 2) lds_chain(q, (threadIdx.x & 63) * 16, lds_iters);
    else if (role == 4) mfma_dep_chain(acc[0], a, b);
    else if (role == 13) fused_like_chain(acc, r, (threadIdx.x & 63) * 16, b, 128);
    else if (role == 9) valu_kind_chain<0>(r, 1.0001f, valu_iters);
    else if (role == 10) valu_kind_chain<1>(r, 1.0001f, valu_iters);
    else if (role == 11) valu_kind_chain<2>(r, 1.0001f, valu_iters);
    else if (role == 12) valu_kind_chain<3>(r, 1.0001f, valu_iters);
    else if (role == 5) mfma_like_chain(acc, (threadIdx.x & 63) * 16, b);
    else if (role == 6) mfma_alt_chain(acc, (threadIdx.x & 63) * 16, b);
    else if (role == 7) xform_like_chain(r, q, (threadIdx.x & 63) * 16, g, 40, false);
    else if (role == 8) xform_like_chain(r, q, (threadIdx.x & 63) * 16, g, 40, true);
    unsigned long long t1 = __builtin_readcyclecounter();
    if ((threadIdx.x & 63) == 0 && blockIdx.x == 0) { out[wave] = t1 - t0; out[8 + wave] = hwid; }
    float s = r[0] + r[1] + r[2] + r[3] + q[0].x + q[1].y + q[2].z + q[3].w;
    for (int i = 0; i < 4; ++i) for (int j = 0; j < 16; ++j) s += acc[i][j];
    if (s == 12345.678f) out[63] = 1;
}

static const char* NAMES[] = {"MFMA", "VALU", "LDS", "idle", "MFMA1", "Mlike", "Malt", "Tlike", "Tvmem", "pkadd", "fmamix", "cvtrtz", "max3", "fused"};
static const float4* G;
static void run(unsigned long long* d, int ra, int rb, int by_half, int vi, int li)
{
    hipLaunchKernelGGL(k, dim3(256), dim3(512), 0, 0, d, G, ra, rb, by_half, vi, li, 1.0f);
    hipDeviceSynchronize();
    unsigned long long h[16]; hipMemcpy(h, d, sizeof h, hipMemcpyDeviceToHost);
    printf("%-5s | %-5s by %-6s:", NAMES[ra], NAMES[rb], by_half ? "half" : "parity");
    for (int w = 0; w < 8; ++w) printf(" w%d[simd %llu] %6llu", w, (h[8 + w] >> 4) & 3, h[w]);
    printf("\n");
}
int main()
{
    unsigned long long* d; hipMalloc(&d, 64 * 8);
    float4* g; hipMalloc(&g, 1 << 20); hipMemset(g, 0, 1 << 20); G = g;
    const int vi = 24, li = 96;          // 24*256 = 6144 v_fma per wave; 96*32 = 3072 ds_read_b128 per wave; 128*32 = 4096 MFMAs per wave
    run(d, 0, 3, 1, vi, li);             // MFMA alone, one wave per SIMD (waves 0-3)
    run(d, 0, 0, 1, vi, li);             // MFMA on all 8 waves (two per SIMD)
    run(d, 1, 3, 1, vi, li);             // VALU alone, one wave per SIMD
    run(d, 1, 1, 1, vi, li);             // VALU on all 8
    run(d, 2, 3, 1, vi, li);             // LDS alone, 4 waves
    run(d, 2, 2, 1, vi, li);             // LDS on all 8
    for (int by_half = 0; by_half < 2; ++by_half) {
        run(d, 0, 1, by_half, vi, li);   // MFMA | VALU
        run(d, 0, 2, by_half, vi, li);   // MFMA | LDS
        run(d, 1, 2, by_half, vi, li);   // VALU | LDS
    }
    // closer to the Winograd kernel: the multiplying wave as compiled (LDS operand reads + dependent MFMA triples) against the transforming wave
    for (int m : {4, 5, 6}) run(d, m, 3, 1, vi, li);
    for (int t : {7, 8}) run(d, t, 3, 1, vi, li);
    for (int m : {0, 4, 5, 6}) for (int t : {1, 7, 8}) run(d, m, t, 1, vi, li);
    run(d, 13, 3, 1, vi, li);            // one wave per SIMD, both jobs in one stream: 4 096 MFMAs, each followed by 7 vector ops
    run(d, 13, 13, 1, vi, li);           // two such waves per SIMD
    for (int t : {9, 10, 11, 12}) { run(d, t, 3, 1, vi, li); run(d, 0, t, 1, vi, li); run(d, 5, t, 1, vi, li); }
    return 0;
}
