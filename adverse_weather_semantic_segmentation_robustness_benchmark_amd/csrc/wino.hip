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
// 32*mt.. and couts 32*nt.., all 16 positions -> 16 accumulator tiles of 32x32 (256 AGPRs per
// lane, one wave per SIMD).  Input channels stream in chunks of 8 (four k-steps):
//   * the chunk's raw 18x18-pixel patch is LDS-DMA'd once (every pixel fetched once, not once per
//     overlapping tile) two chunks ahead, through a buffer descriptor: per-lane 32-bit offsets, a
//     scalar chunk offset, and the hardware bounds check supplies the zero padding;
//   * each thread turns one (tile, 2 adjacent channels) of it into V with ds_read_b64, packed
//     v_pk_add_f32 (kept packed by inline asm) and ds_write_b64, a quarter of that work riding
//     behind each group of 16 MFMAs of the previous chunk;
//   * A fragments come from V by ds_read_b128 (four k-steps per read), B fragments straight from L2
//     through a buffer descriptor with scalar offsets (the host stores U as the fragment image),
//     both one position group ahead of their MFMAs.
// fp32-input MFMA runs at the vector rate and vector instructions do NOT hide behind it — the loop
// was tuned by removing them (DESIGN.md §5a has the measurements and the ablation builds).
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
constexpr int PW = 2 * TB + 2;                // 18 x 18 input pixels feed the block's 8 x 8 tiles
constexpr int PRS = PW;                       // patch row stride in 16-byte slots (dense)
constexpr int PH1 = PW * PW;                  // slot offset of the second channel quad
constexpr int P_UNITS = 11 * 64;              // slots per patch buffer (2 * 324 = 648, whole wave instructions)
constexpr int P_FLOATS = P_UNITS * 4;         // 11 KB per buffer
constexpr int LDS_FLOATS = 2 * V_FLOATS + 2 * P_FLOATS;   // 92 KB: transformed input and the raw patch, double-buffered

struct wino_args {
    const float* x; const float* U; const float* shift; const float* residual; const float* w2; const float* b2;
    float* out;
    int H, W, Cin, Cout, dil, act, nbx, nby, ngroups, batch;
};

__device__ __forceinline__ float act_apply(float v, int act)
{
    if (act == AWSEG_ACT_RELU) return v > 0.f ? v : 0.f;
    return v;
}

// LDS-DMA of 16 bytes per lane through a buffer descriptor (buffer_load_dwordx4 ... lds): lane l's bytes land at
// lds_base + 16*l; per-lane 32-bit byte offset + scalar byte offset, bounds-checked by the hardware (an out-of-range
// lane reads zeros) — no 64-bit address arithmetic and no zero row for padding pixels.  Issued through inline asm on
// purpose: the LDS-DMA builtins make hipcc treat every later ds_read as possibly aliasing the DMA's LDS write and
// wait vmcnt(0) right after the first MFMA of a step, which serialises the copy with the math.  The kernel orders
// the copy itself: a vmcnt wait before the barrier that ends the step in which the copy was issued.
// (Inline asm is only used where its operands come from / go to LDS and plain VALU results: hipcc pads MFMA-result
// and transcendental-result use hazards only for instructions it emitted itself — see attn.hip.)
__device__ __forceinline__ void bufdma16(__amdgpu_buffer_rsrc_t rsrc, uint32_t voff, uint32_t soff, uint32_t lds_base)
{
    asm volatile("s_mov_b32 m0, %3\n\ts_nop 0\n\tbuffer_load_dwordx4 %0, %1, %2 offen lds" : : "v"(voff), "s"(rsrc), "s"(soff), "s"(lds_base) : "m0");
}
__device__ __forceinline__ void glds_wait() { asm volatile("s_waitcnt vmcnt(0)" : : : "memory"); }
// the four youngest vector-memory operations (the next chunk's first weight fragments, issued after
// the step's LDS-DMAs) may stay in flight across the barrier
__device__ __forceinline__ void glds_wait_keep4() { asm volatile("s_waitcnt vmcnt(4)" : : : "memory"); }
__device__ __forceinline__ uint32_t lds_addr(const float* p)
{
    return (uint32_t)(uintptr_t)(__attribute__((address_space(3))) const void*)p;
}

// two channels per lane as a 2-vector: a - b / a + b compile to ONE v_pk_add_f32.  fp32 MFMA runs at the
// vector rate — VALU instructions do not hide behind it, every one removed from the loop is MFMA time back.
typedef float v2f __attribute__((ext_vector_type(2)));
// (hipcc's pre-emit peephole splits v_pk_add_f32 back into two v_add_f32 when it sits behind an MFMA, assuming
// the MFMA hides them; inline asm keeps the packed form.)
__device__ __forceinline__ v2f pk_add(v2f a, v2f b)
{
    v2f d;
    asm("v_pk_add_f32 %0, %1, %2" : "=v"(d) : "v"(a), "v"(b));
    return d;
}
__device__ __forceinline__ v2f pk_sub(v2f a, v2f b)
{
    v2f d;
    asm("v_pk_add_f32 %0, %1, %2 neg_lo:[0,1] neg_hi:[0,1]" : "=v"(d) : "v"(a), "v"(b));
    return d;
}

// Images.  sV[p][hk][tile][s] in LDS, U[chunk][p][hk][cout][s] in global memory: k-step s pairs the chunk's input
// channels c(s,0), c(s,1) with c(s,hk) = 4*(s>>1) + 2*hk + (s&1) (lane l of the 32x32x2 MFMA takes
// half hk = l>>5), so a lane's four k-steps of one position are ONE conflict-free ds_read_b128 and
// a transform thread's two channels are adjacent in memory.  Raw patch: 16-byte slots (4 channels of one
// pixel), slot = quad*PH1 + py*PRS + (px ^ ((py>>1)&1)), dense: the column flip on every other row pair
// keeps the transform's ds_read_b64 (8 of a slot's 16 bytes per lane) at its 2-way floor for every tap.
//
// V = B^T d B of one (tile, 2 adjacent channels) in two stages so the work can ride between the MFMA groups
// of a step: rows first (t = B^T d), then one output row i
// of (t B) -> positions 4i..4i+3, stored as float2 (k-steps 2*cp, 2*cp+1 of half hk).
struct patch_ptrs { const float* e_lo; const float* o_lo; const float* e_hi; const float* o_hi; };

__device__ __forceinline__ void patch_read(const patch_ptrs& pp, v2f (&r)[16])
{
#pragma unroll
    for (int i = 0; i < 4; ++i)
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            const float* base = i < 2 ? ((j & 1) ? pp.o_lo : pp.e_lo) : ((j & 1) ? pp.o_hi : pp.e_hi);
            r[i * 4 + j] = *reinterpret_cast<const v2f*>(base + (i * PRS + j) * 4);
        }
}
__device__ __forceinline__ void xform_rows(const v2f (&r)[16], v2f (&t)[16])
{
#pragma unroll
    for (int j = 0; j < 4; ++j) {
        t[0 * 4 + j] = pk_sub(r[0 * 4 + j], r[2 * 4 + j]);
        t[1 * 4 + j] = pk_add(r[1 * 4 + j], r[2 * 4 + j]);
        t[2 * 4 + j] = pk_sub(r[2 * 4 + j], r[1 * 4 + j]);
        t[3 * 4 + j] = pk_sub(r[1 * 4 + j], r[3 * 4 + j]);
    }
}
__device__ __forceinline__ void xform_cols_store(const v2f (&t)[16], int i, float* __restrict__ d)
{
    v2f v[4];
    v[0] = pk_sub(t[i * 4 + 0], t[i * 4 + 2]);
    v[1] = pk_add(t[i * 4 + 1], t[i * 4 + 2]);
    v[2] = pk_sub(t[i * 4 + 2], t[i * 4 + 1]);
    v[3] = pk_sub(t[i * 4 + 1], t[i * 4 + 3]);
#pragma unroll
    for (int j = 0; j < 4; ++j) *reinterpret_cast<v2f*>(d + (i * 4 + j) * (KC * NTILE)) = v[j];
}

template <int MODE>   // 0 FULL, 1 HEAD1
__global__ __launch_bounds__(WT, 1)
void conv3x3_wino_kernel(wino_args a)
{
    extern __shared__ float smem[];
    float* sV = smem;                        // [2][16][KC][NTILE]
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    // 1-D grid, XCD-aware: consecutive block ids alternate over the 8 XCDs (one L2 each).  Spatial tile t goes to
    // XCD t % 8 and its cout groups run back to back there, so the groups share the tile's input patch in that L2
    // instead of each fetching it again.
    const int xcd = blockIdx.x & 7, jj = blockIdx.x >> 3;
    const int ng = jj % a.ngroups, t = (jj / a.ngroups) * 8 + xcd;
    const int gx = a.nbx * a.dil, gy = a.nby * a.dil;
    if (t >= gx * gy * a.batch) return;
    const int b = t / (gx * gy), txy = t - b * (gx * gy), tyy = txy / gx, txx = txy - tyy * gx;
    const int bx = txx % a.nbx, rx = txx / a.nbx;
    const int by = tyy % a.nby, ry = tyy / a.nby;
    const int Hs = (a.H - ry + a.dil - 1) / a.dil, Ws = (a.W - rx + a.dil - 1) / a.dil;   // sub-grid extent of this residue
    if (by * 2 * TB >= Hs || bx * 2 * TB >= Ws) return;
    const float* xb = a.x + (int64_t)b * a.H * a.W * a.Cin;
    const int n0 = ng * NB;
    const int nchunks = a.Cin / KC;          // even (Cin % 16 == 0)

    float* sP = smem + 2 * V_FLOATS;         // [2][P_UNITS] raw input patch of a chunk
    // ---- raw patch: the 18 x 18 pixels x KC channels the block's tiles are cut from, LDS-DMA'd in
    // 16-byte units.  Fetching every pixel once (instead of once per overlapping tile straight into
    // registers) keeps the texture-address path off the critical path: a scattered per-lane load costs
    // it one request per lane.  Pixels outside the image fetch a clamped address; the mask zeroes them.
    const __amdgpu_buffer_rsrc_t x_rsrc = __builtin_amdgcn_make_buffer_rsrc((void*)xb, 0, (int)((size_t)a.H * a.W * a.Cin * 4), 0x00020000);
    uint32_t pvoff[3];
#pragma unroll
    for (int i = 0; i < 3; ++i) {
        const int q = (wave + 4 * i) * 64 + lane;                   // destination slot
        const int h = q >= PH1 ? 1 : 0, r = q - h * PH1;
        int py = r / PRS, pxs = r - py * PRS;
        if (py >= PW) py = PW - 1;                                  // padding slots re-fetch a valid pixel
        const int px = pxs ^ ((py >> 1) & 1);
        const int sy = by * 2 * TB - 1 + py, sx = bx * 2 * TB - 1 + px;
        const int y = ry + a.dil * sy, x = rx + a.dil * sx;
        const bool ok = sy >= 0 && sx >= 0 && y < a.H && x < a.W;
        pvoff[i] = ok ? (uint32_t)(((y * a.W + x) * a.Cin + h * 4) * 4) : 0x80000000u;   // out of range -> zeros
    }
    const int n_pinstr = wave < 3 ? 3 : 2;                          // 11 wave instructions over 4 waves
    const uint32_t p_lds = __builtin_amdgcn_readfirstlane(lds_addr(sP) + wave * 1024);       // scalar LDS byte address of this wave's first slot run
    auto glds_patch = [&](int chunk, int buf) {
        const uint32_t soff = (uint32_t)((chunk < nchunks ? chunk : nchunks - 1) * KC * 4);
#pragma unroll
        for (int i = 0; i < 3; ++i)
            if (i < n_pinstr) bufdma16(x_rsrc, pvoff[i], soff, p_lds + (uint32_t)(buf * P_FLOATS * 4 + i * 4096));
    };
    // transform item of this thread: half hk = wave & 1 (channels 4*cp + 2*hk, +1 = k-steps 2cp, 2cp+1), tile =
    // 32*(wave>>1) + lane/2, cp = lane & 1; validity of its 4 x 4 pixels
    const int xhk = wave & 1, xcp = lane & 1, xtile = (wave >> 1) * 32 + (lane >> 1);
    const int xty = xtile >> 3, xtx = xtile & 7;
    const int xf0 = xty & 1, xf1 = xf0 ^ 1;
    const int pbase = (xcp * PH1 + 2 * xty * PRS + 2 * xtx) * 4 + 2 * xhk;
    const int pe_lo = pbase + xf0 * 4, po_lo = pbase - xf0 * 4, pe_hi = pbase + xf1 * 4, po_hi = pbase - xf1 * 4;
    const int vdoff = xhk * (4 * NTILE) + xtile * 4 + 2 * xcp;     // sV[p][hk][tile][2cp..2cp+1]
    // ---- weights: the host lays U out as the B fragments, [chunk][p][hk][cout][s]: a lane's four
    // k-steps of one position are 16 contiguous bytes, a wave's two halves read two 512-byte runs.
    // They go straight from L2 to registers one position group ahead of their MFMAs (an LDS-DMA
    // stage for them cost more issue time than it saved: ~130 cycles per 1 KiB instruction).
    // Buffer addressing: descriptor + ONE per-lane byte offset (VGPR) + a scalar byte offset per load (chunk,
    // position, cout block: scalar adds) — no 64-bit vector address arithmetic beside the MFMAs.
    const __amdgpu_buffer_rsrc_t u_rsrc = __builtin_amdgcn_make_buffer_rsrc((void*)a.U, 0, 0x7fffffff, 0x00020000);
    const uint32_t ulane = (uint32_t)(((lane >> 5) * a.Cout + (lane & 31)) * 16);
    const uint32_t ub = (uint32_t)(n0 + (__builtin_amdgcn_readfirstlane(wave) & 1) * 32) * 16;      // bytes, wave-uniform
    const uint32_t u_pos = (uint32_t)2 * a.Cout * 16, u_chunk = (uint32_t)32 * a.Cout * 16;         // bytes

    f32x16 acc[16];                       // first written by the first chunk's MFMAs (zero C operand: no 256-register clear)

    const int mt = wave >> 1, nt = wave & 1, hk = lane >> 5, li = lane & 31;
    const int aoff = hk * (4 * NTILE) + (mt * 32 + li) * 4;

    glds_patch(0, 0);
    glds_patch(1, 1);
    glds_wait();
    __syncthreads();
    {
        v2f r[16], t[16];
        const patch_ptrs pp = {sP + pe_lo, sP + po_lo, sP + pe_hi, sP + po_hi};
        patch_read(pp, r);
        xform_rows(r, t);
#pragma unroll
        for (int i = 0; i < 4; ++i) xform_cols_store(t, i, sV + vdoff);
    }
    __syncthreads();

    // One step = the MFMAs of chunk c on buffers BUF while chunk c+1 is staged into the other ones:
    // its patch (landed during the previous step) is transformed into V[BUF^1], its weights are
    // LDS-DMA'd into U[BUF^1], and the patch buffer the previous step's transform freed is refilled
    // with chunk c+2.  No conditionals: past the end the loads re-read the last chunk into buffers
    // nobody consumes.  The step is software-pipelined by hand in four groups of four positions
    // (operands of group g+1 are read while the MFMAs of g issue; a quarter of the transform rides
    // in each group).
#define WINO_LOAD_A(G, AV)                                                                     \
    _Pragma("unroll") for (int q = 0; q < 4; ++q)                                              \
        AV[q] = *reinterpret_cast<const float4*>(pa + (4 * (G) + q) * (KC * NTILE));
#define WINO_LOAD_B(UP, G, BV)                                                                 \
    _Pragma("unroll") for (int q = 0; q < 4; ++q)                                              \
        BV[q] = __builtin_bit_cast(float4, __builtin_amdgcn_raw_buffer_load_b128(u_rsrc, ulane, (UP) + (4 * (G) + q) * u_pos, 0));
    // FULL keeps tiles on the accumulator rows and couts on its columns (a lane = one cout: coalesced
    // NHWC stores).  HEAD1 swaps the operands -> couts on the rows, so the 1x1 convolution's sum over
    // the 64 couts is 15 in-register adds + one cross-half shuffle + one cross-wave LDS add per pixel
    // instead of a 32-lane butterfly per pixel.
#define WINO_MM(X, Y, CACC) (MODE == 1 ? __builtin_amdgcn_mfma_f32_32x32x2f32(Y, X, CACC, 0, 0, 0)   \
                                       : __builtin_amdgcn_mfma_f32_32x32x2f32(X, Y, CACC, 0, 0, 0))
#define WINO_MFMAS(G, AV, BV)                                                                  \
    _Pragma("unroll") for (int q = 0; q < 4; ++q) acc[4 * (G) + q] = WINO_MM(AV[q].x, BV[q].x, (FIRST ? kZero16 : acc[4 * (G) + q])); \
    _Pragma("unroll") for (int q = 0; q < 4; ++q) acc[4 * (G) + q] = WINO_MM(AV[q].y, BV[q].y, acc[4 * (G) + q]); \
    _Pragma("unroll") for (int q = 0; q < 4; ++q) acc[4 * (G) + q] = WINO_MM(AV[q].z, BV[q].z, acc[4 * (G) + q]); \
    _Pragma("unroll") for (int q = 0; q < 4; ++q) acc[4 * (G) + q] = WINO_MM(AV[q].w, BV[q].w, acc[4 * (G) + q]);
    // Instruction-mix recipes for the scheduler: each MFMA is followed by a few of the companion
    // instructions of its group, so staging work issues in the shadow of the 64-cycle MFMAs instead of
    // in a clump that leaves the matrix core idle.  (mask 0x008 MFMA, 0x002 VALU, 0x020 VMEM read,
    // 0x100 DS read, 0x200 DS write)
#define WINO_MIX(NVALU, NVMEM, NDSR, NDSW)                                                     \
    _Pragma("unroll") for (int m = 0; m < 16; ++m) {                                           \
        __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);                                     \
        if (NVMEM) __builtin_amdgcn_sched_group_barrier(0x020, NVMEM, 0);                      \
        if (NDSR) __builtin_amdgcn_sched_group_barrier(0x100, NDSR, 0);                        \
        if (NVALU) __builtin_amdgcn_sched_group_barrier(0x002, NVALU, 0);                      \
        if (NDSW) __builtin_amdgcn_sched_group_barrier(0x200, NDSW, 0);                        \
    }
    // B0 enters holding the weights of (chunk C, group 0) and leaves holding those of (C+1, group 0).
#define WINO_STEP(C, BUF, IS_FIRST)                                                            \
    do {                                                                                       \
        constexpr bool FIRST = IS_FIRST;                                                       \
        const float* pa = sV + (BUF) * V_FLOATS + aoff;                                        \
        float* vd = sV + ((BUF) ^ 1) * V_FLOATS + vdoff;                                       \
        const float* pn = sP + ((BUF) ^ 1) * P_FLOATS;                                         \
        const patch_ptrs pp = {pn + pe_lo, pn + po_lo, pn + pe_hi, pn + po_hi};                \
        const uint32_t uc = ub + (uint32_t)(C) * u_chunk;                                      \
        const uint32_t un = ub + (uint32_t)((C) + 1 < nchunks ? (C) + 1 : (C)) * u_chunk;      \
        float4 a0[4], a1[4];                                                                   \
        v2f r[16], t[16];                                                                      \
        WINO_LOAD_A(0, a0)                                                                     \
        __builtin_amdgcn_sched_barrier(0);                                                     \
        /* group 0: MFMAs + LDS-DMA issue + operands of group 1 + raw patch reads */            \
        glds_patch((C) + 2, (BUF));                                                            \
        WINO_LOAD_B(uc, 1, b1)                                                                 \
        WINO_LOAD_A(1, a1)                                                                     \
        patch_read(pp, r);                                                                     \
        WINO_MFMAS(0, a0, b0)                                                                  \
        WINO_MIX(2, 1, 2, 0)                                                                   \
        __builtin_amdgcn_sched_barrier(0);                                                     \
        /* group 1: MFMAs + row transform + operands of group 2 */                              \
        xform_rows(r, t);                                                               \
        WINO_LOAD_B(uc, 2, b0)                                                                 \
        WINO_LOAD_A(2, a0)                                                                     \
        WINO_MFMAS(1, a1, b1)                                                                  \
        WINO_MIX(4, 1, 1, 0)                                                                   \
        __builtin_amdgcn_sched_barrier(0);                                                     \
        /* group 2: MFMAs + half of the column transform + operands of group 3 */               \
        xform_cols_store(t, 0, vd);                                                            \
        xform_cols_store(t, 1, vd);                                                            \
        WINO_LOAD_B(uc, 3, b1)                                                                 \
        WINO_LOAD_A(3, a1)                                                                     \
        WINO_MFMAS(2, a0, b0)                                                                  \
        WINO_MIX(2, 1, 1, 1)                                                                   \
        __builtin_amdgcn_sched_barrier(0);                                                     \
        /* group 3: MFMAs + the other half of the column transform + next chunk's first weights */ \
        xform_cols_store(t, 2, vd);                                                            \
        xform_cols_store(t, 3, vd);                                                            \
        WINO_LOAD_B(un, 0, b0)                                                                 \
        WINO_MFMAS(3, a1, b1)                                                                  \
        WINO_MIX(2, 1, 0, 1)                                                                   \
        glds_wait_keep4();                                                                     \
        __syncthreads();                                                                       \
    } while (0)
    float4 b0[4], b1[4];
    WINO_LOAD_B(ub, 0, b0)
    static_assert(KC == 8, "the step below is written for four k-steps per chunk");
    const f32x16 kZero16 = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
    WINO_STEP(0, 0, true);
    WINO_STEP(1, 1, false);
    for (int c = 2; c < nchunks; c += 2) {
        WINO_STEP(c, 0, false);
        WINO_STEP(c + 1, 1, false);
    }
#undef WINO_LOAD_A
#undef WINO_LOAD_B
#undef WINO_MFMAS
#undef WINO_MM
#undef WINO_STEP
#undef WINO_MIX

    // ---- output transform + epilogue.  acc[p][r] is element (row (r&3) + 8*(r>>2) + 4*hk, column li) of
    // this wave's 32x32 block of M_p: FULL rows = tiles of m-tile mt, columns = couts of n-tile nt;
    // HEAD1 rows = couts of nt, columns = tiles of mt.
    float* red = smem;                                   // HEAD1: [2][NTILE][4] partial sums (after the last sync)
    // Two accumulator rows at a time: rows r, r+1 of one position sit in adjacent registers, so the inverse
    // transform (24 adds per row) runs as packed 2-vector adds.
    if (MODE == 0) {
        // Stores (and residual loads) go through buffer descriptors: the pixel part of the address is a SCALAR
        // per (row, aa, bb) — tile row / column come from the compile-time accumulator row and the wave-uniform
        // m-tile — and the lane part (cout, and the 4-tile column shift of the upper half-wave) is one VGPR per
        // (r & 3, bb) with the x-bound folded in as an out-of-range VECTOR offset (the part the hardware range check is defined
        // on); the y-bound is a scalar branch.  No per-store
        // vector address arithmetic.
        const int n = n0 + nt * 32 + li;
        const float sh = a.shift[n];
        const size_t img = (size_t)a.H * a.W * a.Cout;
        const __amdgpu_buffer_rsrc_t o_rsrc = __builtin_amdgcn_make_buffer_rsrc((void*)(a.out + (size_t)b * img), 0, (int)(img * 4), 0x00020000);
        const __amdgpu_buffer_rsrc_t r_rsrc = __builtin_amdgcn_make_buffer_rsrc((void*)(a.residual ? a.residual + (size_t)b * img : a.out), 0, (int)(img * 4), 0x00020000);
        const bool has_res = a.residual != nullptr;
        const int mt_u = __builtin_amdgcn_readfirstlane(mt);
        const uint32_t kOob = 0x80000000u;
        uint32_t vsel[4][2];                               // lane byte offset for column (r & 3, bb), or out of range
#pragma unroll
        for (int c = 0; c < 4; ++c)
#pragma unroll
            for (int bb = 0; bb < 2; ++bb) {
                const int xs = rx + a.dil * (bx * 2 * TB + 2 * c + bb);          // scalar part of x
                const int xl = a.dil * 8 * hk;                                   // the upper half-wave sits 4 tiles to the right
                vsel[c][bb] = (xs + xl < a.W) ? (uint32_t)((xl * a.Cout + n) * 4) : kOob;
            }
#pragma unroll
        for (int r = 0; r < 16; r += 2) {
            v2f m[4][4];
#pragma unroll
            for (int p = 0; p < 16; ++p) m[p >> 2][p & 3] = v2f{acc[p][r], acc[p][r + 1]};
            v2f t0[4], t1[4];
#pragma unroll
            for (int j = 0; j < 4; ++j) { t0[j] = m[0][j] + m[1][j] + m[2][j]; t1[j] = m[1][j] - m[2][j] - m[3][j]; }
            v2f y[2][2];
            y[0][0] = t0[0] + t0[1] + t0[2]; y[0][1] = t0[1] - t0[2] - t0[3];
            y[1][0] = t1[0] + t1[1] + t1[2]; y[1][1] = t1[1] - t1[2] - t1[3];
#pragma unroll
            for (int e = 0; e < 2; ++e) {
                const int rr = r + e;
                const int ty = mt_u * 4 + (rr >> 2), c = rr & 3;                 // tile row (scalar), tile column within the half
#pragma unroll
                for (int aa = 0; aa < 2; ++aa) {
                    const int yy = ry + a.dil * (by * 2 * TB + 2 * ty + aa);
                    if (yy >= a.H) continue;                                     // wave-uniform: a scalar branch
#pragma unroll
                    for (int bb = 0; bb < 2; ++bb) {
                        const int xs = rx + a.dil * (bx * 2 * TB + 2 * c + bb);
                        const uint32_t soff = (uint32_t)((yy * a.W + xs) * a.Cout * 4);   // always in range; x-bound is in vsel
                        float v = y[aa][bb][e] + sh;
                        if (has_res) v += __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(r_rsrc, vsel[c][bb], soff, 0));
                        __builtin_amdgcn_raw_buffer_store_b32(__builtin_bit_cast(uint32_t, act_apply(v, a.act)), o_rsrc, vsel[c][bb], soff, 0);
                    }
                }
            }
        }
    } else {
        // two accumulator rows (= two consecutive couts) at a time: adjacent registers -> packed adds
        v2f z[4] = {v2f{0.f, 0.f}, v2f{0.f, 0.f}, v2f{0.f, 0.f}, v2f{0.f, 0.f}};
#pragma unroll
        for (int r4 = 0; r4 < 4; ++r4) {
            const int c4 = n0 + nt * 32 + 8 * r4 + 4 * hk;                      // rows 4*r4 .. 4*r4+3 are four consecutive couts
            const float4 sh4 = *reinterpret_cast<const float4*>(a.shift + c4);
            const float4 w4 = *reinterpret_cast<const float4*>(a.w2 + c4);
            const v2f shv[2] = {v2f{sh4.x, sh4.y}, v2f{sh4.z, sh4.w}}, wv[2] = {v2f{w4.x, w4.y}, v2f{w4.z, w4.w}};
#pragma unroll
            for (int rr = 0; rr < 2; ++rr) {
                const int r = r4 * 4 + 2 * rr;
                v2f m[4][4];
#pragma unroll
                for (int p = 0; p < 16; ++p) m[p >> 2][p & 3] = v2f{acc[p][r], acc[p][r + 1]};
                v2f t0[4], t1[4];
#pragma unroll
                for (int j = 0; j < 4; ++j) { t0[j] = m[0][j] + m[1][j] + m[2][j]; t1[j] = m[1][j] - m[2][j] - m[3][j]; }
                const v2f yv[4] = {t0[0] + t0[1] + t0[2], t0[1] - t0[2] - t0[3], t1[0] + t1[1] + t1[2], t1[1] - t1[2] - t1[3]};
#pragma unroll
                for (int q = 0; q < 4; ++q) {
                    const v2f v = yv[q] + shv[rr];
                    z[q] = z[q] + __builtin_elementwise_max(v, v2f{0.f, 0.f}) * wv[rr];
                }
            }
        }
        float zs[4];
#pragma unroll
        for (int q = 0; q < 4; ++q) {
            zs[q] = z[q][0] + z[q][1];
            zs[q] += __shfl_xor(zs[q], 32, 64);                                 // the other half holds rows +4 of the same tile
        }
        if (hk == 0) *reinterpret_cast<float4*>(red + (nt * NTILE + mt * 32 + li) * 4) = make_float4(zs[0], zs[1], zs[2], zs[3]);
        __syncthreads();
        const int tile = tid >> 2, q = tid & 3;
        const int uy = by * 2 * TB + 2 * (tile >> 3) + (q >> 1), ux = bx * 2 * TB + 2 * (tile & 7) + (q & 1);
        const int yy = ry + a.dil * uy, xx = rx + a.dil * ux;
        if (yy < a.H && xx < a.W) {
            const float zz = red[tile * 4 + q] + red[(NTILE + tile) * 4 + q] + a.b2[0];
            a.out[((int64_t)b * a.H + yy) * a.W + xx] = 1.0f / (1.0f + expf(-zz));
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
    const int64_t tiles = (int64_t)a.nbx * a.dil * a.nby * a.dil * batch;
    const int64_t nblocks = ((tiles + 7) / 8) * a.ngroups * 8;
    if (nblocks >= ((int64_t)1 << 31)) return AWSEG_ERANGE;
    hipLaunchKernelGGL(kern, dim3((unsigned)nblocks), dim3(WT), lds, s, a);
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
    a.nbx = (ws + 2 * TB - 1) / (2 * TB); a.nby = (hs + 2 * TB - 1) / (2 * TB); a.ngroups = cout / NB; a.batch = batch;
    return w2 ? launch_wino<1>(a, batch, awseg_s(stream)) : launch_wino<0>(a, batch, awseg_s(stream));
}
