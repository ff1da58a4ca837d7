// wino_split.hip — the operator of wino.hip (3x3 stride-1 "same" convolution as Winograd F(2x2,3x3), NHWC, fused
// epilogues; PKG/models/model.py:42-52, :219-221 and the ResNet bottleneck 3x3s behind :349) with its 16 GEMMs on the
// f16 matrix cores and SPLIT float32 operands.  wino.hip is bound by the fp32-input MFMA itself (64 cycles per
// 32x32x2 product tile: 63 % of a 157 TFLOP/s instruction); v_mfma_f32_32x32x16_f16 retires 16x the
// multiply-adds per cycle, and a float32-grade product is three f16 products:
//     V = Vh + Vl,  Vh = f16(V),  Vl = f16(V - Vh)      (22 significant bits while |V| >= 2^-3; absolute 2^-25 below)
//     V * U = Vh*Uh + Vh*Ul + Vl*Uh                      (+ Vl*Ul, below float32 rounding noise, dropped)
// ONE accumulator per position (256 registers per wave leave no room for a separate correction tile): the low parts are
// used unscaled, which is exact as long as they are normal f16 numbers, so
//   * U = G g G^T is NORMALISED on the host side (device ops, no synchronisation): stored as U * 2^-eu with max|U| in
//     [2^13, 2^14), 2^eu in a trailer — tiny or huge filters keep 22 bits;
//   * activations are taken as they are (BatchNorm/ReLU outputs are O(1)) while every thread tracks max|x| of the
//     patches it transforms; a block whose maximum is >= 2^13 (4 max|x| could leave the f16 range in V = B^T x B) or
//     < 2^-4 (low parts would go subnormal) runs its tile again with x * 2^sx, sx from the observed maximum, and the
//     epilogue multiplies 2^(eu - sx) back.  Same guard as gemm_split.hip / attn.hip.
//
// Block = 8x8 tiles (16x16 outputs) x 64 output channels, 4 waves, input channels in chunks of 16 (one K step of the
// MFMA).  Wave (nt, ph) owns ALL 64 tiles x 32 couts x 8 of the 16 positions (V rows 2ph, 2ph+1): 2 m-tiles x 8
// positions = 256 accumulator registers.  With this split every U fragment is loaded by exactly ONE wave of the block,
// straight from L2 into registers (64 KB per chunk per CU; sharing U through LDS would need 64 KB of LDS per chunk, a
// second wave pair loading the same fragments 128 KB of L2 traffic), and a wave's U fragment of a position feeds two
// m-tiles.  The partial inverse transforms of the two position halves meet through LDS once, in the epilogue.
//   * raw 18x18-pixel patch of a chunk: LDS-DMA, three chunks ahead, ring of three 21 KB slots;
//   * V (f16 high | low parts, [position][tile][16 hi | 16 lo], 16-byte chunks XOR-swizzled by (tile >> 2) & 3:
//     conflict-free ds_read_b128 A fragments) is SINGLE-buffered in 64 KB and refilled in halves behind the MFMAs that
//     consumed them: a chunk is two slots — slot A: MFMAs on V rows {0, 2} while rows {1, 3} of the same chunk are
//     written and the next chunk's patch is read and row-transformed; slot B: MFMAs on rows {1, 3} while rows {0, 2}
//     of the next chunk are written — one barrier per slot;
//   * transform item of a thread = (tile, 4 channels): ds_read_b64 of the patch, packed f32 adds, split, ds_write_b64.
#include "awseg_common.h"

namespace {

typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef _Float16 h8 __attribute__((ext_vector_type(8)));
typedef _Float16 h2 __attribute__((ext_vector_type(2)));
typedef unsigned u32x2 __attribute__((ext_vector_type(2)));
typedef unsigned u32x4 __attribute__((ext_vector_type(4)));
typedef float v2f __attribute__((ext_vector_type(2)));

constexpr int WT = 256;                    // threads per block
constexpr int TB = 8;                      // 8 x 8 tiles per block
constexpr int NTILE = TB * TB;             // 64
constexpr int KC = 16;                     // input channels per chunk = one K step of v_mfma_f32_32x32x16_f16
constexpr int NB = 64;                     // output channels per block
constexpr int PW = 2 * TB + 2;             // 18 x 18 input pixels feed the block's tiles
constexpr int NPIX = PW * PW;              // 324 pixels, 64 bytes (16 channels) each: 4 consecutive DMA lanes fetch one pixel's 64 contiguous bytes
constexpr int P_INSTR = (4 * NPIX + 63) / 64; // 21 wave-wide LDS-DMA instructions per chunk
constexpr int P_BYTES = P_INSTR * 1024;    // 21504
constexpr int P_RING = 3;
constexpr int V_POS = NTILE * 64;          // bytes of one position: 64 tiles x (16 hi + 16 lo halfs)
constexpr int V_BYTES = 16 * V_POS;        // 65536
constexpr int LDS_BYTES = V_BYTES + P_RING * P_BYTES + 64;

struct ws_args {
    const float* x; const uint16_t* U; const float* shift; const float* residual; const float* w2; const float* b2;
    float* out;
    int H, W, Cin, Cout, dil, act, nbx, nby, ngroups, batch;
    int nblocks, tpb;                      // wino8p_kernel: linear block indices in all, tiles per (persistent) block
    int span;                              // wino8_kernel: spatial tiles of one XCD that run the same cout group back to back (block order)
    int64_t u_halfs;                       // halfs of U in front of the trailer {2^eu as float}
    // MODE 2 (wino8p_kernel only): the input map is not read but GENERATED per block from the bilinear forms of
    // awseg_upconv_forms (depthfuse.hip): forms [batch][F4 | F2] float32, fh x fw = the low-resolution grid (H = 32 fh, W = 32 fw)
    const float* forms; int fh, fw;
};

__device__ __forceinline__ float act_apply(float v, int act) { return (act == AWSEG_ACT_RELU) ? (v > 0.f ? v : 0.f) : v; }

// LDS-DMA of 16 bytes per lane through a buffer descriptor (see wino.hip: inline asm on purpose, hardware range check
// supplies the zero padding)
__device__ __forceinline__ void bufdma16(__amdgpu_buffer_rsrc_t rsrc, uint32_t voff, uint32_t soff, uint32_t lds_base)
{
    asm volatile("s_mov_b32 m0, %3\n\ts_nop 0\n\tbuffer_load_dwordx4 %0, %1, %2 offen lds" : : "v"(voff), "s"(rsrc), "s"(soff), "s"(lds_base) : "m0");
}
__device__ __forceinline__ void vm_wait_all() { asm volatile("s_waitcnt vmcnt(0)" : : : "memory"); }
// everything but the six youngest vector-memory operations (the U fragments of the next three positions, fetched last)
__device__ __forceinline__ void vm_wait_keep6() { asm volatile("s_waitcnt vmcnt(6)" : : : "memory"); }
__device__ __forceinline__ uint32_t lds_addr(const void* p)
{
    return (uint32_t)(uintptr_t)(__attribute__((address_space(3))) const void*)p;
}

#ifdef AWSEG_WS_ASM_PK
__device__ __forceinline__ v2f pk_add(v2f a, v2f b) { v2f d; asm("v_pk_add_f32 %0, %1, %2" : "=v"(d) : "v"(a), "v"(b)); return d; }
__device__ __forceinline__ v2f pk_sub(v2f a, v2f b)
{
    v2f d;
    asm("v_pk_add_f32 %0, %1, %2 neg_lo:[0,1] neg_hi:[0,1]" : "=v"(d) : "v"(a), "v"(b));
    return d;
}
#else
// plain vector arithmetic: the scheduler can classify these (sched_group_barrier) and, next to f16 MFMAs, whether hipcc
// keeps them packed matters less than where they are placed
__device__ __forceinline__ v2f pk_add(v2f a, v2f b) { return a + b; }
__device__ __forceinline__ v2f pk_sub(v2f a, v2f b) { return a - b; }
#endif

// (a, b) -> packed f16 high parts and packed f16 low parts (a - f16(a), b - f16(b): exact in float32, then rounded to f16)
// Three instructions: v_cvt_pkrtz_f16_f32, then one mixed-precision FMA per value — v_fma_mixlo/mixhi_f16 computes
// x * 1.0 + (-hi) with the f16 source widened and the sum taken in float32 (exact: hi is x's leading bits), rounds it to f16
// and writes one half of the destination.  (The plain form costs six: two v_cvt_f32_f16, two subtracts, a second pack.)
__device__ __forceinline__ void split_pair(v2f v, unsigned& hi, unsigned& lo)
{
    hi = __builtin_bit_cast(unsigned, __builtin_amdgcn_cvt_pkrtz(v.x, v.y));
    asm("v_fma_mixlo_f16 %0, %1, 1.0, -%3 op_sel_hi:[0,0,1]\n\t"
        "v_fma_mixhi_f16 %0, %2, 1.0, -%3 op_sel:[0,0,1] op_sel_hi:[0,0,1]"
        : "=&v"(lo) : "v"(v.x), "v"(v.y), "v"(hi));
}

// the same split for values that go STRAIGHT into an MFMA operand register: a vector write needs two wait states before an MFMA
// reads the register, and hipcc pads only instructions it emitted itself (gemm_split3.hip)
__device__ __forceinline__ void split_pair_mfma(float x, float y, unsigned& hi, unsigned& lo)
{
    asm("v_cvt_pkrtz_f16_f32 %0, %2, %3\n\t"
        "v_fma_mixlo_f16 %1, %2, 1.0, -%0 op_sel_hi:[0,0,1]\n\t"
        "v_fma_mixhi_f16 %1, %3, 1.0, -%0 op_sel:[0,0,1] op_sel_hi:[0,0,1]\n\t"
        "s_nop 1"
        : "=&v"(hi), "=&v"(lo) : "v"(x), "v"(y));
}

#ifdef AWSEG_WS_STAMP
// tools/probe_wino_stamps.hip: s_memtime stamps of block 0, wave 0: [slot A work, barrier A, slot B work, barrier B,
// chunks, prologue, epilogue, whole block]
__device__ unsigned long long g_ws_stamp[8];
__device__ unsigned long long g_ws_stamp2[8];                      // slot A in detail: [wait, DMA issue, group 0, group 1, group 2]
#define WS_T(var) const unsigned long long var = __builtin_readcyclecounter()
#else
#define WS_T(var)
#endif

struct awseg_false { static constexpr bool value = false; };
struct awseg_true { static constexpr bool value = true; };
template <int N> struct awseg_int { static constexpr int value = N; };

__device__ __forceinline__ float pow2f(int e) { return __builtin_bit_cast(float, (unsigned)(127 + e) << 23); }   // -126 <= e <= 127

typedef __bf16 bf8 __attribute__((ext_vector_type(8)));
__device__ __forceinline__ unsigned pack_bf16(v2f v)
{
    typedef __bf16 bf2 __attribute__((ext_vector_type(2)));
    return __builtin_bit_cast(unsigned, __builtin_convertvector(v, bf2));
}

// MODE: 0 FULL (NHWC map out), 1 HEAD1 (fused 1x1 + sigmoid, Cout == 64).
// BF16: the same kernel with ONE v_mfma_f32_32x32x16_bf16 per product tile (BASELINE config 5, the bf16 MFMA path): V is
// rounded to bf16 after the float32 input transform, U comes as bf16 in the "high part" slots of the same image (the low
// slots are neither stored nor fetched: half the U traffic), float32 accumulation, transforms and epilogue.  bf16 has
// float32's exponent range: no range guard.
template <int MODE, bool BF16>
__global__ __launch_bounds__(WT, 1)
void wino_split_kernel(ws_args a)
{
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    unsigned char* sV = smem;
    unsigned char* sP = smem + V_BYTES;
    unsigned* sMax = reinterpret_cast<unsigned*>(smem + V_BYTES + P_RING * P_BYTES);
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int hk = lane >> 5, li = lane & 31;
    // 1-D grid, XCD-aware (wino.hip): spatial tile t -> XCD t % 8, its cout groups back to back there
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
    const int nchunks = a.Cin / KC;

    // ---- raw patch DMA, pixel-major: slot = 4 * g + quad (16 bytes = 4 channels), pixel group g = py * 18 + pos with the
    // even columns of a patch row first (pos = px / 2 for even px, 9 + px / 2 for odd px): the four lanes of a group
    // read 64 contiguous bytes of global memory (one request instead of four), and the eight tiles of a tile row — two
    // pixels apart — sit in consecutive 64-byte records, which halves the bank conflicts of the transform's reads.
    // Slots >= 4 * 324 fetch out of range (zeros).
    const __amdgpu_buffer_rsrc_t x_rsrc = __builtin_amdgcn_make_buffer_rsrc((void*)xb, 0, (int)((size_t)a.H * a.W * a.Cin * 4), 0x00020000);
    uint32_t pvoff[6];
#pragma unroll
    for (int j = 0; j < 6; ++j) {
        const int q = (wave + 4 * j) * 64 + lane;
        const int g = q >> 2, h = q & 3;
        const int py = g / PW, pos = g - py * PW;
        const int px = pos < PW / 2 ? 2 * pos : 2 * (pos - PW / 2) + 1;
        const int sy = by * 2 * TB - 1 + py, sx = bx * 2 * TB - 1 + px;
        const int y = ry + a.dil * sy, x = rx + a.dil * sx;
        const bool ok = g < NPIX && sy >= 0 && sx >= 0 && y < a.H && x < a.W;
        pvoff[j] = ok ? (uint32_t)(((y * a.W + x) * a.Cin + h * 4) * 4) : 0x80000000u;
    }
    const int n_pinstr = wave == 0 ? 6 : 5;                         // 21 instructions over 4 waves
    const uint32_t p_lds = __builtin_amdgcn_readfirstlane(lds_addr(sP) + wave * 1024);
    auto glds_patch = [&](int chunk, int slot) {
        const uint32_t soff = (uint32_t)((chunk < nchunks ? chunk : nchunks - 1) * KC * 4);
#pragma unroll
        for (int j = 0; j < 6; ++j)
            if (j < n_pinstr) bufdma16(x_rsrc, pvoff[j], soff, p_lds + (uint32_t)(slot * P_BYTES + j * 4096));
    };

    // ---- transform role: tile xtile, channel quad xq (4 channels = two packed pairs)
    const int xtile = tid >> 2, xq = tid & 3;
    const int xty = xtile >> 3, xtx = xtile & 7;
    const int prd = (2 * xty * PW + xtx) * 64 + xq * 16;            // patch byte offset of the tile's pixel (0,0), this quad
    const int xsw = (xtile >> 2) & 3;
    const int vw_hi = xtile * 64 + (((xq >> 1) ^ xsw) * 16) + (xq & 1) * 8;          // hi chunk = quad >> 1 (channels 0-7 | 8-15)
    const int vw_lo = xtile * 64 + (((2 + (xq >> 1)) ^ xsw) * 16) + (xq & 1) * 8;

    // ---- MFMA role: wave (nt, ph): couts n0 + 32 nt .., positions 8 ph .. 8 ph + 7, both m-tiles
    const int nt = wave & 1, ph = wave >> 1;
    int a_hi[2], a_lo[2];
#pragma unroll
    for (int m = 0; m < 2; ++m) {
        const int tile = m * 32 + li, sw = (tile >> 2) & 3;
        a_hi[m] = tile * 64 + ((hk ^ sw) * 16);
        a_lo[m] = tile * 64 + (((2 + hk) ^ sw) * 16);
    }
    // U: [chunk][position][cout block of 32][hi h0 | hi h1 | lo h0 | lo h1][32 couts][8 halfs]: 2 KB per (chunk, p, cb)
    const int ncb = a.Cout / 32, cb = (n0 >> 5) + nt;
    const __amdgpu_buffer_rsrc_t u_rsrc = __builtin_amdgcn_make_buffer_rsrc((void*)a.U, 0, (int)(a.u_halfs * 2), 0x00020000);
    const uint32_t ulane = (uint32_t)(hk * 512 + li * 16);
    const uint32_t u_p = (uint32_t)ncb * 2048u, u_c = 16u * u_p;
    const uint32_t u_w = (uint32_t)(__builtin_amdgcn_readfirstlane(8 * ph) * (int)u_p + __builtin_amdgcn_readfirstlane(cb) * 2048);
    const float uscale = *reinterpret_cast<const float*>(a.U + a.u_halfs);              // 2^eu

    WS_T(blk0);
    f32x16 acc[8][2];
    float amax = 0.f;
    float xs = 1.0f;                                                 // activation scale of a second pass (2^sx)
    int sx = 0;
    if (tid == 0) sMax[0] = 0u;

    // One pass over the input channels.  SC::value: activations are multiplied by xs (second pass of the range guard);
    // the first pass tracks max|x| instead.
    auto run = [&](auto SC) {
        constexpr bool SCALED = decltype(SC)::value;
        v2f tA[16], tB[16];                                          // B^T d of the item's two channel pairs, alive across a slot boundary
        // ---- transform pieces ----------------------------------------------------------------------------------------
        auto patch_rows = [&](int slot, int e, v2f (&tt)[16]) {
            const unsigned char* pp = sP + slot * P_BYTES + prd + e * 8;
            v2f r[16];
#pragma unroll
            for (int i = 0; i < 4; ++i)
#pragma unroll
                for (int j = 0; j < 4; ++j) r[i * 4 + j] = *reinterpret_cast<const v2f*>(pp + (i * PW + (j & 1) * (PW / 2) + (j >> 1)) * 64);
            if (BF16) {
            } else if (!SCALED) {
#pragma unroll
                for (int i = 0; i < 16; ++i) amax = __builtin_fmaxf(__builtin_fmaxf(amax, __builtin_fabsf(r[i].x)), __builtin_fabsf(r[i].y));
            } else {
#pragma unroll
                for (int i = 0; i < 16; ++i) r[i] = r[i] * v2f{xs, xs};
            }
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                tt[0 * 4 + j] = pk_sub(r[0 * 4 + j], r[2 * 4 + j]);
                tt[1 * 4 + j] = pk_add(r[1 * 4 + j], r[2 * 4 + j]);
                tt[2 * 4 + j] = pk_sub(r[2 * 4 + j], r[1 * 4 + j]);
                tt[3 * 4 + j] = pk_sub(r[1 * 4 + j], r[3 * 4 + j]);
            }
        };
        // row i of (t B) -> positions 4i .. 4i+3, split, stored as {hi pair A, hi pair B} and {lo pair A, lo pair B}
        auto cols_store = [&](int i) {
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                v2f va, vb;
                if (j == 0) { va = pk_sub(tA[i * 4 + 0], tA[i * 4 + 2]); vb = pk_sub(tB[i * 4 + 0], tB[i * 4 + 2]); }
                else if (j == 1) { va = pk_add(tA[i * 4 + 1], tA[i * 4 + 2]); vb = pk_add(tB[i * 4 + 1], tB[i * 4 + 2]); }
                else if (j == 2) { va = pk_sub(tA[i * 4 + 2], tA[i * 4 + 1]); vb = pk_sub(tB[i * 4 + 2], tB[i * 4 + 1]); }
                else { va = pk_sub(tA[i * 4 + 1], tA[i * 4 + 3]); vb = pk_sub(tB[i * 4 + 1], tB[i * 4 + 3]); }
                u32x2 H, L; unsigned h, l;
                if (BF16) {
                    H[0] = pack_bf16(va); H[1] = pack_bf16(vb);
                    *reinterpret_cast<u32x2*>(sV + (i * 4 + j) * V_POS + vw_hi) = H;
                    continue;
                }
                split_pair(va, h, l); H[0] = h; L[0] = l;
                split_pair(vb, h, l); H[1] = h; L[1] = l;
                *reinterpret_cast<u32x2*>(sV + (i * 4 + j) * V_POS + vw_hi) = H;
                *reinterpret_cast<u32x2*>(sV + (i * 4 + j) * V_POS + vw_lo) = L;
            }
        };
        // ---- MFMA pieces -----------------------------------------------------------------------------------------------
        auto u_load = [&](int c, int lp, h8& uh, h8& ul) {
            const uint32_t so = (uint32_t)c * u_c + u_w + (uint32_t)lp * u_p;
            uh = __builtin_bit_cast(h8, __builtin_amdgcn_raw_buffer_load_b128(u_rsrc, ulane, so, 0));
            if (!BF16) ul = __builtin_bit_cast(h8, __builtin_amdgcn_raw_buffer_load_b128(u_rsrc, ulane + 1024u, so, 0));
        };
        auto a_load = [&](int lp, h8 (&vh)[2], h8 (&vl)[2]) {
            const unsigned char* vp = sV + (8 * ph + lp) * V_POS;
#pragma unroll
            for (int m = 0; m < 2; ++m) {
                vh[m] = *reinterpret_cast<const h8*>(vp + a_hi[m]);
                if (!BF16) vl[m] = *reinterpret_cast<const h8*>(vp + a_lo[m]);
            }
        };
        auto mfma6 = [&](int lp, const h8 (&vh)[2], const h8 (&vl)[2], const h8& uh, const h8& ul) {
#pragma unroll
            for (int m = 0; m < 2; ++m) {
                f32x16 z = acc[lp][m];
                if (BF16) {
                    if (MODE == 1) z = __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf8, uh), __builtin_bit_cast(bf8, vh[m]), z, 0, 0, 0);
                    else z = __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf8, vh[m]), __builtin_bit_cast(bf8, uh), z, 0, 0, 0);
                } else if (MODE == 1) {                              // couts on the accumulator rows (in-register sum over couts)
                    z = __builtin_amdgcn_mfma_f32_32x32x16_f16(uh, vh[m], z, 0, 0, 0);
                    z = __builtin_amdgcn_mfma_f32_32x32x16_f16(ul, vh[m], z, 0, 0, 0);
                    z = __builtin_amdgcn_mfma_f32_32x32x16_f16(uh, vl[m], z, 0, 0, 0);
                } else {
                    z = __builtin_amdgcn_mfma_f32_32x32x16_f16(vh[m], uh, z, 0, 0, 0);
                    z = __builtin_amdgcn_mfma_f32_32x32x16_f16(vh[m], ul, z, 0, 0, 0);
                    z = __builtin_amdgcn_mfma_f32_32x32x16_f16(vl[m], uh, z, 0, 0, 0);
                }
                acc[lp][m] = z;
            }
        };
        // instruction-mix recipe for one position group: each of the 6 MFMAs is followed by a share of the group's
        // companion work (mask 0x008 MFMA, 0x002 VALU, 0x020 VMEM read, 0x100 DS read, 0x200 DS write)
#define WS_MIX(NVALU, NDSR, NDSW)                                                                \
        _Pragma("unroll") for (int mm = 0; mm < 6; ++mm) {                                       \
            __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);                                   \
            if (mm < 2) __builtin_amdgcn_sched_group_barrier(0x020, 1, 0);                       \
            if (NDSR) __builtin_amdgcn_sched_group_barrier(0x100, NDSR, 0);                      \
            if (NVALU) __builtin_amdgcn_sched_group_barrier(0x002, NVALU, 0);                    \
            if (NDSW) __builtin_amdgcn_sched_group_barrier(0x200, NDSW, 0);                      \
        }

        // ---- prologue ----------------------------------------------------------------------------------------------------
#pragma unroll
        for (int lp = 0; lp < 8; ++lp)
#pragma unroll
            for (int m = 0; m < 2; ++m)
#pragma unroll
                for (int r = 0; r < 16; ++r) acc[lp][m][r] = 0.f;
        glds_patch(0, 0);
        glds_patch(1, 1);
        glds_patch(2, 2);
        // U fragments are fetched THREE positions ahead of their MFMAs into a ring of four register pairs (8 positions per
        // chunk: the ring phase is the same in every chunk).  The kernel is bound by what a CU can pull from L2 (U: 64 KB
        // per chunk, patch: 21 KB; measured 12-14 B/clk/CU with two positions in flight): bytes in flight are what raises it.
        h8 u0h, u0l, u1h, u1l, u2h, u2l, u3h, u3l;
        // only patch 0 has to be there to start: the DMAs of patches 1 and 2 (this wave's 2 x 5 or 2 x 6 youngest operations —
        // the U loads are issued behind the wait, so the count is exact) stay in flight through the first transform
        if (wave == 0) asm volatile("s_waitcnt vmcnt(12)" ::: "memory"); else asm volatile("s_waitcnt vmcnt(10)" ::: "memory");
        __syncthreads();                                             // (also orders sMax[0] = 0 / the previous pass's V reads)
        u_load(0, 0, u0h, u0l);
        u_load(0, 1, u1h, u1l);
        u_load(0, 2, u2h, u2l);
        patch_rows(0, 0, tA);
        patch_rows(0, 1, tB);
        cols_store(0);
        cols_store(2);
        __syncthreads();

        // ---- main loop: two slots per chunk ------------------------------------------------------------------------------
#ifdef AWSEG_WS_STAMP
        unsigned long long fa[5] = {0, 0, 0, 0, 0};
#define WS_ACC2(t0, t1, t2, t3, t4, t5) { fa[0] += t1 - t0; fa[1] += t2 - t1; fa[2] += t3 - t2; fa[3] += t4 - t3; fa[4] += t5 - t4; }
        unsigned long long sa = 0, ba = 0, sb = 0, bb = 0, nn = 0;
#define WS_ACC(t0, t1, t2, t3, t4) { sa += t1 - t0; ba += t2 - t1; sb += t3 - t2; bb += t4 - t3; nn += 1; }
#else
#define WS_ACC2(t0, t1, t2, t3, t4, t5)
#define WS_ACC(t0, t1, t2, t3, t4)
#endif
        WS_T(loop0);
        for (int c = 0; c < nchunks; ++c) {
            h8 vh0[2], vl0[2], vh1[2], vl1[2];
            const int cn = c + 1 < nchunks ? c + 1 : c;              // chunk of the U prefetches behind position 4 (clamped)
            // slot A: MFMAs on V rows {0, 2} of chunk c | rows {1, 3} of chunk c written, pair A of patch c+1 read and row-transformed
            WS_T(st0);
            vm_wait_keep6();                                         // the DMA issued a chunk ago (long landed)
            WS_T(sa1);
            glds_patch(c + 3, c % 3);                                // that ring slot held patch c (read a slot ago, behind a barrier)
            WS_T(sa2);
            a_load(0, vh0, vl0);
            u_load(c, 3, u3h, u3l); a_load(1, vh1, vl1);
            __builtin_amdgcn_sched_barrier(0);
            cols_store(1);
            mfma6(0, vh0, vl0, u0h, u0l);
            WS_MIX(6, 0, 2)
            __builtin_amdgcn_sched_barrier(0);
            WS_T(sa3);
            u_load(c, 4, u0h, u0l); a_load(2, vh0, vl0);
            __builtin_amdgcn_sched_barrier(0);
            cols_store(3);
            mfma6(1, vh1, vl1, u1h, u1l);
            WS_MIX(6, 0, 2)
            __builtin_amdgcn_sched_barrier(0);
            WS_T(sa4);
            u_load(c, 5, u1h, u1l); a_load(3, vh1, vl1);
            __builtin_amdgcn_sched_barrier(0);
            mfma6(2, vh0, vl0, u2h, u2l);
            __builtin_amdgcn_sched_barrier(0);
            WS_T(sa5);
            WS_ACC2(st0, sa1, sa2, sa3, sa4, sa5)
            u_load(c, 6, u2h, u2l);
            patch_rows((c + 1) % 3, 0, tA);
            mfma6(3, vh1, vl1, u3h, u3l);
            WS_MIX(6, 3, 0)
            __builtin_amdgcn_sched_barrier(0);
            WS_T(st1);
            __syncthreads();
            WS_T(st2);
            // slot B: MFMAs on rows {1, 3} of chunk c | pair B of patch c+1, rows {0, 2} of chunk c+1 written
            a_load(4, vh0, vl0);
            u_load(c, 7, u3h, u3l); a_load(5, vh1, vl1);
            __builtin_amdgcn_sched_barrier(0);
            patch_rows((c + 1) % 3, 1, tB);
            mfma6(4, vh0, vl0, u0h, u0l);
            WS_MIX(6, 3, 0)
            __builtin_amdgcn_sched_barrier(0);
            u_load(cn, 0, u0h, u0l); a_load(6, vh0, vl0);
            __builtin_amdgcn_sched_barrier(0);
            cols_store(0);
            mfma6(5, vh1, vl1, u1h, u1l);
            WS_MIX(6, 0, 2)
            __builtin_amdgcn_sched_barrier(0);
            u_load(cn, 1, u1h, u1l); a_load(7, vh1, vl1);
            __builtin_amdgcn_sched_barrier(0);
            cols_store(2);
            mfma6(6, vh0, vl0, u2h, u2l);
            WS_MIX(6, 0, 2)
            __builtin_amdgcn_sched_barrier(0);
            u_load(cn, 2, u2h, u2l);
            mfma6(7, vh1, vl1, u3h, u3l);
            __builtin_amdgcn_sched_barrier(0);
            WS_T(st3);
            __syncthreads();
            WS_T(st4);
            WS_ACC(st0, st1, st2, st3, st4)
        }
        vm_wait_all();
#ifdef AWSEG_WS_STAMP
        if (blockIdx.x == 0 && tid == 0) {
            g_ws_stamp[0] += sa; g_ws_stamp[1] += ba; g_ws_stamp[2] += sb; g_ws_stamp[3] += bb; g_ws_stamp[4] += nn;
            g_ws_stamp[5] += loop0 - blk0;
            for (int i = 0; i < 5; ++i) g_ws_stamp2[i] += fa[i];
        }
#endif
#undef WS_ACC
#undef WS_ACC2
#undef WS_MIX
    };

    run(awseg_false{});
    if (!BF16) {
        // ---- range guard: one more pass with scaled activations? --------------------------------------------------------
        if (amax > 0.f) atomicMax(&sMax[0], __builtin_bit_cast(unsigned, amax));
        __syncthreads();
        const unsigned mx = sMax[0];
        const int ex = (int)(mx >> 23) & 0xff;
        const float mf = __builtin_bit_cast(float, mx);
        if (!(mx == 0u || ex == 0xff || (mf < 8192.0f && mf >= 0.0625f))) {     // out of range (and not all zero / Inf / NaN)
            sx = 11 - (ex - 127);                                    // max|x| * 2^sx in [2^11, 2^12): 4 max|x| < 2^14
            sx = sx > 126 ? 126 : sx;
            xs = pow2f(sx);
            run(awseg_true{});
        }
    }

    // ---- output transform.  Wave (nt, ph) holds M rows 2ph, 2ph+1 (positions 8ph + 4 i' + j); Y = A^T M A is linear in M:
    // each half computes its partial 2x2 and the halves meet through LDS — m-tile ph is finished by wave ph.
    //   tmp[0][j] = M0j + M1j + M2j, tmp[1][j] = M1j - M2j - M3j;  Y[a][0] = tmp[a][0] + tmp[a][1] + tmp[a][2],
    //   Y[a][1] = tmp[a][1] - tmp[a][2] - tmp[a][3]
    float* xch = reinterpret_cast<float*>(sV);                        // [nt][dest ph][4 outputs][16 regs][64 lanes]
    // (register by register: the whole-tile form keeps eight 16-register temporaries alive and spills — scratch loads in the
    // epilogue cost the big depth-head launch, 8 chunks per block, a fifth of its time)
    auto partial = [&](int m, f32x16 (&yp)[4]) {
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            float t0[4], t1[4];
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                const float lo = acc[j][m][r], hi = acc[4 + j][m][r];
                if (ph == 0) { t0[j] = lo + hi; t1[j] = hi; }
                else { t0[j] = lo; t1[j] = -lo - hi; }
            }
            yp[0][r] = t0[0] + t0[1] + t0[2]; yp[1][r] = t0[1] - t0[2] - t0[3];
            yp[2][r] = t1[0] + t1[1] + t1[2]; yp[3][r] = t1[1] - t1[2] - t1[3];
        }
    };
    f32x16 y[4];
    {
        // the m-tile the partner finishes goes to LDS first (frees its accumulators), then this wave's own
        float* dst = xch + ((nt * 2 + (1 - ph)) * 4) * 16 * 64 + lane;
        if (ph == 0) partial(1, y); else partial(0, y);
#pragma unroll
        for (int o = 0; o < 4; ++o)
#pragma unroll
            for (int r = 0; r < 16; ++r) dst[(o * 16 + r) * 64] = y[o][r];
        if (ph == 0) partial(0, y); else partial(1, y);
    }
    __syncthreads();
    const float ysc = uscale * pow2f(-sx);                            // 2^(eu - sx)
    {
        const float* src = xch + ((nt * 2 + ph) * 4) * 16 * 64 + lane;
#pragma unroll
        for (int o = 0; o < 4; ++o)
#pragma unroll
            for (int r = 0; r < 16; ++r) y[o][r] = (y[o][r] + src[(o * 16 + r) * 64]) * ysc;
    }
    const int mt = ph;                                               // this wave's m-tile in the epilogue

    if (MODE == 0) {
        // rows = tiles of m-tile mt (tile row 4 mt + (r >> 2), tile column 4 hk + (r & 3)), columns = couts (see wino.hip)
        const int n = n0 + nt * 32 + li;
        const float sh = a.shift[n];
        const size_t img = (size_t)a.H * a.W * a.Cout;
        const __amdgpu_buffer_rsrc_t o_rsrc = __builtin_amdgcn_make_buffer_rsrc((void*)(a.out + (size_t)b * img), 0, (int)(img * 4), 0x00020000);
        const __amdgpu_buffer_rsrc_t r_rsrc = __builtin_amdgcn_make_buffer_rsrc((void*)(a.residual ? a.residual + (size_t)b * img : a.out), 0, (int)(img * 4), 0x00020000);
        const bool has_res = a.residual != nullptr;
        const int mt_u = __builtin_amdgcn_readfirstlane(mt);
        const uint32_t kOob = 0x80000000u;
        uint32_t vsel[4][2];
#pragma unroll
        for (int c = 0; c < 4; ++c)
#pragma unroll
            for (int bb = 0; bb < 2; ++bb) {
                const int xs0 = rx + a.dil * (bx * 2 * TB + 2 * c + bb);
                const int xl = a.dil * 8 * hk;
                vsel[c][bb] = (xs0 + xl < a.W) ? (uint32_t)((xl * a.Cout + n) * 4) : kOob;
            }
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            const int ty = mt_u * 4 + (r >> 2), c = r & 3;
#pragma unroll
            for (int aa = 0; aa < 2; ++aa) {
                const int yy = ry + a.dil * (by * 2 * TB + 2 * ty + aa);
                if (yy >= a.H) continue;                             // wave-uniform
#pragma unroll
                for (int bb = 0; bb < 2; ++bb) {
                    const int xs0 = rx + a.dil * (bx * 2 * TB + 2 * c + bb);
                    const uint32_t soff = (uint32_t)((yy * a.W + xs0) * a.Cout * 4);
                    float v = y[aa * 2 + bb][r] + sh;
                    if (has_res) v += __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(r_rsrc, vsel[c][bb], soff, 0));
                    __builtin_amdgcn_raw_buffer_store_b32(__builtin_bit_cast(uint32_t, act_apply(v, a.act)), o_rsrc, vsel[c][bb], soff, 0);
                }
            }
        }
    } else {
        // rows = couts n0 + 32 nt + (r & 3) + 8 (r >> 2) + 4 hk, columns = tiles of m-tile mt (tile = 32 mt + li)
        float z[4] = {0.f, 0.f, 0.f, 0.f};
        // rows 4g .. 4g+3 of a lane are four consecutive couts: one 16-byte load each for the shift and the 1x1 weights
        float shv[16], wv[16];
#pragma unroll
        for (int g4 = 0; g4 < 4; ++g4) {
            const int co = n0 + nt * 32 + 8 * g4 + 4 * hk;
            const float4 s4 = *reinterpret_cast<const float4*>(a.shift + co), w4 = *reinterpret_cast<const float4*>(a.w2 + co);
            shv[4 * g4] = s4.x; shv[4 * g4 + 1] = s4.y; shv[4 * g4 + 2] = s4.z; shv[4 * g4 + 3] = s4.w;
            wv[4 * g4] = w4.x; wv[4 * g4 + 1] = w4.y; wv[4 * g4 + 2] = w4.z; wv[4 * g4 + 3] = w4.w;
        }
#pragma unroll
        for (int r = 0; r < 16; ++r) {
#pragma unroll
            for (int o = 0; o < 4; ++o) z[o] += fmaxf(y[o][r] + shv[r], 0.f) * wv[r];
        }
#pragma unroll
        for (int o = 0; o < 4; ++o) z[o] += __shfl_xor(z[o], 32, 64);   // the other half-wave holds rows + 4 of the same tile
        __syncthreads();                                             // every wave has read its exchange data
        float* red = reinterpret_cast<float*>(sV);                   // [nt][tile][4]
        if (hk == 0) *reinterpret_cast<float4*>(red + (nt * NTILE + mt * 32 + li) * 4) = make_float4(z[0], z[1], z[2], z[3]);
        __syncthreads();
        const int tile = tid >> 2, q = tid & 3;
        const int uy = by * 2 * TB + 2 * (tile >> 3) + (q >> 1), ux = bx * 2 * TB + 2 * (tile & 7) + (q & 1);
        const int yy = ry + a.dil * uy, xx = rx + a.dil * ux;
        if (yy < a.H && xx < a.W) {
            const float zz = red[tile * 4 + q] + red[(NTILE + tile) * 4 + q] + a.b2[0];
            a.out[((int64_t)b * a.H + yy) * a.W + xx] = 1.0f / (1.0f + expf(-zz));
        }
    }
#ifdef AWSEG_WS_STAMP
    { WS_T(blk1); if (blockIdx.x == 0 && tid == 0) g_ws_stamp[7] += blk1 - blk0; }
#endif
}


// ------------------------------------------------------------------------------------------------------------------------------
// Round 3: the same operator with EIGHT waves per block that ALTERNATE between two roles.  The four-wave kernel above keeps one
// wave per SIMD: its ~280 transform instructions, 64 LDS accesses and 48 MFMAs per chunk share ONE instruction stream, and the
// stamps show 5 100 ticks per chunk against 1 536 of matrix time.  Here a wave owns ONE V row (4 positions) x all
// 64 tiles x 32 couts — 128 accumulator registers, two waves per SIMD — and every U fragment is still fetched by exactly one wave:
//   * waves 0-3 own V rows {0, 2} ("even"), waves 4-7 rows {1, 3} ("odd"); wave w and w + 4 share a SIMD;
//   * slot A of chunk c: the even waves run their 24 MFMAs on rows {0, 2} while the odd waves turn patch c into rows {1, 3} of
//     the same chunk (they need only rows 1..3 of the 4x4 input tile: B^T rows 1 and 3), fetch their U fragments of chunk c and
//     issue the LDS-DMA of patch c + 2; slot B: the odd waves run their MFMAs, the even waves build rows {0, 2} of chunk c + 1
//     (input rows 0..2) and fetch their U of chunk c + 1.  One barrier per slot.  On every SIMD one wave feeds the matrix pipe
//     while its partner does vector / LDS work, and the roles swap each slot — no wave is ever a dedicated loader;
//   * the partial inverse transforms of the four V rows meet through LDS in the epilogue (V and the patches are dead by then):
//     wave (nt, row) finishes m-tile row >> 1, output row row & 1.
constexpr int W8T = 512;
#ifdef AWSEG_WS_STAMP
__device__ unsigned long long g_w8_stamp[2][8];                       // [wave 0 | wave 4][slot A work, barrier A, slot B work, barrier B, chunks, whole block]
#endif
#ifndef AWSEG_W8_ABL
#define AWSEG_W8_ABL 0
#endif
constexpr int X_BYTES = 8 * 64 * 64 * 4;                              // epilogue exchange: 8 waves x 64 lanes x 64 floats

template <int MODE, bool BF16>
__global__ __launch_bounds__(W8T, 2)
void wino8_kernel(ws_args a)
{
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    unsigned char* sV = smem;
    unsigned char* sP = smem + V_BYTES;
    unsigned* sMax = reinterpret_cast<unsigned*>(smem + V_BYTES + P_RING * P_BYTES);
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int hk = lane >> 5, li = lane & 31;
    const int grp = wave >> 2;                                       // 0: V rows {0, 2}; 1: rows {1, 3}
    const int gw = wave & 3;                                         // wave within its group
    const int nt = wave & 1;
    const int vrow = 2 * ((wave >> 1) & 1) + grp;                    // this wave's V row (positions 4 vrow .. 4 vrow + 3)
    // Block order.  Blocks b, b + 8, ... share an XCD (and its L2).  U of a layer with many channels does not fit an L2
    // (2048 -> 256: 33 MB) and streams from the Infinity Cache at ~8.6 TB/s chip-wide — the measured pace of both kernels —
    // unless the CUs of an XCD read the SAME cout group's U at about the same time: `span` consecutive spatial tiles of the
    // XCD run cout group 0, then the same tiles group 1, ... (span = 1: a tile's groups back to back, the round-2 order).
    const int xcd = blockIdx.x & 7, jj = blockIdx.x >> 3;
    const int per = a.span * a.ngroups, sg = jj / per, rr = jj - sg * per;
    const int ng = rr / a.span, t = (sg * a.span + (rr - ng * a.span)) * 8 + xcd;
    const int gx = a.nbx * a.dil, gy = a.nby * a.dil;
    if (t >= gx * gy * a.batch) return;
    const int b = t / (gx * gy), txy = t - b * (gx * gy), tyy = txy / gx, txx = txy - tyy * gx;
    const int bx = txx % a.nbx, rx = txx / a.nbx;
    const int by = tyy % a.nby, ry = tyy / a.nby;
    const int Hs = (a.H - ry + a.dil - 1) / a.dil, Ws = (a.W - rx + a.dil - 1) / a.dil;
    if (by * 2 * TB >= Hs || bx * 2 * TB >= Ws) return;
    const float* xb = a.x + (int64_t)b * a.H * a.W * a.Cin;
    const int n0 = ng * NB;
    const int nchunks = a.Cin / KC;
    constexpr int abl = AWSEG_W8_ABL;                                // timing experiments only: compile-time switches that REMOVE one ingredient of the chunk loop

    // ---- patch LDS-DMA (issued by the odd group at the start of its transform slot: 21 instructions over its 4 waves), layout as
    // in the kernel above.  (Measured and dropped: every wave issuing its share and its next U fragments from its MFMA slot — that
    // slot went from 1 060 to 2 750 ticks for 10 vector-memory instructions, the chunk from 4 950 to 5 700.)
    const __amdgpu_buffer_rsrc_t x_rsrc = __builtin_amdgcn_make_buffer_rsrc((void*)xb, 0, (int)((size_t)a.H * a.W * a.Cin * 4), 0x00020000);
    uint32_t pvoff[6];
#pragma unroll
    for (int j = 0; j < 6; ++j) {
        const int q = (gw + 4 * j) * 64 + lane;
        const int g = q >> 2, h = q & 3;
        const int py = g / PW, pos = g - py * PW;
        const int px = pos < PW / 2 ? 2 * pos : 2 * (pos - PW / 2) + 1;
        const int sy = by * 2 * TB - 1 + py, sx = bx * 2 * TB - 1 + px;
        const int y = ry + a.dil * sy, x = rx + a.dil * sx;
        const bool ok = g < NPIX && sy >= 0 && sx >= 0 && y < a.H && x < a.W;
        pvoff[j] = ok ? (uint32_t)(((y * a.W + x) * a.Cin + h * 4) * 4) : 0x80000000u;
    }
    const int gw_u = __builtin_amdgcn_readfirstlane(gw);
    const int n_pinstr = gw_u == 0 ? 6 : 5;
    const uint32_t p_lds = __builtin_amdgcn_readfirstlane(lds_addr(sP)) + (uint32_t)gw_u * 1024u;
    auto glds_patch = [&](int chunk, int slot) {
        const uint32_t soff = (uint32_t)__builtin_amdgcn_readfirstlane((chunk < nchunks ? chunk : nchunks - 1) * KC * 4);
        const uint32_t base = (uint32_t)__builtin_amdgcn_readfirstlane((int)(p_lds + (uint32_t)(slot * P_BYTES)));
#pragma unroll
        for (int j = 0; j < 6; ++j)
            if (j < n_pinstr) bufdma16(x_rsrc, pvoff[j], soff, base + (uint32_t)(j * 4096));
    };

    // ---- transform role: item (tile, channel quad) of the group's 256 threads
    const int it = tid & 255;
    const int xtile = it >> 2, xq = it & 3;
    const int xty = xtile >> 3, xtx = xtile & 7;
    const int prd = (2 * xty * PW + xtx) * 64 + xq * 16;
    const int xsw = (xtile >> 2) & 3;
    const int vw_hi = xtile * 64 + (((xq >> 1) ^ xsw) * 16) + (xq & 1) * 8;
    const int vw_lo = xtile * 64 + (((2 + (xq >> 1)) ^ xsw) * 16) + (xq & 1) * 8;

    // ---- MFMA role.  (Measured and dropped: V as float32 in LDS, split by the multiplying wave beside its MFMAs — the MFMA slot
    // went from 1 060 to 2 000 - 2 270 ticks, the transform slot from 2 100 to 1 400, the chunk stayed at 4 950.)
    int a_hi[2], a_lo[2];
#pragma unroll
    for (int m = 0; m < 2; ++m) {
        const int tile = m * 32 + li, sw = (tile >> 2) & 3;
        a_hi[m] = tile * 64 + ((hk ^ sw) * 16);
        a_lo[m] = tile * 64 + (((2 + hk) ^ sw) * 16);
    }
    const int ncb = a.Cout / 32, cb = (n0 >> 5) + nt;
    const __amdgpu_buffer_rsrc_t u_rsrc = __builtin_amdgcn_make_buffer_rsrc((void*)a.U, 0, (int)(a.u_halfs * 2), 0x00020000);
    const uint32_t ulane = (uint32_t)(hk * 512 + li * 16);
    const uint32_t u_p = (uint32_t)ncb * 2048u, u_c = 16u * u_p;
    const uint32_t u_w = (uint32_t)(__builtin_amdgcn_readfirstlane(4 * vrow) * (int)u_p + __builtin_amdgcn_readfirstlane(cb) * 2048);
    const float uscale = *reinterpret_cast<const float*>(a.U + a.u_halfs);

    f32x16 acc[4][2];
    float amax = 0.f, xs = 1.0f;
    int sx = 0;
    if (tid == 0) sMax[0] = 0u;

    auto run = [&](auto SC) {
        constexpr bool SCALED = decltype(SC)::value;
        h8 uh[4], ul[4];
        auto u_fetch = [&](int c) {
#pragma unroll
            for (int lp = 0; lp < 4; ++lp) {
                const uint32_t so = (uint32_t)c * u_c + u_w + (uint32_t)lp * u_p;
                uh[lp] = __builtin_bit_cast(h8, __builtin_amdgcn_raw_buffer_load_b128(u_rsrc, ulane, so, 0));
                if (!BF16) ul[lp] = __builtin_bit_cast(h8, __builtin_amdgcn_raw_buffer_load_b128(u_rsrc, ulane + 1024u, so, 0));
            }
        };
        // patch `slot` -> the group's two V rows (grp, grp + 2).  B^T rows:  0: d0 - d2   1: d1 + d2   2: d2 - d1   3: d1 - d3, so the
        // even group reads input rows 0, 1, 2 (shared: d2) and the odd group rows 1, 2, 3 (shared: d1).  One V row at a time — the
        // second row's private input row is loaded after the first row's four positions are stored — keeps 32 registers of patch
        // data alive instead of 80 (the wave's 128 accumulators and 32 U registers leave ~70 for either role).
        auto transform = [&](int slot) {
            const unsigned char* pp = sP + slot * P_BYTES + prd;
            auto load_row = [&](int i, v2f (&d)[2][4]) {               // input row i of the tile's 4 x 4 patch, both channel pairs
                // ONE 16-byte read per pixel (the quad's four channels are contiguous in the pixel-major patch): the four lanes of
                // a tile cover 64 contiguous bytes and 16 lanes 256 — conflict-free, where the two 8-byte reads per pixel of the
                // four-wave kernel are 2-way conflicted (a fifth of its LDS cycles)
#pragma unroll
                for (int j = 0; j < 4; ++j) {
                    float4 q = make_float4(uscale, xs, uscale, xs);
                    if (!(abl & 32)) q = *reinterpret_cast<const float4*>(pp + (i * PW + (j & 1) * (PW / 2) + (j >> 1)) * 64);
                    v2f v0 = {q.x, q.y}, v1 = {q.z, q.w};
                    if (BF16) {
                    } else if (!SCALED) {
                        if (!(abl & 1)) {
                        amax = __builtin_fmaxf(__builtin_fmaxf(amax, __builtin_fabsf(v0.x)), __builtin_fabsf(v0.y));
                        amax = __builtin_fmaxf(__builtin_fmaxf(amax, __builtin_fabsf(v1.x)), __builtin_fabsf(v1.y));
                        }
                    } else { v0 = v0 * v2f{xs, xs}; v1 = v1 * v2f{xs, xs}; }
                    d[0][j] = v0; d[1][j] = v1;
                }
            };
            auto cols_store = [&](int vr, const v2f (&tt)[2][4]) {     // row vr of (B^T d) -> positions 4 vr .. 4 vr + 3, split, stored
#pragma unroll
                for (int j = 0; j < 4; ++j) {
                    v2f va, vb;
                    if (j == 0) { va = pk_sub(tt[0][0], tt[0][2]); vb = pk_sub(tt[1][0], tt[1][2]); }
                    else if (j == 1) { va = pk_add(tt[0][1], tt[0][2]); vb = pk_add(tt[1][1], tt[1][2]); }
                    else if (j == 2) { va = pk_sub(tt[0][2], tt[0][1]); vb = pk_sub(tt[1][2], tt[1][1]); }
                    else { va = pk_sub(tt[0][1], tt[0][3]); vb = pk_sub(tt[1][1], tt[1][3]); }
                    u32x2 H, L; unsigned h, l;
                    if (BF16) {
                        H[0] = pack_bf16(va); H[1] = pack_bf16(vb);
                        *reinterpret_cast<u32x2*>(sV + (vr * 4 + j) * V_POS + vw_hi) = H;
                        continue;
                    }
                    if (abl & 4) { H[0] = __builtin_bit_cast(unsigned, va.x); L[0] = __builtin_bit_cast(unsigned, va.y); H[1] = __builtin_bit_cast(unsigned, vb.x); L[1] = __builtin_bit_cast(unsigned, vb.y); }
                    else {
                    split_pair(va, h, l); H[0] = h; L[0] = l;
                    split_pair(vb, h, l); H[1] = h; L[1] = l;
                    }
                    if (abl & 2) { if (H[0] == 0x12345678u && L[1] == 0x9abcdef0u) *reinterpret_cast<u32x2*>(sV + vw_hi) = H; continue; }
                    *reinterpret_cast<u32x2*>(sV + (vr * 4 + j) * V_POS + vw_hi) = H;
                    *reinterpret_cast<u32x2*>(sV + (vr * 4 + j) * V_POS + vw_lo) = L;
                }
            };
            v2f sh_[2][4], pr[2][4];                                   // shared input row, private input row -> B^T d row (in place)
            if (grp == 0) {
                load_row(2, sh_); load_row(0, pr);
#pragma unroll
                for (int e = 0; e < 2; ++e)
#pragma unroll
                    for (int j = 0; j < 4; ++j) pr[e][j] = pk_sub(pr[e][j], sh_[e][j]);            // row 0: d0 - d2
                cols_store(0, pr);
                load_row(1, pr);
#pragma unroll
                for (int e = 0; e < 2; ++e)
#pragma unroll
                    for (int j = 0; j < 4; ++j) pr[e][j] = pk_sub(sh_[e][j], pr[e][j]);            // row 2: d2 - d1
                cols_store(2, pr);
            } else {
                load_row(1, sh_); load_row(2, pr);
#pragma unroll
                for (int e = 0; e < 2; ++e)
#pragma unroll
                    for (int j = 0; j < 4; ++j) pr[e][j] = pk_add(sh_[e][j], pr[e][j]);            // row 1: d1 + d2
                cols_store(1, pr);
                load_row(3, pr);
#pragma unroll
                for (int e = 0; e < 2; ++e)
#pragma unroll
                    for (int j = 0; j < 4; ++j) pr[e][j] = pk_sub(sh_[e][j], pr[e][j]);            // row 3: d1 - d3
                cols_store(3, pr);
            }
        };
        auto mma = [&]() {
#pragma unroll
            for (int lp = 0; lp < 4; ++lp) {
                const unsigned char* vp = sV + (4 * vrow + lp) * V_POS;
                h8 vh[2], vl[2];
#pragma unroll
                for (int m = 0; m < 2; ++m) {
                    vh[m] = *reinterpret_cast<const h8*>(vp + a_hi[m]);
                    if (!BF16) vl[m] = *reinterpret_cast<const h8*>(vp + a_lo[m]);
                }
#pragma unroll
                for (int m = 0; m < 2; ++m) {
                    f32x16 z = acc[lp][m];
                    if (BF16) {
                        if (MODE == 1) z = __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf8, uh[lp]), __builtin_bit_cast(bf8, vh[m]), z, 0, 0, 0);
                        else z = __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf8, vh[m]), __builtin_bit_cast(bf8, uh[lp]), z, 0, 0, 0);
                    } else if (MODE == 1) {
                        z = __builtin_amdgcn_mfma_f32_32x32x16_f16(uh[lp], vh[m], z, 0, 0, 0);
                        z = __builtin_amdgcn_mfma_f32_32x32x16_f16(ul[lp], vh[m], z, 0, 0, 0);
                        z = __builtin_amdgcn_mfma_f32_32x32x16_f16(uh[lp], vl[m], z, 0, 0, 0);
                    } else {
                        z = __builtin_amdgcn_mfma_f32_32x32x16_f16(vh[m], uh[lp], z, 0, 0, 0);
                        z = __builtin_amdgcn_mfma_f32_32x32x16_f16(vh[m], ul[lp], z, 0, 0, 0);
                        z = __builtin_amdgcn_mfma_f32_32x32x16_f16(vl[m], uh[lp], z, 0, 0, 0);
                    }
                    acc[lp][m] = z;
                }
            }
        };

#pragma unroll
        for (int lp = 0; lp < 4; ++lp)
#pragma unroll
            for (int m = 0; m < 2; ++m)
#pragma unroll
                for (int r = 0; r < 16; ++r) acc[lp][m][r] = 0.f;
        // ---- prologue: patches 0 and 1 in flight, patch 0 landed; the even waves build rows {0, 2} of chunk 0
        if (grp == 1) {
            glds_patch(0, 0);
            glds_patch(1, 1);
            if (gw_u == 0) asm volatile("s_waitcnt vmcnt(6)" ::: "memory"); else asm volatile("s_waitcnt vmcnt(5)" ::: "memory");
        }
        __syncthreads();                                             // (also orders sMax[0] = 0 and the previous pass's V reads)
        if (grp == 0) { u_fetch(0); transform(0); }
        __syncthreads();
#ifdef AWSEG_WS_STAMP
        unsigned long long w8s[5] = {0, 0, 0, 0, 0};
#define W8_T(v) const unsigned long long v = __builtin_readcyclecounter()
#define W8_ACC(t0, t1, t2, t3, t4) { w8s[0] += t1 - t0; w8s[1] += t2 - t1; w8s[2] += t3 - t2; w8s[3] += t4 - t3; w8s[4] += 1; }
#else
#define W8_T(v)
#define W8_ACC(t0, t1, t2, t3, t4)
#endif
        for (int c = 0; c < nchunks; ++c) {
            W8_T(q0);
            // slot A: even waves multiply rows {0, 2} of chunk c | odd waves: DMA of patch c + 2, U of chunk c, patch c -> rows {1, 3}
            if (grp == 0) { if (!(abl & 64)) mma(); }
            else {
                if (!(abl & 16)) glds_patch(c + 2, (c + 2) % 3);     // that ring slot held patch c - 1 (last read in slot A of chunk c - 1)
                if (!(abl & 8)) u_fetch(c);
                transform(c % 3);
                // patch c + 1 (DMA issued a chunk ago, older than this slot's n_pinstr + 8 operations) has landed
                if (BF16) { if (gw_u == 0) asm volatile("s_waitcnt vmcnt(10)" ::: "memory"); else asm volatile("s_waitcnt vmcnt(9)" ::: "memory"); }
                else { if (gw_u == 0) asm volatile("s_waitcnt vmcnt(14)" ::: "memory"); else asm volatile("s_waitcnt vmcnt(13)" ::: "memory"); }
            }
            W8_T(q1);
            __syncthreads();
            W8_T(q2);
            // slot B: odd waves multiply rows {1, 3} of chunk c | even waves: patch c + 1 -> rows {0, 2} of chunk c + 1, U of chunk c + 1
            if (grp == 1) { if (!(abl & 64)) mma(); }
            else if (c + 1 < nchunks) { if (!(abl & 8)) u_fetch(c + 1); transform((c + 1) % 3); }
            W8_T(q3);
            __syncthreads();
            W8_T(q4);
            W8_ACC(q0, q1, q2, q3, q4)
        }
        vm_wait_all();
#ifdef AWSEG_WS_STAMP
        if (blockIdx.x == 0 && (tid == 0 || tid == 256)) for (int i = 0; i < 5; ++i) g_w8_stamp[tid >> 8][i] += w8s[i];
#endif
#undef W8_T
#undef W8_ACC
    };

    run(awseg_false{});
    if (!BF16) {
        if (amax > 0.f) atomicMax(&sMax[0], __builtin_bit_cast(unsigned, amax));
        __syncthreads();
        const unsigned mx = sMax[0];
        const int ex = (int)(mx >> 23) & 0xff;
        const float mf = __builtin_bit_cast(float, mx);
        if (!(mx == 0u || ex == 0xff || (mf < 8192.0f && mf >= 0.0625f))) {
            sx = 11 - (ex - 127);
            sx = sx > 126 ? 126 : sx;
            xs = pow2f(sx);
            run(awseg_true{});
        }
    }

    // ---- output transform.  This wave holds M[vrow][0..3] (its four positions); Y = A^T M A:
    //   R_0 = M_i0 + M_i1 + M_i2, R_1 = M_i1 - M_i2 - M_i3 (column factor, in registers), then over the V rows i
    //   Y[0][b] = R_b(0) + R_b(1) + R_b(2),  Y[1][b] = R_b(1) - R_b(2) - R_b(3)  — through LDS.
    // exchange layout: float4 [nt][V row][m][b][r >> 2][lane]
    float* xch = reinterpret_cast<float*>(smem);
    {
        float* dst = xch + ((size_t)(nt * 4 + vrow) * 16 * 64 + lane) * 4;
#pragma unroll
        for (int m = 0; m < 2; ++m)
#pragma unroll
            for (int r4 = 0; r4 < 4; ++r4) {
                float4 q0, q1;
                float* p0 = &q0.x; float* p1 = &q1.x;
#pragma unroll
                for (int q = 0; q < 4; ++q) {
                    const int r = 4 * r4 + q;
                    p0[q] = acc[0][m][r] + acc[1][m][r] + acc[2][m][r];
                    p1[q] = acc[1][m][r] - acc[2][m][r] - acc[3][m][r];
                }
                *reinterpret_cast<float4*>(dst + ((m * 2 + 0) * 4 + r4) * 64 * 4) = q0;
                *reinterpret_cast<float4*>(dst + ((m * 2 + 1) * 4 + r4) * 64 * 4) = q1;
            }
    }
    __syncthreads();
    const int mt = vrow >> 1, oa = vrow & 1;                          // this wave finishes m-tile mt, output row oa
    const float ysc = uscale * pow2f(-sx);
    f32x16 y[2];                                                      // [output column b]
    {
#pragma unroll
        for (int bcol = 0; bcol < 2; ++bcol)
#pragma unroll
            for (int r4 = 0; r4 < 4; ++r4) {
                float4 s0, s1, s2;
                auto ld = [&](int i) {
                    return *reinterpret_cast<const float4*>(xch + (((size_t)(nt * 4 + i) * 16 + (mt * 2 + bcol) * 4 + r4) * 64 + lane) * 4);
                };
                s0 = ld(oa); s1 = ld(oa + 1); s2 = ld(oa + 2);       // oa = 0: rows 0, 1, 2 (+ + +); oa = 1: rows 1, 2, 3 (+ - -)
                const float* f0 = &s0.x; const float* f1 = &s1.x; const float* f2 = &s2.x;
#pragma unroll
                for (int q = 0; q < 4; ++q) y[bcol][4 * r4 + q] = (oa == 0 ? f0[q] + f1[q] + f2[q] : f0[q] - f1[q] - f2[q]) * ysc;
            }
    }

    if (MODE == 0) {
        // rows = tiles of m-tile mt (tile row 4 mt + (r >> 2), tile column 4 hk + (r & 3)), columns (lanes) = couts; output row 2 ty + oa
        const int n = n0 + nt * 32 + li;
        const float sh = a.shift[n];
        const size_t img = (size_t)a.H * a.W * a.Cout;
        const __amdgpu_buffer_rsrc_t o_rsrc = __builtin_amdgcn_make_buffer_rsrc((void*)(a.out + (size_t)b * img), 0, (int)(img * 4), 0x00020000);
        const __amdgpu_buffer_rsrc_t r_rsrc = __builtin_amdgcn_make_buffer_rsrc((void*)(a.residual ? a.residual + (size_t)b * img : a.out), 0, a.residual ? (int)(img * 4) : 0, 0x00020000);
        const int mt_u = __builtin_amdgcn_readfirstlane(mt), oa_u = __builtin_amdgcn_readfirstlane(oa);
        const uint32_t kOob = 0x80000000u;
        uint32_t vsel[4][2];
#pragma unroll
        for (int c = 0; c < 4; ++c)
#pragma unroll
            for (int bb = 0; bb < 2; ++bb) {
                const int xs0 = rx + a.dil * (bx * 2 * TB + 2 * c + bb);
                const int xl = a.dil * 8 * hk;
                vsel[c][bb] = (xs0 + xl < a.W) ? (uint32_t)((xl * a.Cout + n) * 4) : kOob;
            }
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            const int ty = mt_u * 4 + (r >> 2), c = r & 3;
            const int yy = ry + a.dil * (by * 2 * TB + 2 * ty + oa_u);
            if (yy >= a.H) continue;                                 // wave-uniform
#pragma unroll
            for (int bb = 0; bb < 2; ++bb) {
                const int xs0 = rx + a.dil * (bx * 2 * TB + 2 * c + bb);
                const uint32_t soff = (uint32_t)((yy * a.W + xs0) * a.Cout * 4);
                float v = y[bb][r] + sh;
                v += __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(r_rsrc, vsel[c][bb], soff, 0));   // zero-record descriptor without a residual
                __builtin_amdgcn_raw_buffer_store_b32(__builtin_bit_cast(uint32_t, act_apply(v, a.act)), o_rsrc, vsel[c][bb], soff, 0);
            }
        }
    } else {
        // rows = couts n0 + 32 nt + (r & 3) + 8 (r >> 2) + 4 hk, columns (lanes) = tiles of m-tile mt (tile = 32 mt + li); output row oa
        float z[2] = {0.f, 0.f};
        float shv[16], wv[16];
#pragma unroll
        for (int g4 = 0; g4 < 4; ++g4) {
            const int co = n0 + nt * 32 + 8 * g4 + 4 * hk;
            const float4 s4 = *reinterpret_cast<const float4*>(a.shift + co), w4 = *reinterpret_cast<const float4*>(a.w2 + co);
            shv[4 * g4] = s4.x; shv[4 * g4 + 1] = s4.y; shv[4 * g4 + 2] = s4.z; shv[4 * g4 + 3] = s4.w;
            wv[4 * g4] = w4.x; wv[4 * g4 + 1] = w4.y; wv[4 * g4 + 2] = w4.z; wv[4 * g4 + 3] = w4.w;
        }
#pragma unroll
        for (int r = 0; r < 16; ++r) {
#pragma unroll
            for (int o = 0; o < 2; ++o) z[o] += fmaxf(y[o][r] + shv[r], 0.f) * wv[r];
        }
#pragma unroll
        for (int o = 0; o < 2; ++o) z[o] += __shfl_xor(z[o], 32, 64);
        __syncthreads();                                             // every wave has read its exchange data
        float* red = reinterpret_cast<float*>(smem);                 // [nt][tile][output row][2]
        if (hk == 0) *reinterpret_cast<float2*>(red + ((nt * NTILE + mt * 32 + li) * 2 + oa) * 2) = make_float2(z[0], z[1]);
        __syncthreads();
        if (tid < 256) {
            const int tile = tid >> 2, q = tid & 3;                  // q = 2 * output row + output column
            const int uy = by * 2 * TB + 2 * (tile >> 3) + (q >> 1), ux = bx * 2 * TB + 2 * (tile & 7) + (q & 1);
            const int yy = ry + a.dil * uy, xx = rx + a.dil * ux;
            if (yy < a.H && xx < a.W) {
                const float zz = red[tile * 4 + q] + red[(NTILE + tile) * 4 + q] + a.b2[0];
                a.out[((int64_t)b * a.H + yy) * a.W + xx] = 1.0f / (1.0f + expf(-zz));
            }
        }
    }
}


// ------------------------------------------------------------------------------------------------------------------------------
// The eight-wave block with SYMMETRIC slots.  wino8_kernel gives the two waves of a SIMD opposite roles per slot, but the roles
// are not equally long: the stamps (tools/probe_wino_stamps.hip) read 1 060 cycles for a wave's 24 MFMAs and 2 100 - 2 600 for
// its partner's transform of two V rows, so the multiplying group waits at the barrier for half of every slot and a chunk costs
// two transform slots (~4 950 cycles).  tools/scratch/pipe_overlap.hip shows the pipes themselves DO run side by side when
// different waves of a SIMD feed them (MFMA + LDS + vector work of two waves: both at their stand-alone pace), with two
// exceptions this kernel avoids: v_pk_*_f32 waits for the matrix pipe to go idle (7x slower beside MFMAs), and MFMAs that
// depend on their predecessor hold the SIMD's vector issue.  Here EVERY wave does half of both jobs in every slot:
//   * wave (nt, j) owns V COLUMN j — positions j, 4 + j, 8 + j, 12 + j — x all 64 tiles x 32 couts (128 accumulators);
//   * slot A of chunk c: 12 MFMAs on its positions of V rows {0, 2}, and ONE V row of the transform (threads 0-255: row 1,
//     256-511: row 3, of the same chunk); slot B: 12 MFMAs on rows {1, 3}, and row 0 / row 2 of chunk c + 1;
//   * waves 0-3 multiply first and transform second, waves 4-7 (their SIMD partners) the other way round — at any time one
//     wave of a SIMD feeds the matrix pipe while the other does LDS / vector work, and both reach the barrier together;
//   * the 12 MFMAs of a slot rotate over four accumulators (dependency distance 4);
//   * U: the two rows' fragments are re-fetched right behind the MFMAs that consumed them (for the next chunk): 4 loads per slot;
//   * epilogue: the inverse transform over the V rows happens in registers (a wave holds all four rows of its column), the
//     columns meet through LDS: wave (nt, j) finishes m-tile j >> 1, output COLUMN j & 1, both output rows.
// Stamps: 3 700 cycles per chunk (wino8_kernel: 4 950), both slots ~1 500-1 900 with ~150 at each barrier; the clock under the
// kernel drops with it (2.13 -> 1.77 GHz from time / cycles), so the launch times fall by 7-10 %, not 25 %.
// (Measured and dropped: (1) issuing the slot's LDS reads for BOTH jobs up front — 256 registers, 3-4 spilled, 2 % slower;
// (2) ONE instruction stream per slot, a piece of the transform pinned behind each MFMA with scheduling barriers — an MFMA leaves
// the issue port after 8 of its 32 cycles, so the vector work would ride for free: needs ~270 registers next to 128 accumulators
// and 32 of U; hipcc spills whole accumulators of the rows the slot does not touch and the kernel runs 2x slower.  It fits the
// bf16 variant (236 registers) only.)
#ifdef AWSEG_WS_STAMP
__device__ unsigned long long g_w8s_block[16];                         // block 0, wave 0: [prologue, chunk loop, epilogue, blocks] of the block in the middle of the grid
#endif
template <int MODE, bool BF16>
__global__ __launch_bounds__(W8T, 2)
void wino8s_kernel(ws_args a)
{
#ifdef AWSEG_WS_STAMP
    const unsigned long long w8b0 = __builtin_readcyclecounter();
    unsigned long long w8b1 = 0, w8b2 = 0, w8p[5] = {0, 0, 0, 0, 0};
#endif
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    unsigned char* sV = smem;
    unsigned char* sP = smem + V_BYTES;
    unsigned* sMax = reinterpret_cast<unsigned*>(smem + V_BYTES + P_RING * P_BYTES);
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int hk = lane >> 5, li = lane & 31;
    const int grp = wave >> 2;                                       // 0: multiply, then transform; 1: transform, then multiply
    const int nt = wave & 1, vcol = wave >> 1;                       // cout half, V column
    const int xcd = blockIdx.x & 7, jj = blockIdx.x >> 3;
    int ng, b, bx, by, rx, ry;
    if (a.dil == 1 && a.span == 1) {
        // the common case with three integer divisions instead of ten (the stamps put 1 900 cycles of a block's 6 700-cycle prologue
        // into the index arithmetic in front of the first DMA request)
        const int sg = jj / a.ngroups;
        ng = jj - sg * a.ngroups;
        const int t = sg * 8 + xcd, per_img = a.nbx * a.nby;
        if (t >= per_img * a.batch) return;
        b = t / per_img;
        const int txy = t - b * per_img;
        by = txy / a.nbx; bx = txy - by * a.nbx; rx = 0; ry = 0;
        if (by * 2 * TB >= a.H || bx * 2 * TB >= a.W) return;
    } else {
        const int per = a.span * a.ngroups, sg = jj / per, rr = jj - sg * per;
        ng = rr / a.span;
        const int t = (sg * a.span + (rr - ng * a.span)) * 8 + xcd;
        const int gx = a.nbx * a.dil, gy = a.nby * a.dil;
        if (t >= gx * gy * a.batch) return;
        b = t / (gx * gy);
        const int txy = t - b * (gx * gy), tyy = txy / gx, txx = txy - tyy * gx;
        bx = txx % a.nbx; rx = txx / a.nbx;
        by = tyy % a.nby; ry = tyy / a.nby;
        const int Hs = (a.H - ry + a.dil - 1) / a.dil, Ws = (a.W - rx + a.dil - 1) / a.dil;
        if (by * 2 * TB >= Hs || bx * 2 * TB >= Ws) return;
    }
    const float* xb = a.x + (int64_t)b * a.H * a.W * a.Cin;
    const int n0 = ng * NB;
    const int nchunks = a.Cin / KC;

    // ---- patch LDS-DMA: the 21 wave-wide instructions of a patch are dealt over ALL eight waves (instruction wave + 8 k: three for
    // waves 0-4, two for 5-7) — issuing one costs a wave ~100 cycles, and the six per wave of wino8_kernel made slot A 700 cycles
    // longer for the issuing group than for its partners.  Layout as in wino8_kernel.
    const __amdgpu_buffer_rsrc_t x_rsrc = __builtin_amdgcn_make_buffer_rsrc((void*)xb, 0, (int)((size_t)a.H * a.W * a.Cin * 4), 0x00020000);
    uint32_t pvoff[3];
#pragma unroll
    for (int j = 0; j < 3; ++j) {
        const int q = (wave + 8 * j) * 64 + lane;
        const int g = q >> 2, h = q & 3;
        const int py = g / PW, pos = g - py * PW;
        const int px = pos < PW / 2 ? 2 * pos : 2 * (pos - PW / 2) + 1;
        const int sy = by * 2 * TB - 1 + py, sx = bx * 2 * TB - 1 + px;
        const int y = ry + a.dil * sy, x = rx + a.dil * sx;
        const bool ok = g < NPIX && sy >= 0 && sx >= 0 && y < a.H && x < a.W;
        pvoff[j] = ok ? (uint32_t)(((y * a.W + x) * a.Cin + h * 4) * 4) : 0x80000000u;
    }
    const int wave_u = __builtin_amdgcn_readfirstlane(wave);
    const bool three = wave_u < P_INSTR - 16;                        // this wave issues three (else two) instructions per patch
    const uint32_t p_lds = __builtin_amdgcn_readfirstlane(lds_addr(sP)) + (uint32_t)wave_u * 1024u;
    auto glds_patch = [&](int chunk, int slot) {
        const uint32_t soff = (uint32_t)__builtin_amdgcn_readfirstlane((chunk < nchunks ? chunk : nchunks - 1) * KC * 4);
        const uint32_t base = (uint32_t)__builtin_amdgcn_readfirstlane((int)(p_lds + (uint32_t)(slot * P_BYTES)));
        bufdma16(x_rsrc, pvoff[0], soff, base);
        bufdma16(x_rsrc, pvoff[1], soff, base + 8192u);
        if (three) bufdma16(x_rsrc, pvoff[2], soff, base + 16384u);
    };
    // s_waitcnt vmcnt(n + 2 | n + 3): everything but this wave's youngest n register loads and ONE patch's DMA instructions
    auto vm_wait_keep_patch_and = [&](auto NC) {
        constexpr int N = decltype(NC)::value;
        if (three) asm volatile("s_waitcnt vmcnt(%0)" :: "n"(N + 3) : "memory"); else asm volatile("s_waitcnt vmcnt(%0)" :: "n"(N + 2) : "memory");
    };

    auto vm_wait_keep = [&](auto NC) { constexpr int N = decltype(NC)::value; asm volatile("s_waitcnt vmcnt(%0)" :: "n"(N) : "memory"); };

    // ---- transform item: (tile, channel quad) x ONE V row per slot (the thread's half of the block picks the row)
    const int it = tid & 255;
    const int xtile = it >> 2, xq = it & 3;
    const int xty = xtile >> 3, xtx = xtile & 7;
    const int prd = (2 * xty * PW + xtx) * 64 + xq * 16;
    const int xsw = (xtile >> 2) & 3;
    const int vw_hi = xtile * 64 + (((xq >> 1) ^ xsw) * 16) + (xq & 1) * 8;
    const int vw_lo = xtile * 64 + (((2 + (xq >> 1)) ^ xsw) * 16) + (xq & 1) * 8;

    // ---- MFMA operands
    int a_hi[2], a_lo[2];
#pragma unroll
    for (int m = 0; m < 2; ++m) {
        const int tile = m * 32 + li, sw = (tile >> 2) & 3;
        a_hi[m] = tile * 64 + ((hk ^ sw) * 16);
        a_lo[m] = tile * 64 + (((2 + hk) ^ sw) * 16);
    }
    const int ncb = a.Cout / 32, cb = (n0 >> 5) + nt;
    const __amdgpu_buffer_rsrc_t u_rsrc = __builtin_amdgcn_make_buffer_rsrc((void*)a.U, 0, (int)(a.u_halfs * 2), 0x00020000);
    const uint32_t ulane = (uint32_t)(hk * 512 + li * 16);
    const uint32_t u_p = (uint32_t)ncb * 2048u, u_c = 16u * u_p;
    const uint32_t u_w = (uint32_t)(__builtin_amdgcn_readfirstlane(vcol) * (int)u_p + __builtin_amdgcn_readfirstlane(cb) * 2048);
    const float uscale = *reinterpret_cast<const float*>(a.U + a.u_halfs);

    f32x16 acc[4][2];                                                // [V row][m-tile]
    float amax = 0.f, xs = 1.0f;
    int sx = 0;
    if (tid == 0) sMax[0] = 0u;

    auto run = [&](auto SC) {
        constexpr bool SCALED = decltype(SC)::value;
        h8 uh[4], ul[4];                                             // [V row]
        auto u_fetch2 = [&](int c, int r0) {                         // rows r0 and r0 + 2 of chunk c
#pragma unroll
            for (int e = 0; e < 2; ++e) {
                const int i = r0 + 2 * e;
                const uint32_t so = (uint32_t)c * u_c + u_w + (uint32_t)(4 * i) * u_p;
                uh[i] = __builtin_bit_cast(h8, __builtin_amdgcn_raw_buffer_load_b128(u_rsrc, ulane, so, 0));
                if (!BF16) ul[i] = __builtin_bit_cast(h8, __builtin_amdgcn_raw_buffer_load_b128(u_rsrc, ulane + 1024u, so, 0));
            }
        };
        // patch `slot` -> ONE V row.  B^T rows:  0: d0 - d2   1: d1 + d2   2: d2 - d1   3: d1 - d3.  PHASE 0 (slot A): threads 0-255 build
        // row 1, 256-511 row 3 (input rows 1, 2 | 1, 3); PHASE 1 (slot B, next chunk's patch): row 0 | row 2 (input rows 0, 2 | 2, 1).
        // max|x| is taken once per input row and block: d1 by the row-1 threads, d3 by row 3, d0 and d2 by row 0.
        auto transform = [&](int slot, auto PH) {
            constexpr int PHASE = decltype(PH)::value ? 1 : 0;
            const unsigned char* pp = sP + slot * P_BYTES + prd;
            // scalar float32 arithmetic on purpose: v_pk_add_f32 waits for the matrix pipe (see the header of this kernel)
            auto load_row = [&](int i, bool track, float (&d)[4][4]) {
#pragma unroll
                for (int j = 0; j < 4; ++j) {
                    const float4 q = *reinterpret_cast<const float4*>(pp + (i * PW + (j & 1) * (PW / 2) + (j >> 1)) * 64);
                    d[j][0] = q.x; d[j][1] = q.y; d[j][2] = q.z; d[j][3] = q.w;
                    if (BF16) {
                    } else if (!SCALED) {
                        if (track) {
                            amax = __builtin_fmaxf(__builtin_fmaxf(amax, __builtin_fabsf(q.x)), __builtin_fabsf(q.y));
                            amax = __builtin_fmaxf(__builtin_fmaxf(amax, __builtin_fabsf(q.z)), __builtin_fabsf(q.w));
                        }
                    } else {
#pragma unroll
                        for (int ch = 0; ch < 4; ++ch) d[j][ch] *= xs;
                    }
                }
            };
            auto cols_store = [&](int vr, const float (&tt)[4][4]) {   // row vr of (B^T d) -> positions 4 vr .. 4 vr + 3, split, stored
#pragma unroll
                for (int j = 0; j < 4; ++j) {
                    float o[4];
#pragma unroll
                    for (int ch = 0; ch < 4; ++ch)
                        o[ch] = j == 0 ? tt[0][ch] - tt[2][ch] : j == 1 ? tt[1][ch] + tt[2][ch] : j == 2 ? tt[2][ch] - tt[1][ch] : tt[1][ch] - tt[3][ch];
                    const v2f va = {o[0], o[1]}, vb = {o[2], o[3]};
                    u32x2 H, L; unsigned h, l;
                    if (BF16) {
                        H[0] = pack_bf16(va); H[1] = pack_bf16(vb);
                        *reinterpret_cast<u32x2*>(sV + (vr * 4 + j) * V_POS + vw_hi) = H;
                        continue;
                    }
                    split_pair(va, h, l); H[0] = h; L[0] = l;
                    split_pair(vb, h, l); H[1] = h; L[1] = l;
                    *reinterpret_cast<u32x2*>(sV + (vr * 4 + j) * V_POS + vw_hi) = H;
                    *reinterpret_cast<u32x2*>(sV + (vr * 4 + j) * V_POS + vw_lo) = L;
                }
            };
            float p[4][4], q[4][4];
            auto combine = [&](bool plus) {
#pragma unroll
                for (int j = 0; j < 4; ++j)
#pragma unroll
                    for (int ch = 0; ch < 4; ++ch) p[j][ch] = plus ? p[j][ch] + q[j][ch] : p[j][ch] - q[j][ch];
            };
            if (PHASE == 0) {
                if (grp == 0) { load_row(1, true, p); load_row(2, false, q); combine(true); cols_store(1, p); }     // row 1: d1 + d2
                else { load_row(1, false, p); load_row(3, true, q); combine(false); cols_store(3, p); }             // row 3: d1 - d3
            } else {
                if (grp == 0) { load_row(0, true, p); load_row(2, true, q); combine(false); cols_store(0, p); }     // row 0: d0 - d2
                else { load_row(2, false, p); load_row(1, false, q); combine(false); cols_store(2, p); }            // row 2: d2 - d1
            }
        };
        // 12 MFMAs on V rows r0 and r0 + 2 of this wave's column: the three product terms, each over the four accumulators
        auto mma2 = [&](int r0) {
            h8 vh[2][2], vl[2][2];
#pragma unroll
            for (int e = 0; e < 2; ++e) {
                const unsigned char* vp = sV + (4 * (r0 + 2 * e)) * V_POS + vcol * V_POS;
#pragma unroll
                for (int m = 0; m < 2; ++m) {
                    vh[e][m] = *reinterpret_cast<const h8*>(vp + a_hi[m]);
                    if (!BF16) vl[e][m] = *reinterpret_cast<const h8*>(vp + a_lo[m]);
                }
            }
#pragma unroll
            for (int term = 0; term < (BF16 ? 1 : 3); ++term)
#pragma unroll
                for (int e = 0; e < 2; ++e)
#pragma unroll
                    for (int m = 0; m < 2; ++m) {
                        const int i = r0 + 2 * e;
                        f32x16 z = acc[i][m];
                        if (BF16) {
                            if (MODE == 1) z = __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf8, uh[i]), __builtin_bit_cast(bf8, vh[e][m]), z, 0, 0, 0);
                            else z = __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf8, vh[e][m]), __builtin_bit_cast(bf8, uh[i]), z, 0, 0, 0);
                        } else {
                            const h8 va = term == 2 ? vl[e][m] : vh[e][m];
                            const h8 ua = term == 1 ? ul[i] : uh[i];
                            if (MODE == 1) z = __builtin_amdgcn_mfma_f32_32x32x16_f16(ua, va, z, 0, 0, 0);
                            else z = __builtin_amdgcn_mfma_f32_32x32x16_f16(va, ua, z, 0, 0, 0);
                        }
                        acc[i][m] = z;
                    }
        };

        // ---- prologue: patches 0 and 1 in flight, U of chunk 0, patch 0 landed; every thread builds its row (0 | 2) of chunk 0
#ifdef AWSEG_WS_STAMP
        if (!SCALED) w8p[0] = __builtin_readcyclecounter();
#endif
        glds_patch(0, 0);
        if (nchunks > 1) glds_patch(1, 1);
        u_fetch2(0, 0);
        u_fetch2(0, 1);
#ifdef AWSEG_WS_STAMP
        if (!SCALED) w8p[1] = __builtin_readcyclecounter();
#endif
        asm volatile("" ::: "memory");                               // (the 128 accumulator moves go BEHIND the requests: the round trip hides them)
#pragma unroll
        for (int i = 0; i < 4; ++i)
#pragma unroll
            for (int m = 0; m < 2; ++m)
#pragma unroll
                for (int r = 0; r < 16; ++r) acc[i][m][r] = 0.f;
        if (nchunks > 1) vm_wait_keep_patch_and(awseg_int<BF16 ? 4 : 8>{});   // patch 0: everything but patch 1 and the U fragments behind it
        else vm_wait_keep(awseg_int<BF16 ? 4 : 8>{});
#ifdef AWSEG_WS_STAMP
        if (!SCALED) w8p[2] = __builtin_readcyclecounter();
#endif
        __syncthreads();                                             // (also orders sMax[0] = 0 and the previous pass's V reads)
#ifdef AWSEG_WS_STAMP
        if (!SCALED) w8p[3] = __builtin_readcyclecounter();
#endif
        transform(0, awseg_true{});
#ifdef AWSEG_WS_STAMP
        if (!SCALED) w8p[4] = __builtin_readcyclecounter();
#endif
        __syncthreads();
#ifdef AWSEG_WS_STAMP
        unsigned long long w8s[5] = {0, 0, 0, 0, 0}, w8f[2] = {0, 0};
#define W8_T(v) const unsigned long long v = __builtin_readcyclecounter()
#define W8_ACC(t0, t1, t2, t3, t4) { w8s[0] += t1 - t0; w8s[1] += t2 - t1; w8s[2] += t3 - t2; w8s[3] += t4 - t3; w8s[4] += 1; }
#define W8_FINE(t0, t1, t2) { w8f[0] += t1 - t0; w8f[1] += t2 - t1; }
#else
#define W8_T(v)
#define W8_ACC(t0, t1, t2, t3, t4)
#define W8_FINE(t0, t1, t2)
#endif
#ifdef AWSEG_WS_STAMP
        if (!SCALED) w8b1 = __builtin_readcyclecounter();
#endif
        for (int c = 0; c < nchunks; ++c) {
            const bool more = c + 1 < nchunks;
            W8_T(q0);
            // slot A: MFMAs on rows {0, 2} of chunk c | patch c -> rows {1, 3} of chunk c | DMA of patch c + 2 (its ring slot held patch
            // c - 1, last read in slot A of chunk c - 1).  At the end patch c + 1 (DMA issued a chunk ago) has landed: behind it this
            // wave issued, in program order, the U loads counted below and one patch's DMA instructions.
            // (no DMA past the last chunk: the two clamped re-fetches per block were 43 KB of wasted reads and a memory round trip
            // in front of the epilogue; the wait then counts the U loads only)
            const bool dma = c + 2 < nchunks;
            if (grp == 0) {
                mma2(0);
                if (more) u_fetch2(c + 1, 0);
                W8_T(qa);
                if (dma) glds_patch(c + 2, (c + 2) % 3);
                W8_T(qb);
                transform(c % 3, awseg_false{});
                W8_FINE(q0, qa, qb)
                if (more) { if (dma) vm_wait_keep_patch_and(awseg_int<BF16 ? 4 : 8>{}); else vm_wait_keep(awseg_int<BF16 ? 4 : 8>{}); }   // U rows {1, 3} of chunk c (slot B), rows {0, 2} of chunk c + 1
            } else {
                W8_T(qa);
                if (dma) glds_patch(c + 2, (c + 2) % 3);
                W8_T(qb);
                transform(c % 3, awseg_false{});
                W8_T(qc);
                W8_FINE(qa, qb, qc)
                mma2(0);
                if (more) {
                    u_fetch2(c + 1, 0);
                    if (dma) vm_wait_keep_patch_and(awseg_int<BF16 ? 6 : 12>{}); else vm_wait_keep(awseg_int<BF16 ? 6 : 12>{});            // U rows {0, 2} and {1, 3} of chunk c, rows {0, 2} of chunk c + 1
                }
            }
            W8_T(q1);
            __syncthreads();
            W8_T(q2);
            // slot B: MFMAs on rows {1, 3} of chunk c | patch c + 1 -> rows {0, 2} of chunk c + 1
            if (grp == 0) {
                mma2(1);
                if (more) { u_fetch2(c + 1, 1); transform((c + 1) % 3, awseg_true{}); }
            } else {
                if (more) transform((c + 1) % 3, awseg_true{});
                mma2(1);
                if (more) u_fetch2(c + 1, 1);
            }
            W8_T(q3);
            __syncthreads();
            W8_T(q4);
            W8_ACC(q0, q1, q2, q3, q4)
        }
#ifdef AWSEG_WS_STAMP
        if (!SCALED) w8b2 = __builtin_readcyclecounter();
#endif
        vm_wait_all();
#ifdef AWSEG_WS_STAMP
        if (blockIdx.x == 0 && (tid == 0 || tid == 256)) { for (int i = 0; i < 5; ++i) g_w8_stamp[tid >> 8][i] += w8s[i]; g_w8_stamp[tid >> 8][5] += w8f[0]; g_w8_stamp[tid >> 8][6] += w8f[1]; }
#endif
#undef W8_T
#undef W8_ACC
#undef W8_FINE
    };

    run(awseg_false{});
    if (!BF16) {
        // (the stamps put 5 k cycles here: 512 lanes' LDS atomics on ONE word serialise — reduce over the wave first, one atomic per wave)
#pragma unroll
        for (int o = 32; o >= 1; o >>= 1) amax = __builtin_fmaxf(amax, __shfl_xor(amax, o, 64));
        if (lane == 0 && amax > 0.f) atomicMax(&sMax[0], __builtin_bit_cast(unsigned, amax));
        __syncthreads();
        const unsigned mx = sMax[0];
        const int ex = (int)(mx >> 23) & 0xff;
        const float mf = __builtin_bit_cast(float, mx);
        if (!(mx == 0u || ex == 0xff || (mf < 8192.0f && mf >= 0.0625f))) {
            sx = 11 - (ex - 127);
            sx = sx > 126 ? 126 : sx;
            xs = pow2f(sx);
            run(awseg_true{});
        }
    }

#ifdef AWSEG_WS_STAMP
    const unsigned long long w8e1 = __builtin_readcyclecounter();
#endif
    // ---- output transform.  This wave holds M[0..3][vcol]; Y = A^T M A:
    //   C_0 = M_0j + M_1j + M_2j, C_1 = M_1j - M_2j - M_3j (row factor, in registers), then over the V columns j
    //   Y[a][0] = C_a(0) + C_a(1) + C_a(2),  Y[a][1] = C_a(1) - C_a(2) - C_a(3)  — through LDS.
    // exchange layout: float4 [nt][V column][m][a][r >> 2][lane]
    float* xch = reinterpret_cast<float*>(smem);
    {
        float* dst = xch + ((size_t)(nt * 4 + vcol) * 16 * 64 + lane) * 4;
#pragma unroll
        for (int m = 0; m < 2; ++m)
#pragma unroll
            for (int r4 = 0; r4 < 4; ++r4) {
                float4 q0, q1;
                float* p0 = &q0.x; float* p1 = &q1.x;
#pragma unroll
                for (int q = 0; q < 4; ++q) {
                    const int r = 4 * r4 + q;
                    p0[q] = acc[0][m][r] + acc[1][m][r] + acc[2][m][r];
                    p1[q] = acc[1][m][r] - acc[2][m][r] - acc[3][m][r];
                }
                *reinterpret_cast<float4*>(dst + ((m * 2 + 0) * 4 + r4) * 64 * 4) = q0;
                *reinterpret_cast<float4*>(dst + ((m * 2 + 1) * 4 + r4) * 64 * 4) = q1;
            }
    }
    __syncthreads();
#ifdef AWSEG_WS_STAMP
    const unsigned long long w8e2 = __builtin_readcyclecounter();
#endif
    const int mt = vcol >> 1, ob = vcol & 1;                          // this wave finishes m-tile mt, output column ob
    const float ysc = uscale * pow2f(-sx);
    f32x16 y[2];                                                      // [output row a]
    {
#pragma unroll
        for (int arow = 0; arow < 2; ++arow)
#pragma unroll
            for (int r4 = 0; r4 < 4; ++r4) {
                float4 s0, s1, s2;
                auto ld = [&](int j) {
                    return *reinterpret_cast<const float4*>(xch + (((size_t)(nt * 4 + j) * 16 + (mt * 2 + arow) * 4 + r4) * 64 + lane) * 4);
                };
                s0 = ld(ob); s1 = ld(ob + 1); s2 = ld(ob + 2);       // ob = 0: columns 0, 1, 2 (+ + +); ob = 1: columns 1, 2, 3 (+ - -)
                const float* f0 = &s0.x; const float* f1 = &s1.x; const float* f2 = &s2.x;
#pragma unroll
                for (int q = 0; q < 4; ++q) y[arow][4 * r4 + q] = (ob == 0 ? f0[q] + f1[q] + f2[q] : f0[q] - f1[q] - f2[q]) * ysc;
            }
    }

#ifdef AWSEG_WS_STAMP
    const unsigned long long w8e3 = __builtin_readcyclecounter();
#endif
    if (MODE == 0) {
        // rows = tiles of m-tile mt (tile row 4 mt + (r >> 2), tile column 4 hk + (r & 3)), columns (lanes) = couts; output column 2 tx + ob
        const int n = n0 + nt * 32 + li;
        float sh;
        {
            // the shift arrives through a vector-memory load; hipcc's wait insertion re-arms `s_waitcnt vmcnt(0)` for it in every
            // basic block of the branchy store loop below — which on gfx9 also waits for the previous STORE.  Passing the value
            // through one asm move ends the dependence on the load here.
            const float sh_ld = a.shift[n];
            asm volatile("v_mov_b32 %0, %1" : "=v"(sh) : "v"(sh_ld));
        }
        const bool relu = a.act == AWSEG_ACT_RELU;
        const size_t img = (size_t)a.H * a.W * a.Cout;
        const __amdgpu_buffer_rsrc_t o_rsrc = __builtin_amdgcn_make_buffer_rsrc((void*)(a.out + (size_t)b * img), 0, (int)(img * 4), 0x00020000);
        const __amdgpu_buffer_rsrc_t r_rsrc = __builtin_amdgcn_make_buffer_rsrc((void*)(a.residual ? a.residual + (size_t)b * img : a.out), 0, a.residual ? (int)(img * 4) : 0, 0x00020000);
        const int mt_u = __builtin_amdgcn_readfirstlane(mt), ob_u = __builtin_amdgcn_readfirstlane(ob);
        const uint32_t kOob = 0x80000000u;
        uint32_t vsel[4];
#pragma unroll
        for (int c = 0; c < 4; ++c) {
            const int xs0 = rx + a.dil * (bx * 2 * TB + 2 * c + ob_u);
            const int xl = a.dil * 8 * hk;
            vsel[c] = (xs0 + xl < a.W) ? (uint32_t)((xl * a.Cout + n) * 4) : kOob;
        }
        // The stamps put 10-11 k cycles of an 18 k-cycle epilogue into these 32 stores per wave: gfx9's vmcnt counts loads AND stores,
        // so a residual load in front of every store (through a zero-record descriptor when there is no residual) made each store
        // wait for the previous one's write acknowledgement.  Two separate code paths: without a residual — every 3x3 of the ResNet
        // bottlenecks — no load is issued and nothing waits; with one, all 32 loads go out first.
        auto store_all = [&](auto HR) {
            constexpr bool HAS_RES = decltype(HR)::value;
            float rv[16][2];
            if (HAS_RES) {
#pragma unroll
                for (int r = 0; r < 16; ++r)
#pragma unroll
                    for (int aa = 0; aa < 2; ++aa) {
                        const int ty = mt_u * 4 + (r >> 2), c = r & 3;
                        const int xs0 = rx + a.dil * (bx * 2 * TB + 2 * c + ob_u);
                        const int yy = ry + a.dil * (by * 2 * TB + 2 * ty + aa);
                        const uint32_t soff = (uint32_t)((yy * a.W + xs0) * a.Cout * 4);
                        rv[r][aa] = yy < a.H ? __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(r_rsrc, vsel[c], soff, 0)) : 0.f;
                    }
            }
#pragma unroll
            for (int r = 0; r < 16; ++r)
#pragma unroll
                for (int aa = 0; aa < 2; ++aa) {
                    const int ty = mt_u * 4 + (r >> 2), c = r & 3;
                    const int xs0 = rx + a.dil * (bx * 2 * TB + 2 * c + ob_u);
                    const int yy = ry + a.dil * (by * 2 * TB + 2 * ty + aa);
                    if (yy >= a.H) continue;                         // wave-uniform
                    const uint32_t soff = (uint32_t)((yy * a.W + xs0) * a.Cout * 4);
                    float v = y[aa][r] + sh;
                    if (HAS_RES) v += rv[r][aa];
                    if (relu) asm("v_max_f32 %0, 0, %0" : "+v"(v));
                    __builtin_amdgcn_raw_buffer_store_b32(__builtin_bit_cast(uint32_t, v), o_rsrc, vsel[c], soff, 0);
                }
        };
        if (a.residual) store_all(awseg_true{}); else store_all(awseg_false{});      // block-uniform; dword accesses: no 16-byte store hazard
    } else {
        // rows = couts n0 + 32 nt + (r & 3) + 8 (r >> 2) + 4 hk, columns (lanes) = tiles of m-tile mt (tile = 32 mt + li); output column ob
        float z[2] = {0.f, 0.f};
        float shv[16], wv[16];
#pragma unroll
        for (int g4 = 0; g4 < 4; ++g4) {
            const int co = n0 + nt * 32 + 8 * g4 + 4 * hk;
            const float4 s4 = *reinterpret_cast<const float4*>(a.shift + co), w4 = *reinterpret_cast<const float4*>(a.w2 + co);
            shv[4 * g4] = s4.x; shv[4 * g4 + 1] = s4.y; shv[4 * g4 + 2] = s4.z; shv[4 * g4 + 3] = s4.w;
            wv[4 * g4] = w4.x; wv[4 * g4 + 1] = w4.y; wv[4 * g4 + 2] = w4.z; wv[4 * g4 + 3] = w4.w;
        }
#pragma unroll
        for (int r = 0; r < 16; ++r) {
#pragma unroll
            for (int o = 0; o < 2; ++o) z[o] += fmaxf(y[o][r] + shv[r], 0.f) * wv[r];
        }
#pragma unroll
        for (int o = 0; o < 2; ++o) z[o] += __shfl_xor(z[o], 32, 64);
        __syncthreads();                                             // every wave has read its exchange data
        float* red = reinterpret_cast<float*>(smem);                 // [nt][tile][output row][output column]
        if (hk == 0) {
            red[((nt * NTILE + mt * 32 + li) * 2 + 0) * 2 + ob] = z[0];
            red[((nt * NTILE + mt * 32 + li) * 2 + 1) * 2 + ob] = z[1];
        }
        __syncthreads();
        if (tid < 256) {
            const int tile = tid >> 2, q = tid & 3;                  // q = 2 * output row + output column
            const int uy = by * 2 * TB + 2 * (tile >> 3) + (q >> 1), ux = bx * 2 * TB + 2 * (tile & 7) + (q & 1);
            const int yy = ry + a.dil * uy, xx = rx + a.dil * ux;
            if (yy < a.H && xx < a.W) {
                const float zz = red[tile * 4 + q] + red[(NTILE + tile) * 4 + q] + a.b2[0];
                a.out[((int64_t)b * a.H + yy) * a.W + xx] = 1.0f / (1.0f + expf(-zz));
            }
        }
    }
#ifdef AWSEG_WS_STAMP
    if (blockIdx.x == gridDim.x / 2 && tid == 0) { const unsigned long long w8b3 = __builtin_readcyclecounter(); g_w8s_block[0] += w8b1 - w8b0; g_w8s_block[1] += w8b2 - w8b1; g_w8s_block[2] += w8b3 - w8b2; g_w8s_block[3] += 1; g_w8s_block[4] += w8e1 - w8b2; g_w8s_block[5] += w8e2 - w8e1; g_w8s_block[6] += w8e3 - w8e2; g_w8s_block[7] += w8b3 - w8e3; g_w8s_block[8] += w8p[0] - w8b0; g_w8s_block[9] += w8p[1] - w8p[0]; g_w8s_block[10] += w8p[2] - w8p[1]; g_w8s_block[11] += w8p[3] - w8p[2]; g_w8s_block[12] += w8p[4] - w8p[3]; g_w8s_block[13] += w8b1 - w8p[4]; }
#endif
}


// ------------------------------------------------------------------------------------------------------------------------------
// wino8s_kernel as a PERSISTENT block: a.tpb tiles per block (linear block indices blockIdx.x, + gridDim.x, ...).  One block fits a
// CU (the 128 KB exchange buffer of the epilogue), so the prologue of every tile — ~6.5 k cycles whose critical path is the first
// patch's memory round trip — and its epilogue (~8 k) are dead time for the matrix pipe: 30 % of a 4-chunk block, 25 % of the
// depth head's 8-chunk blocks (block-level stamps).  Here chunk 0 of every tile lives in a FOURTH patch slot behind
// the exchange buffer, and a block requests the next tile's first patch in front of its own epilogue's stores: the round trip is
// covered by the epilogue and the next tile's index arithmetic.  Everything else is wino8s_kernel (same arithmetic, same order).
constexpr int LDS8P_BYTES = X_BYTES + P_BYTES + 64;
// MODE 2 — the SegFormer depth head in ONE launch (PKG/models/model.py:42-52 on the upsampled features, :219-221): the block's
// input patch is not fetched but GENERATED.  The first 3x3 runs on a x32 bilinear upsampling, so inside one cell of that
// upsampling its pre-activation is an exact bilinear form A + B t + C s + D t s of the local pixel coordinates (t, s); the few
// pixels whose 3x3 window crosses a cell boundary or the image border are single columns / rows (E + F s, E' + F' t) and single
// pixels (constants).  awseg_upconv_forms (depthfuse.hip) builds those forms once per frame at the encoder's resolution; here a
// 16 x 16 tile lies inside ONE cell, and per chunk of 16 channels
//   * waves 0-3 fetch the tile's coefficient set by ONE LDS-DMA instruction each (3.75 KB: the cell's form, the forms of the <= 4
//     special rows, the <= 4 special columns and their crossings; unused / out-of-image pieces are out-of-range offsets: zeros),
//     three chunks ahead, ring of three 4 KB slots behind the exchange buffer;
//   * 432 threads = (patch row, 6 position groups, channel quad) evaluate relu(R + S t) for three pixels each — two FMAs for the
//     row's R, S, one FMA + one max per value — and write the chunk's patch (same LDS image the DMA used to fill) in slot A of
//     the chunk before;
// so the 128-channel full-resolution map (8.6 GB written, 10.6 GB read per batch of 8 at 1024 x 2048) never exists.
constexpr int CF_BYTES = 4096;                                         // one coefficient set: 240 pieces of 16 bytes
constexpr int CF_F2 = 1280;                                            // byte offset of the special-column pieces inside a set
constexpr int LDS8G_BYTES = X_BYTES + 3 * CF_BYTES + 64;
constexpr int GEN_THREADS = 18 * 24;                                   // (patch row, position group 0-5, channel quad)
template <int MODE, bool BF16>
__global__ __launch_bounds__(W8T, 2)
void wino8p_kernel(ws_args a)
{
    constexpr bool GEN = MODE == 2;
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    unsigned char* sV = smem;
    unsigned* sMax = reinterpret_cast<unsigned*>(smem + X_BYTES + (GEN ? 3 * CF_BYTES : P_BYTES));   // behind the fourth patch slot / the coefficient ring (the exchange buffer covers everything below X_BYTES)
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int hk = lane >> 5, li = lane & 31;
    const int grp = wave >> 2;                                       // 0: multiply, then transform; 1: transform, then multiply
    const int nt = wave & 1, vcol = wave >> 1;                       // cout half, V column
    // tile of linear block index bidx (the order of wino8s_kernel); false: nothing to do there
    auto decode = [&](int bidx, int& ng, int& b, int& bx, int& by, int& rx, int& ry) -> bool {
        if (bidx >= a.nblocks) return false;
        const int xcd = bidx & 7, jj = bidx >> 3;
        if (a.dil == 1 && a.span == 1) {
            // the common case with three integer divisions instead of ten (the stamps put 1 900 cycles of a block's 6 700-cycle prologue
            // into the index arithmetic in front of the first DMA request)
            const int sg = jj / a.ngroups;
            ng = jj - sg * a.ngroups;
            const int t = sg * 8 + xcd, per_img = a.nbx * a.nby;
            if (t >= per_img * a.batch) return false;
            b = t / per_img;
            const int txy = t - b * per_img;
            by = txy / a.nbx; bx = txy - by * a.nbx; rx = 0; ry = 0;
            if (by * 2 * TB >= a.H || bx * 2 * TB >= a.W) return false;
        } else {
            const int per = a.span * a.ngroups, sg = jj / per, rr = jj - sg * per;
            ng = rr / a.span;
            const int t = (sg * a.span + (rr - ng * a.span)) * 8 + xcd;
            const int gx = a.nbx * a.dil, gy = a.nby * a.dil;
            if (t >= gx * gy * a.batch) return false;
            b = t / (gx * gy);
            const int txy = t - b * (gx * gy), tyy = txy / gx, txx = txy - tyy * gx;
            bx = txx % a.nbx; rx = txx / a.nbx;
            by = tyy % a.nby; ry = tyy / a.nby;
            const int Hs = (a.H - ry + a.dil - 1) / a.dil, Ws = (a.W - rx + a.dil - 1) / a.dil;
            if (by * 2 * TB >= Hs || bx * 2 * TB >= Ws) return false;
        }
        return true;
    };
    int ng = 0, b = 0, bx = 0, by = 0, rx = 0, ry = 0, n0 = 0;                 // the current tile
    const int nchunks = a.Cin / KC;

    // ---- patch LDS-DMA: the 21 wave-wide instructions of a patch are dealt over ALL eight waves (instruction wave + 8 k: three for
    // waves 0-4, two for 5-7) — issuing one costs a wave ~100 cycles, and the six per wave of wino8_kernel made slot A 700 cycles
    // longer for the issuing group than for its partners.  Layout as in wino8_kernel.
    __amdgpu_buffer_rsrc_t x_rsrc = __builtin_amdgcn_make_buffer_rsrc((void*)a.x, 0, 0, 0x00020000);
    uint32_t pvoff[3] = {0x80000000u, 0x80000000u, 0x80000000u};
    auto patch_addr = [&](int tb, int tbx, int tby, int trx, int try_, __amdgpu_buffer_rsrc_t& rsrc, uint32_t (&pv)[3]) {
        rsrc = __builtin_amdgcn_make_buffer_rsrc((void*)(a.x + (int64_t)tb * a.H * a.W * a.Cin), 0, (int)((size_t)a.H * a.W * a.Cin * 4), 0x00020000);
#pragma unroll
        for (int j = 0; j < 3; ++j) {
            const int q = (wave + 8 * j) * 64 + lane;
            const int g = q >> 2, h = q & 3;
            const int py = g / PW, pos = g - py * PW;
            const int px = pos < PW / 2 ? 2 * pos : 2 * (pos - PW / 2) + 1;
            const int sy = tby * 2 * TB - 1 + py, sx = tbx * 2 * TB - 1 + px;
            const int y = try_ + a.dil * sy, x = trx + a.dil * sx;
            const bool ok = g < NPIX && sy >= 0 && sx >= 0 && y < a.H && x < a.W;
            pv[j] = ok ? (uint32_t)(((y * a.W + x) * a.Cin + h * 4) * 4) : 0x80000000u;
        }
    };
    const int wave_u = __builtin_amdgcn_readfirstlane(wave);
    const bool three = wave_u < P_INSTR - 16;                        // this wave issues three (else two) instructions per patch
    // patch slots 0-2 behind V as in wino8s_kernel; slot 3 — chunk 0 of every tile — behind the exchange buffer, so that the NEXT
    // tile's first patch can travel while this tile's epilogue uses everything below X_BYTES
    // (MODE 2: the patches are generated, never prefetched across tiles: three slots, chunk c in slot c % 3)
    auto slot_off = [](int slot) { return slot < 3 ? V_BYTES + slot * P_BYTES : X_BYTES; };
    auto pslot = [](int chunk) { return GEN ? chunk % 3 : (chunk == 0 ? 3 : chunk % 3); };
    const uint32_t p_lds = __builtin_amdgcn_readfirstlane(lds_addr(smem)) + (uint32_t)wave_u * 1024u;
    auto glds_patch_of = [&](const __amdgpu_buffer_rsrc_t& rsrc, const uint32_t (&pv)[3], int chunk, int slot) {
        const uint32_t soff = (uint32_t)__builtin_amdgcn_readfirstlane((chunk < nchunks ? chunk : nchunks - 1) * KC * 4);
        const uint32_t base = (uint32_t)__builtin_amdgcn_readfirstlane((int)(p_lds + (uint32_t)slot_off(slot)));
        bufdma16(rsrc, pv[0], soff, base);
        bufdma16(rsrc, pv[1], soff, base + 8192u);
        if (three) bufdma16(rsrc, pv[2], soff, base + 16384u);
    };
    auto glds_patch = [&](int chunk, int slot) { glds_patch_of(x_rsrc, pvoff, chunk, slot); };
    // s_waitcnt vmcnt(n + 2 | n + 3): everything but this wave's youngest n register loads and ONE patch's DMA instructions
    auto vm_wait_keep_patch_and = [&](auto NC) {
        constexpr int N = decltype(NC)::value;
        if (GEN) {                                                   // ONE coefficient DMA per chunk, waves 0-3 only
            if (wave_u < 4) asm volatile("s_waitcnt vmcnt(%0)" :: "n"(N + 1) : "memory"); else asm volatile("s_waitcnt vmcnt(%0)" :: "n"(N) : "memory");
        }
        else if (three) asm volatile("s_waitcnt vmcnt(%0)" :: "n"(N + 3) : "memory"); else asm volatile("s_waitcnt vmcnt(%0)" :: "n"(N + 2) : "memory");
    };

    auto vm_wait_keep = [&](auto NC) { constexpr int N = decltype(NC)::value; asm volatile("s_waitcnt vmcnt(%0)" :: "n"(N) : "memory"); };

    // ---- MODE 2: coefficient sets and the patch generator (see the header of this kernel)
    // special-pixel id along an axis of n low-resolution samples (32 n pixels): the image border pixels and the two pixels at
    // every cell boundary (coordinates 15, 16 mod 32); -1: an interior pixel of its cell — or outside the image
    auto sid = [](int y, int n) -> int {
        const int nn = 32 * n, m = y & 31;
        if (y < 0 || y >= nn) return -1;
        if (y == 0) return 0;
        if (y == nn - 1) return 2 * n + 1;
        return m == 15 ? 1 + 2 * (y >> 5) : (m == 16 ? 2 + 2 * (y >> 5) : -1);
    };
    const int64_t f4_floats = GEN ? (int64_t)(3 * a.fh + 3) * (a.fw + 1) * 4 * a.Cin : 0;
    const int64_t f2_floats = GEN ? (int64_t)(3 * a.fh + 3) * (2 * a.fw + 2) * 2 * a.Cin : 0;
    uint32_t cvoff = 0x80000000u;                                   // this lane's piece of a coefficient set (waves 0-3), chunk 0
    // piece P = 64 wave + lane of a set: P < 80: F4 [row slot 0-4][coefficient A B C D][channel quad]; 80 <= P < 240: F2 [row slot]
    // [column slot 0-3][coefficient E F][channel quad].  Row slot 0 = the tile's cell row; 1-4 = patch rows 0, 1, 16, 17 where they are
    // special; column slots = patch columns 0, 1, 16, 17 where they are special.  Anything else: an offset no descriptor covers (zeros).
    auto coef_addr = [&](int tb, int tbx, int tby, __amdgpu_buffer_rsrc_t& rsrc, uint32_t& cv) {
        rsrc = __builtin_amdgcn_make_buffer_rsrc((void*)(a.forms + (int64_t)tb * (f4_floats + f2_floats)), 0, (int)((f4_floats + f2_floats) * 4), 0x00020000);
        const int P = wave * 64 + lane;
        const int y0 = tby * 16, x0 = tbx * 16;
        const int ky = y0 < 16 ? -1 : (y0 - 16) >> 5, kx = x0 < 16 ? -1 : (x0 - 16) >> 5;
        const bool is4 = P < 80;
        const int Pq = is4 ? P : P - 80;
        const int rs = is4 ? Pq >> 4 : Pq >> 5;
        const int rr = rs == 1 ? 0 : (rs == 2 ? 1 : (rs == 3 ? 16 : 17));
        const int rid = sid(y0 - 1 + rr, a.fh);
        const int rsel = rs == 0 ? ky + 1 : (rid < 0 ? -1 : a.fh + 1 + rid);
        const int cs = (Pq >> 3) & 3;
        const int cc = cs == 0 ? 0 : (cs == 1 ? 1 : (cs == 2 ? 16 : 17));
        const int cid = sid(x0 - 1 + cc, a.fw);
        const int qd = P & 3;
        const int o4 = ((rsel * (a.fw + 1) + kx + 1) * 4 + ((Pq >> 2) & 3)) * a.Cin + qd * 4;
        const int o2 = (int)f4_floats + ((rsel * (2 * a.fw + 2) + cid) * 2 + ((Pq >> 2) & 1)) * a.Cin + qd * 4;
        const bool ok = wave < 4 && P < 240 && rs < 5 && rsel >= 0 && (is4 || cid >= 0);
        cv = ok ? (uint32_t)((is4 ? o4 : o2) * 4) : 0x80000000u;
    };
    auto glds_coef_of = [&](const __amdgpu_buffer_rsrc_t& rsrc, uint32_t cv, int chunk) {      // set `chunk` -> ring slot chunk % 3
        if (wave_u < 4) {
            const uint32_t soff = (uint32_t)__builtin_amdgcn_readfirstlane(chunk * KC * 4);
            const uint32_t base = (uint32_t)__builtin_amdgcn_readfirstlane((int)(p_lds + (uint32_t)(X_BYTES + (chunk % 3) * CF_BYTES)));
            bufdma16(rsrc, cv, soff, base);
        }
    };
    // generator thread = (patch row r, position group pg, channel quad q): positions {0, 1, 8, 9, 16, 17}[pg] (the only ones that can
    // be special columns: patch columns 0, 2, 16, 1, 15, 17), 2 + pg and 10 + pg of the row's 18 (even columns first, as the DMA laid
    // them out); 8 consecutive lanes write 128 contiguous bytes
    int g_c = 0, g_e = 0, g_w0 = 0, g_w1 = 0;
    float g_s = 0.f, g_t0 = 0.f, g_t1 = 0.f;                        // (the third pixel is the second one's left neighbour: t1 - 1)
    bool g_sp = false;
    auto gen_setup = [&](int tbx, int tby) {
        const int gt = tid < GEN_THREADS ? tid : 0;
        const int r = gt / 24, rem = gt - r * 24, pg = rem >> 2, q = rem & 3;
        const int y0 = tby * 16, x0 = tbx * 16;
        const int ky = y0 < 16 ? -1 : (y0 - 16) >> 5, kx = x0 < 16 ? -1 : (x0 - 16) >> 5;
        const int y = y0 - 1 + r;
        const bool rsp = y < 0 || y >= a.H || sid(y, a.fh) >= 0;      // outside the image: its row slot holds zeros
        const int rslot = rsp ? (r == 0 ? 1 : (r == 1 ? 2 : (r == 16 ? 3 : 4))) : 0;
        g_s = rsp ? 0.f : (float)(y - (16 + 32 * ky));
        g_c = rslot * 256 + q * 16;
        const int pos0 = pg < 2 ? pg : (pg < 4 ? 6 + pg : 12 + pg);
        const int pc0 = pos0 < 9 ? 2 * pos0 : 2 * (pos0 - 9) + 1;
        const int xa = x0 - 1 + pc0, xc = 16 + 32 * kx;
        const bool edge = pc0 == 0 || pc0 == 1 || pc0 == 16 || pc0 == 17;
        g_sp = edge && (xa < 0 || xa >= a.W || sid(xa, a.fw) >= 0);
        const int cslot = pc0 == 0 ? 0 : (pc0 == 1 ? 1 : (pc0 == 16 ? 2 : 3));
        g_e = CF_F2 + ((rslot * 4 + cslot) * 2) * 64 + q * 16;
        const int pos1 = 2 + pg;                                     // columns 4 .. 14; position 10 + pg: columns 3 .. 13
        g_t0 = (float)(xa - xc);
        g_t1 = (float)(x0 - 1 + 2 * pos1 - xc);
        g_w0 = (r * PW + pos0) * 64 + q * 16;
        g_w1 = (r * PW + pos1) * 64 + q * 16;
    };
    auto gen_patch = [&](int chunk) {
        if (tid < GEN_THREADS) {
            const unsigned char* cs = smem + X_BYTES + (chunk % 3) * CF_BYTES;
            const float4 fa = *reinterpret_cast<const float4*>(cs + g_c), fb = *reinterpret_cast<const float4*>(cs + g_c + 64);
            const float4 fc = *reinterpret_cast<const float4*>(cs + g_c + 128), fd = *reinterpret_cast<const float4*>(cs + g_c + 192);
            const float4 fe = *reinterpret_cast<const float4*>(cs + g_e), ff = *reinterpret_cast<const float4*>(cs + g_e + 64);
            const float A[4] = {fa.x, fa.y, fa.z, fa.w}, B[4] = {fb.x, fb.y, fb.z, fb.w}, C[4] = {fc.x, fc.y, fc.z, fc.w};
            const float D[4] = {fd.x, fd.y, fd.z, fd.w}, E[4] = {fe.x, fe.y, fe.z, fe.w}, F[4] = {ff.x, ff.y, ff.z, ff.w};
            float R[4], S[4], v0[4], v1[4], v2[4];
#pragma unroll
            for (int ch = 0; ch < 4; ++ch) {
                R[ch] = __builtin_fmaf(C[ch], g_s, A[ch]);
                S[ch] = __builtin_fmaf(D[ch], g_s, B[ch]);
                const float vs = __builtin_fmaf(F[ch], g_s, E[ch]);
                const float vi = __builtin_fmaf(S[ch], g_t0, R[ch]);
                v0[ch] = __builtin_fmaxf(g_sp ? vs : vi, 0.f);
                const float p1 = __builtin_fmaf(S[ch], g_t1, R[ch]);
                v1[ch] = __builtin_fmaxf(p1, 0.f);
                v2[ch] = __builtin_fmaxf(__builtin_fmaf(S[ch], g_t1 - 1.0f, R[ch]), 0.f);
            }
            unsigned char* pw = smem + V_BYTES + (chunk % 3) * P_BYTES;
            *reinterpret_cast<float4*>(pw + g_w0) = make_float4(v0[0], v0[1], v0[2], v0[3]);
            *reinterpret_cast<float4*>(pw + g_w1) = make_float4(v1[0], v1[1], v1[2], v1[3]);
            *reinterpret_cast<float4*>(pw + g_w1 + 8 * 64) = make_float4(v2[0], v2[1], v2[2], v2[3]);
        }
    };

    // ---- transform item: (tile, channel quad) x ONE V row per slot (the thread's half of the block picks the row)
    const int it = tid & 255;
    const int xtile = it >> 2, xq = it & 3;
    const int xty = xtile >> 3, xtx = xtile & 7;
    const int prd = (2 * xty * PW + xtx) * 64 + xq * 16;
    const int xsw = (xtile >> 2) & 3;
    const int vw_hi = xtile * 64 + (((xq >> 1) ^ xsw) * 16) + (xq & 1) * 8;
    const int vw_lo = xtile * 64 + (((2 + (xq >> 1)) ^ xsw) * 16) + (xq & 1) * 8;

    // ---- MFMA operands
    int a_hi[2], a_lo[2];
#pragma unroll
    for (int m = 0; m < 2; ++m) {
        const int tile = m * 32 + li, sw = (tile >> 2) & 3;
        a_hi[m] = tile * 64 + ((hk ^ sw) * 16);
        a_lo[m] = tile * 64 + (((2 + hk) ^ sw) * 16);
    }
    const int ncb = a.Cout / 32;
    const __amdgpu_buffer_rsrc_t u_rsrc = __builtin_amdgcn_make_buffer_rsrc((void*)a.U, 0, (int)(a.u_halfs * 2), 0x00020000);
    const uint32_t ulane = (uint32_t)(hk * 512 + li * 16);
    const uint32_t u_p = (uint32_t)ncb * 2048u, u_c = 16u * u_p;
    uint32_t u_w = 0;                                              // per tile: V column and cout block of this wave
    const float uscale = *reinterpret_cast<const float*>(a.U + a.u_halfs);

    f32x16 acc[4][2];                                                // [V row][m-tile]
    float amax = 0.f, xs = 1.0f;
    int sx = 0;
    bool patch0_requested = false;

    auto run = [&](auto SC) {
        constexpr bool SCALED = decltype(SC)::value;
        h8 uh[4], ul[4];                                             // [V row]
        auto u_fetch2 = [&](int c, int r0) {                         // rows r0 and r0 + 2 of chunk c
#pragma unroll
            for (int e = 0; e < 2; ++e) {
                const int i = r0 + 2 * e;
                const uint32_t so = (uint32_t)c * u_c + u_w + (uint32_t)(4 * i) * u_p;
                uh[i] = __builtin_bit_cast(h8, __builtin_amdgcn_raw_buffer_load_b128(u_rsrc, ulane, so, 0));
                if (!BF16) ul[i] = __builtin_bit_cast(h8, __builtin_amdgcn_raw_buffer_load_b128(u_rsrc, ulane + 1024u, so, 0));
            }
        };
        // patch `slot` -> ONE V row.  B^T rows:  0: d0 - d2   1: d1 + d2   2: d2 - d1   3: d1 - d3.  PHASE 0 (slot A): threads 0-255 build
        // row 1, 256-511 row 3 (input rows 1, 2 | 1, 3); PHASE 1 (slot B, next chunk's patch): row 0 | row 2 (input rows 0, 2 | 2, 1).
        // max|x| is taken once per input row and block: d1 by the row-1 threads, d3 by row 3, d0 and d2 by row 0.
        auto transform = [&](int slot, auto PH) {
            constexpr int PHASE = decltype(PH)::value ? 1 : 0;
            const unsigned char* pp = smem + slot_off(slot) + prd;
            // scalar float32 arithmetic on purpose: v_pk_add_f32 waits for the matrix pipe (see the header of this kernel)
            auto load_row = [&](int i, bool track, float (&d)[4][4]) {
#pragma unroll
                for (int j = 0; j < 4; ++j) {
                    const float4 q = *reinterpret_cast<const float4*>(pp + (i * PW + (j & 1) * (PW / 2) + (j >> 1)) * 64);
                    d[j][0] = q.x; d[j][1] = q.y; d[j][2] = q.z; d[j][3] = q.w;
                    if (BF16) {
                    } else if (!SCALED) {
                        if (track) {
                            amax = __builtin_fmaxf(__builtin_fmaxf(amax, __builtin_fabsf(q.x)), __builtin_fabsf(q.y));
                            amax = __builtin_fmaxf(__builtin_fmaxf(amax, __builtin_fabsf(q.z)), __builtin_fabsf(q.w));
                        }
                    } else {
#pragma unroll
                        for (int ch = 0; ch < 4; ++ch) d[j][ch] *= xs;
                    }
                }
            };
            auto cols_store = [&](int vr, const float (&tt)[4][4]) {   // row vr of (B^T d) -> positions 4 vr .. 4 vr + 3, split, stored
#pragma unroll
                for (int j = 0; j < 4; ++j) {
                    float o[4];
#pragma unroll
                    for (int ch = 0; ch < 4; ++ch)
                        o[ch] = j == 0 ? tt[0][ch] - tt[2][ch] : j == 1 ? tt[1][ch] + tt[2][ch] : j == 2 ? tt[2][ch] - tt[1][ch] : tt[1][ch] - tt[3][ch];
                    const v2f va = {o[0], o[1]}, vb = {o[2], o[3]};
                    u32x2 H, L; unsigned h, l;
                    if (BF16) {
                        H[0] = pack_bf16(va); H[1] = pack_bf16(vb);
                        *reinterpret_cast<u32x2*>(sV + (vr * 4 + j) * V_POS + vw_hi) = H;
                        continue;
                    }
                    split_pair(va, h, l); H[0] = h; L[0] = l;
                    split_pair(vb, h, l); H[1] = h; L[1] = l;
                    *reinterpret_cast<u32x2*>(sV + (vr * 4 + j) * V_POS + vw_hi) = H;
                    *reinterpret_cast<u32x2*>(sV + (vr * 4 + j) * V_POS + vw_lo) = L;
                }
            };
            float p[4][4], q[4][4];
            auto combine = [&](bool plus) {
#pragma unroll
                for (int j = 0; j < 4; ++j)
#pragma unroll
                    for (int ch = 0; ch < 4; ++ch) p[j][ch] = plus ? p[j][ch] + q[j][ch] : p[j][ch] - q[j][ch];
            };
            if (PHASE == 0) {
                if (grp == 0) { load_row(1, true, p); load_row(2, false, q); combine(true); cols_store(1, p); }     // row 1: d1 + d2
                else { load_row(1, false, p); load_row(3, true, q); combine(false); cols_store(3, p); }             // row 3: d1 - d3
            } else {
                if (grp == 0) { load_row(0, true, p); load_row(2, true, q); combine(false); cols_store(0, p); }     // row 0: d0 - d2
                else { load_row(2, false, p); load_row(1, false, q); combine(false); cols_store(2, p); }            // row 2: d2 - d1
            }
        };
        // 12 MFMAs on V rows r0 and r0 + 2 of this wave's column: the three product terms, each over the four accumulators
        auto mma2 = [&](int r0) {
            h8 vh[2][2], vl[2][2];
#pragma unroll
            for (int e = 0; e < 2; ++e) {
                const unsigned char* vp = sV + (4 * (r0 + 2 * e)) * V_POS + vcol * V_POS;
#pragma unroll
                for (int m = 0; m < 2; ++m) {
                    vh[e][m] = *reinterpret_cast<const h8*>(vp + a_hi[m]);
                    if (!BF16) vl[e][m] = *reinterpret_cast<const h8*>(vp + a_lo[m]);
                }
            }
#pragma unroll
            for (int term = 0; term < (BF16 ? 1 : 3); ++term)
#pragma unroll
                for (int e = 0; e < 2; ++e)
#pragma unroll
                    for (int m = 0; m < 2; ++m) {
                        const int i = r0 + 2 * e;
                        f32x16 z = acc[i][m];
                        if (BF16) {
                            if (MODE != 0) z = __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf8, uh[i]), __builtin_bit_cast(bf8, vh[e][m]), z, 0, 0, 0);
                            else z = __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf8, vh[e][m]), __builtin_bit_cast(bf8, uh[i]), z, 0, 0, 0);
                        } else {
                            const h8 va = term == 2 ? vl[e][m] : vh[e][m];
                            const h8 ua = term == 1 ? ul[i] : uh[i];
                            if (MODE != 0) z = __builtin_amdgcn_mfma_f32_32x32x16_f16(ua, va, z, 0, 0, 0);
                            else z = __builtin_amdgcn_mfma_f32_32x32x16_f16(va, ua, z, 0, 0, 0);
                        }
                        acc[i][m] = z;
                    }
        };

        // ---- prologue: patches 0 and 1 in flight, U of chunk 0, patch 0 landed; every thread builds its row (0 | 2) of chunk 0
        if (GEN) {
            // coefficient sets 0-2 (a persistent block requests them before the previous tile's epilogue)
            if (!patch0_requested) { for (int c3 = 0; c3 < 3 && c3 < nchunks; ++c3) glds_coef_of(x_rsrc, cvoff, c3); }
        } else {
        if (!patch0_requested) glds_patch(0, 3);                     // (a persistent block requests it before the previous tile's epilogue)
        if (nchunks > 1) glds_patch(1, 1);
        }
        patch0_requested = false;
        u_fetch2(0, 0);
        u_fetch2(0, 1);
        asm volatile("" ::: "memory");                               // (the 128 accumulator moves go BEHIND the requests: the round trip hides them)
#pragma unroll
        for (int i = 0; i < 4; ++i)
#pragma unroll
            for (int m = 0; m < 2; ++m)
#pragma unroll
                for (int r = 0; r < 16; ++r) acc[i][m][r] = 0.f;
        if (nchunks > 1 && !GEN) vm_wait_keep_patch_and(awseg_int<BF16 ? 4 : 8>{});   // patch 0: everything but patch 1 and the U fragments behind it
        else vm_wait_keep(awseg_int<BF16 ? 4 : 8>{});                // (MODE 2: all three coefficient sets)
        __syncthreads();                                             // (also orders sMax[0] = 0 and the previous pass's V reads)
        if (GEN) { gen_patch(0); __syncthreads(); }
        transform(pslot(0), awseg_true{});
        __syncthreads();
#define W8_T(v)
#define W8_ACC(t0, t1, t2, t3, t4)
#define W8_FINE(t0, t1, t2)
        for (int c = 0; c < nchunks; ++c) {
            const bool more = c + 1 < nchunks;
            W8_T(q0);
            // slot A: MFMAs on rows {0, 2} of chunk c | patch c -> rows {1, 3} of chunk c | DMA of patch c + 2 (its ring slot held patch
            // c - 1, last read in slot A of chunk c - 1).  At the end patch c + 1 (DMA issued a chunk ago) has landed: behind it this
            // wave issued, in program order, the U loads counted below and one patch's DMA instructions.
            // (no DMA past the last chunk: the two clamped re-fetches per block were 43 KB of wasted reads and a memory round trip
            // in front of the epilogue; the wait then counts the U loads only)
            // MODE 2: coefficient set c + 3 requested (its ring slot held set c, read by the generator a chunk ago); the patch of chunk
            // c + 1 is generated behind this slot's transform from set c + 1, which landed a chunk ago (the wait below, last chunk)
            const bool dma = GEN ? c + 3 < nchunks : c + 2 < nchunks;
            if (grp == 0) {
                mma2(0);
                if (more) u_fetch2(c + 1, 0);
                W8_T(qa);
                if (dma) { if (GEN) glds_coef_of(x_rsrc, cvoff, c + 3); else glds_patch(c + 2, (c + 2) % 3); }
                W8_T(qb);
                transform(pslot(c), awseg_false{});
                if (GEN && more) gen_patch(c + 1);
                W8_FINE(q0, qa, qb)
                if (more) { if (dma) vm_wait_keep_patch_and(awseg_int<BF16 ? 4 : 8>{}); else vm_wait_keep(awseg_int<BF16 ? 4 : 8>{}); }   // U rows {1, 3} of chunk c (slot B), rows {0, 2} of chunk c + 1
            } else {
                W8_T(qa);
                if (dma) { if (GEN) glds_coef_of(x_rsrc, cvoff, c + 3); else glds_patch(c + 2, (c + 2) % 3); }
                W8_T(qb);
                transform(pslot(c), awseg_false{});
                if (GEN && more) gen_patch(c + 1);
                W8_T(qc);
                W8_FINE(qa, qb, qc)
                mma2(0);
                if (more) {
                    u_fetch2(c + 1, 0);
                    if (dma) vm_wait_keep_patch_and(awseg_int<BF16 ? 6 : 12>{}); else vm_wait_keep(awseg_int<BF16 ? 6 : 12>{});            // U rows {0, 2} and {1, 3} of chunk c, rows {0, 2} of chunk c + 1
                }
            }
            W8_T(q1);
            __syncthreads();
            W8_T(q2);
            // slot B: MFMAs on rows {1, 3} of chunk c | patch c + 1 -> rows {0, 2} of chunk c + 1
            if (grp == 0) {
                mma2(1);
                if (more) { u_fetch2(c + 1, 1); transform(pslot(c + 1), awseg_true{}); }
            } else {
                if (more) transform(pslot(c + 1), awseg_true{});
                mma2(1);
                if (more) u_fetch2(c + 1, 1);
            }
            W8_T(q3);
            __syncthreads();
            W8_T(q4);
            W8_ACC(q0, q1, q2, q3, q4)
        }
        vm_wait_all();
#undef W8_T
#undef W8_ACC
#undef W8_FINE
    };

    // ---- the tiles of this block: linear block indices blockIdx.x, + gridDim.x, ... (gridDim.x is a multiple of 8: same XCD)
    int bidx = blockIdx.x;
    bool have = decode(bidx, ng, b, bx, by, rx, ry);
    if (have) { if (GEN) coef_addr(b, bx, by, x_rsrc, cvoff); else patch_addr(b, bx, by, rx, ry, x_rsrc, pvoff); }
    for (int iter = 0; iter < a.tpb; ++iter) {
        // the next tile, decoded now so that its first patch can be requested in front of this tile's epilogue
        int n_ng = 0, n_b = 0, n_bx = 0, n_by = 0, n_rx = 0, n_ry = 0;
        const int nbidx = bidx + (int)gridDim.x;
        const bool nhave = iter + 1 < a.tpb && decode(nbidx, n_ng, n_b, n_bx, n_by, n_rx, n_ry);
        __amdgpu_buffer_rsrc_t n_rsrc = x_rsrc;
        uint32_t n_pv[3] = {0x80000000u, 0x80000000u, 0x80000000u};
        uint32_t n_cv = 0x80000000u;
        if (nhave) { if (GEN) coef_addr(n_b, n_bx, n_by, n_rsrc, n_cv); else patch_addr(n_b, n_bx, n_by, n_rx, n_ry, n_rsrc, n_pv); }
        if (have) {
        if (GEN) gen_setup(bx, by);
        n0 = ng * NB;
        u_w = (uint32_t)(__builtin_amdgcn_readfirstlane(vcol) * (int)u_p + __builtin_amdgcn_readfirstlane((n0 >> 5) + nt) * 2048);
        amax = 0.f; xs = 1.0f; sx = 0;
        if (tid == 0) sMax[0] = 0u;                                  // (ordered by the prologue's first barrier)
        run(awseg_false{});
        if (!BF16) {
            // (the stamps put 5 k cycles here: 512 lanes' LDS atomics on ONE word serialise — reduce over the wave first, one atomic per wave)
#pragma unroll
            for (int o = 32; o >= 1; o >>= 1) amax = __builtin_fmaxf(amax, __shfl_xor(amax, o, 64));
            if (lane == 0 && amax > 0.f) atomicMax(&sMax[0], __builtin_bit_cast(unsigned, amax));
            __syncthreads();
            const unsigned mx = sMax[0];
            const int ex = (int)(mx >> 23) & 0xff;
            const float mf = __builtin_bit_cast(float, mx);
            if (!(mx == 0u || ex == 0xff || (mf < 8192.0f && mf >= 0.0625f))) {
                sx = 11 - (ex - 127);
                sx = sx > 126 ? 126 : sx;
                xs = pow2f(sx);
                run(awseg_true{});
            }
        }

        // ---- output transform.  This wave holds M[0..3][vcol]; Y = A^T M A:
        //   C_0 = M_0j + M_1j + M_2j, C_1 = M_1j - M_2j - M_3j (row factor, in registers), then over the V columns j
        //   Y[a][0] = C_a(0) + C_a(1) + C_a(2),  Y[a][1] = C_a(1) - C_a(2) - C_a(3)  — through LDS.
        // exchange layout: float4 [nt][V column][m][a][r >> 2][lane]
        float* xch = reinterpret_cast<float*>(smem);
        {
            float* dst = xch + ((size_t)(nt * 4 + vcol) * 16 * 64 + lane) * 4;
#pragma unroll
            for (int m = 0; m < 2; ++m)
#pragma unroll
                for (int r4 = 0; r4 < 4; ++r4) {
                    float4 q0, q1;
                    float* p0 = &q0.x; float* p1 = &q1.x;
#pragma unroll
                    for (int q = 0; q < 4; ++q) {
                        const int r = 4 * r4 + q;
                        p0[q] = acc[0][m][r] + acc[1][m][r] + acc[2][m][r];
                        p1[q] = acc[1][m][r] - acc[2][m][r] - acc[3][m][r];
                    }
                    *reinterpret_cast<float4*>(dst + ((m * 2 + 0) * 4 + r4) * 64 * 4) = q0;
                    *reinterpret_cast<float4*>(dst + ((m * 2 + 1) * 4 + r4) * 64 * 4) = q1;
                }
        }
        __syncthreads();
        const int mt = vcol >> 1, ob = vcol & 1;                          // this wave finishes m-tile mt, output column ob
        const float ysc = uscale * pow2f(-sx);
        f32x16 y[2];                                                      // [output row a]
        {
#pragma unroll
            for (int arow = 0; arow < 2; ++arow)
#pragma unroll
                for (int r4 = 0; r4 < 4; ++r4) {
                    float4 s0, s1, s2;
                    auto ld = [&](int j) {
                        return *reinterpret_cast<const float4*>(xch + (((size_t)(nt * 4 + j) * 16 + (mt * 2 + arow) * 4 + r4) * 64 + lane) * 4);
                    };
                    s0 = ld(ob); s1 = ld(ob + 1); s2 = ld(ob + 2);       // ob = 0: columns 0, 1, 2 (+ + +); ob = 1: columns 1, 2, 3 (+ - -)
                    const float* f0 = &s0.x; const float* f1 = &s1.x; const float* f2 = &s2.x;
#pragma unroll
                    for (int q = 0; q < 4; ++q) y[arow][4 * r4 + q] = (ob == 0 ? f0[q] + f1[q] + f2[q] : f0[q] - f1[q] - f2[q]) * ysc;
                }
        }

        if (MODE == 0) {
            // rows = tiles of m-tile mt (tile row 4 mt + (r >> 2), tile column 4 hk + (r & 3)), columns (lanes) = couts; output column 2 tx + ob
            const int n = n0 + nt * 32 + li;
            float sh;
            {
                // the shift arrives through a vector-memory load; hipcc's wait insertion re-arms `s_waitcnt vmcnt(0)` for it in every
                // basic block of the branchy store loop below — which on gfx9 also waits for the previous STORE.  Passing the value
                // through one asm move ends the dependence on the load here.
                const float sh_ld = a.shift[n];
                asm volatile("v_mov_b32 %0, %1" : "=v"(sh) : "v"(sh_ld));
            }
            if (nhave) { glds_patch_of(n_rsrc, n_pv, 0, 3); patch0_requested = true; }   // behind the last load this epilogue waits for: nothing below waits on vmcnt
        const bool relu = a.act == AWSEG_ACT_RELU;
            const size_t img = (size_t)a.H * a.W * a.Cout;
            const __amdgpu_buffer_rsrc_t o_rsrc = __builtin_amdgcn_make_buffer_rsrc((void*)(a.out + (size_t)b * img), 0, (int)(img * 4), 0x00020000);
            const __amdgpu_buffer_rsrc_t r_rsrc = __builtin_amdgcn_make_buffer_rsrc((void*)(a.residual ? a.residual + (size_t)b * img : a.out), 0, a.residual ? (int)(img * 4) : 0, 0x00020000);
            const int mt_u = __builtin_amdgcn_readfirstlane(mt), ob_u = __builtin_amdgcn_readfirstlane(ob);
            const uint32_t kOob = 0x80000000u;
            uint32_t vsel[4];
#pragma unroll
            for (int c = 0; c < 4; ++c) {
                const int xs0 = rx + a.dil * (bx * 2 * TB + 2 * c + ob_u);
                const int xl = a.dil * 8 * hk;
                vsel[c] = (xs0 + xl < a.W) ? (uint32_t)((xl * a.Cout + n) * 4) : kOob;
            }
            // The stamps put 10-11 k cycles of an 18 k-cycle epilogue into these 32 stores per wave: gfx9's vmcnt counts loads AND stores,
            // so a residual load in front of every store (through a zero-record descriptor when there is no residual) made each store
            // wait for the previous one's write acknowledgement.  Two separate code paths: without a residual — every 3x3 of the ResNet
            // bottlenecks — no load is issued and nothing waits; with one, all 32 loads go out first.
            auto store_all = [&](auto HR) {
                constexpr bool HAS_RES = decltype(HR)::value;
                float rv[16][2];
                if (HAS_RES) {
#pragma unroll
                    for (int r = 0; r < 16; ++r)
#pragma unroll
                        for (int aa = 0; aa < 2; ++aa) {
                            const int ty = mt_u * 4 + (r >> 2), c = r & 3;
                            const int xs0 = rx + a.dil * (bx * 2 * TB + 2 * c + ob_u);
                            const int yy = ry + a.dil * (by * 2 * TB + 2 * ty + aa);
                            const uint32_t soff = (uint32_t)((yy * a.W + xs0) * a.Cout * 4);
                            rv[r][aa] = yy < a.H ? __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(r_rsrc, vsel[c], soff, 0)) : 0.f;
                        }
                }
#pragma unroll
                for (int r = 0; r < 16; ++r)
#pragma unroll
                    for (int aa = 0; aa < 2; ++aa) {
                        const int ty = mt_u * 4 + (r >> 2), c = r & 3;
                        const int xs0 = rx + a.dil * (bx * 2 * TB + 2 * c + ob_u);
                        const int yy = ry + a.dil * (by * 2 * TB + 2 * ty + aa);
                        if (yy >= a.H) continue;                         // wave-uniform
                        const uint32_t soff = (uint32_t)((yy * a.W + xs0) * a.Cout * 4);
                        float v = y[aa][r] + sh;
                        if (HAS_RES) v += rv[r][aa];
                        if (relu) asm("v_max_f32 %0, 0, %0" : "+v"(v));
                        __builtin_amdgcn_raw_buffer_store_b32(__builtin_bit_cast(uint32_t, v), o_rsrc, vsel[c], soff, 0);
                    }
            };
            if (a.residual) store_all(awseg_true{}); else store_all(awseg_false{});      // block-uniform; dword accesses: no 16-byte store hazard
        } else {
            // rows = couts n0 + 32 nt + (r & 3) + 8 (r >> 2) + 4 hk, columns (lanes) = tiles of m-tile mt (tile = 32 mt + li); output column ob
            float z[2] = {0.f, 0.f};
            float shv[16], wv[16];
#pragma unroll
            for (int g4 = 0; g4 < 4; ++g4) {
                const int co = n0 + nt * 32 + 8 * g4 + 4 * hk;
                const float4 s4 = *reinterpret_cast<const float4*>(a.shift + co), w4 = *reinterpret_cast<const float4*>(a.w2 + co);
                shv[4 * g4] = s4.x; shv[4 * g4 + 1] = s4.y; shv[4 * g4 + 2] = s4.z; shv[4 * g4 + 3] = s4.w;
                wv[4 * g4] = w4.x; wv[4 * g4 + 1] = w4.y; wv[4 * g4 + 2] = w4.z; wv[4 * g4 + 3] = w4.w;
            }
#pragma unroll
            for (int r = 0; r < 16; ++r) {
#pragma unroll
                for (int o = 0; o < 2; ++o) z[o] += fmaxf(y[o][r] + shv[r], 0.f) * wv[r];
            }
        if (nhave) {                                                 // (shift and w2 have arrived: nothing below waits on vmcnt)
            if (GEN) { for (int c3 = 0; c3 < 3 && c3 < nchunks; ++c3) glds_coef_of(n_rsrc, n_cv, c3); }   // the ring lies behind the exchange buffer
            else glds_patch_of(n_rsrc, n_pv, 0, 3);
            patch0_requested = true;
        }
#pragma unroll
            for (int o = 0; o < 2; ++o) z[o] += __shfl_xor(z[o], 32, 64);
            __syncthreads();                                             // every wave has read its exchange data
            float* red = reinterpret_cast<float*>(smem);                 // [nt][tile][output row][output column]
            if (hk == 0) {
                red[((nt * NTILE + mt * 32 + li) * 2 + 0) * 2 + ob] = z[0];
                red[((nt * NTILE + mt * 32 + li) * 2 + 1) * 2 + ob] = z[1];
            }
            __syncthreads();
            if (tid < 256) {
                const int tile = tid >> 2, q = tid & 3;                  // q = 2 * output row + output column
                const int uy = by * 2 * TB + 2 * (tile >> 3) + (q >> 1), ux = bx * 2 * TB + 2 * (tile & 7) + (q & 1);
                const int yy = ry + a.dil * uy, xx = rx + a.dil * ux;
                if (yy < a.H && xx < a.W) {
                    const float zz = red[tile * 4 + q] + red[(NTILE + tile) * 4 + q] + a.b2[0];
                    a.out[((int64_t)b * a.H + yy) * a.W + xx] = 1.0f / (1.0f + expf(-zz));
                }
            }
        }

        }   // have
        __syncthreads();                                             // every wave is done with the exchange buffer / the reduction scratch
        bidx = nbidx; have = nhave;
        ng = n_ng; b = n_b; bx = n_bx; by = n_by; rx = n_rx; ry = n_ry;
        x_rsrc = n_rsrc; pvoff[0] = n_pv[0]; pvoff[1] = n_pv[1]; pvoff[2] = n_pv[2]; cvoff = n_cv;
    }
}


// (Measured and dropped in round 3: the same block on SIXTEEN waves of 128 registers — wave = 2 positions x 64 tiles x 32 couts, every
// slot all 1 024 threads share the transform (item = tile x channel pair x one V row), eight waves multiply.  Correct on the first
// run and 5-10 % slower than this kernel on every shape: 4, 8 and 16 waves all land at ~5 000 cycles per chunk, with MFMA 32 %,
// LDS 40-48 % and vector ALU 28 % busy (profiles/r03_pmc_winograd_lds.csv) — the pipes take turns instead of overlapping.)

template <int MODE, bool BF16>
int launch_ws(const ws_args& a, hipStream_t s)
{
    static int w8 = -1;                                              // AWSEG_WINO8=0: the four-wave kernel of round 2, 1: eight waves with alternating roles, 2: symmetric slots, 3: symmetric slots in persistent blocks (A/B measurements)
    if (w8 < 0) { const char* e = getenv("AWSEG_WINO8"); w8 = e ? atoi(e) : 3; }
    const int64_t tiles = (int64_t)a.nbx * a.dil * a.nby * a.dil * a.batch;
    const int64_t nblocks = ((tiles + 7) / 8) * a.ngroups * 8;
    if (nblocks >= ((int64_t)1 << 31)) return AWSEG_ERANGE;
    if (w8) {
        static int span_env = -1;
        if (span_env < 0) { const char* e = getenv("AWSEG_WINO8_SPAN"); span_env = e ? atoi(e) : 0; }
        ws_args a8 = a;
        const int64_t tiles_x = (tiles + 7) / 8;                        // spatial tiles per XCD
        int span = span_env > 0 ? span_env : 1;
        if (span > tiles_x) span = (int)tiles_x;
        while (tiles_x % span) --span;                                   // whole spans only (the grid stays a rectangle)
        a8.span = span;
        static int tpb_env = -1;
        if (tpb_env < 0) { const char* e = getenv("AWSEG_WINO8_TPB"); tpb_env = e ? atoi(e) : 0; }
        // two persistent blocks per CU in sequence where the map has that many tiles (measured flat from 2 to 4 per CU, and
        // 1.7x slower once the grid no longer covers the 256 CUs); at most 64 tiles a block.  A map of <= 512 blocks gains
        // nothing from persistence and runs the plain kernel.
        int tpb = tpb_env > 0 ? tpb_env : (int)((nblocks + 511) / 512);
        tpb = tpb < 1 ? 1 : (tpb > 64 ? 64 : tpb);
        if (w8 == 3 && tpb > 1) {
            a8.nblocks = (int)nblocks; a8.tpb = tpb;
            const int64_t grid = ((nblocks + tpb - 1) / tpb + 7) / 8 * 8;
            auto kp = wino8p_kernel<MODE, BF16>;
            hipError_t ep = hipFuncSetAttribute(reinterpret_cast<const void*>(kp), hipFuncAttributeMaxDynamicSharedMemorySize, LDS8P_BYTES);
            if (ep != hipSuccess) return (int)ep;
            hipLaunchKernelGGL(kp, dim3((unsigned)grid), dim3(W8T), LDS8P_BYTES, s, a8);
            AWSEG_LAUNCH_CHECK();
            return 0;
        }
        auto k8 = w8 >= 2 ? wino8s_kernel<MODE, BF16> : wino8_kernel<MODE, BF16>;
        constexpr int LDS8 = (LDS_BYTES > X_BYTES ? LDS_BYTES : X_BYTES);
        hipError_t e8 = hipFuncSetAttribute(reinterpret_cast<const void*>(k8), hipFuncAttributeMaxDynamicSharedMemorySize, LDS8);
        if (e8 != hipSuccess) return (int)e8;
        hipLaunchKernelGGL(k8, dim3((unsigned)nblocks), dim3(W8T), LDS8, s, a8);
        AWSEG_LAUNCH_CHECK();
        return 0;
    }
    auto kern = wino_split_kernel<MODE, BF16>;
    hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, LDS_BYTES);
    if (e != hipSuccess) return (int)e;
    hipLaunchKernelGGL(kern, dim3((unsigned)nblocks), dim3(WT), LDS_BYTES, s, a);
    AWSEG_LAUNCH_CHECK();
    return 0;
}

}  // namespace

AWSEG_API int64_t awseg_winograd_split_weight_halfs(int cin, int cout)
{
    if (cin < KC || cin % KC || cout < NB || cout % NB) return -1;
    return (int64_t)16 * cin * cout * 2;                             // 16 positions x (hi + lo)
}

namespace {
int ws_entry(bool bf16, const float* x, int batch, int height, int width, int cin, int cout, int dilation,
             const uint16_t* u_split, const float* shift, const float* residual, int act,
             const float* w2, const float* b2, float* out, awseg_stream_t stream)
{
    if (batch == 0) return 0;
    if (!x || !u_split || !shift || !out || batch < 0 || height < 1 || width < 1 || dilation < 1) return AWSEG_EINVAL;
    if (cin < KC || (cin % KC) || cout < NB || (cout % NB)) return AWSEG_ERANGE;
    if ((w2 == nullptr) != (b2 == nullptr)) return AWSEG_EINVAL;
    if (w2 && (cout != NB || residual)) return AWSEG_ERANGE;
    if (w2 && (((uintptr_t)w2 & 15) || ((uintptr_t)shift & 15))) return AWSEG_EALIGN;       // 16-byte loads in the fused-head epilogue
    if (act != AWSEG_ACT_NONE && act != AWSEG_ACT_RELU) return AWSEG_ERANGE;
    if (((uintptr_t)x & 15) || ((uintptr_t)u_split & 15)) return AWSEG_EALIGN;
    if ((int64_t)height * width * cin >= (int64_t)1 << 29 || (int64_t)height * width * cout >= (int64_t)1 << 29 ||
        (int64_t)32 * cin * cout * 2 >= (int64_t)1 << 31) return AWSEG_ERANGE;       // 32-bit byte offsets
    ws_args a;
    a.x = x; a.U = u_split; a.shift = shift; a.residual = residual; a.w2 = w2; a.b2 = b2; a.out = out;
    a.H = height; a.W = width; a.Cin = cin; a.Cout = cout; a.dil = dilation; a.act = act;
    const int hs = (height + dilation - 1) / dilation, ws = (width + dilation - 1) / dilation;
    a.nbx = (ws + 2 * TB - 1) / (2 * TB); a.nby = (hs + 2 * TB - 1) / (2 * TB); a.ngroups = cout / NB; a.batch = batch;
    a.u_halfs = (int64_t)16 * cin * cout * 2;
    a.span = 1; a.nblocks = 0; a.tpb = 1;
    a.forms = nullptr; a.fh = 0; a.fw = 0;
    if (bf16) return w2 ? launch_ws<1, true>(a, awseg_s(stream)) : launch_ws<0, true>(a, awseg_s(stream));
    return w2 ? launch_ws<1, false>(a, awseg_s(stream)) : launch_ws<0, false>(a, awseg_s(stream));
}

// MODE 2 of wino8p_kernel: always the persistent kernel (the generator lives there only)
template <bool BF16>
int launch_gen(ws_args a, hipStream_t s)
{
    const int64_t tiles = (int64_t)a.nbx * a.nby * a.batch;
    const int64_t nblocks = ((tiles + 7) / 8) * 8;
    if (nblocks >= ((int64_t)1 << 31)) return AWSEG_ERANGE;
    static int tpb_env = -1;
    if (tpb_env < 0) { const char* e = getenv("AWSEG_WINO8_TPB"); tpb_env = e ? atoi(e) : 0; }
    int tpb = tpb_env > 0 ? tpb_env : (int)((nblocks + 511) / 512);
    tpb = tpb < 1 ? 1 : (tpb > 64 ? 64 : tpb);
    a.nblocks = (int)nblocks; a.tpb = tpb;
    const int64_t grid = ((nblocks + tpb - 1) / tpb + 7) / 8 * 8;
    auto kp = wino8p_kernel<2, BF16>;
    hipError_t ep = hipFuncSetAttribute(reinterpret_cast<const void*>(kp), hipFuncAttributeMaxDynamicSharedMemorySize, LDS8G_BYTES);
    if (ep != hipSuccess) return (int)ep;
    hipLaunchKernelGGL(kp, dim3((unsigned)grid), dim3(W8T), LDS8G_BYTES, s, a);
    AWSEG_LAUNCH_CHECK();
    return 0;
}
}  // namespace

AWSEG_API int awseg_depth_head_fused(const float* forms, int batch, int h, int w, int cmid, const uint16_t* u_split, int u_is_bf16,
                                     const float* shift2, const float* w2, const float* b2, float* out, awseg_stream_t stream)
{
    if (batch == 0) return 0;
    if (!forms || !u_split || !shift2 || !w2 || !b2 || !out || batch < 0 || h < 1 || w < 1) return AWSEG_EINVAL;
    if (cmid < KC || (cmid % KC)) return AWSEG_ERANGE;
    if (((uintptr_t)forms & 15) || ((uintptr_t)u_split & 15) || ((uintptr_t)w2 & 15) || ((uintptr_t)shift2 & 15)) return AWSEG_EALIGN;
    const int64_t per_img = awseg_upconv_forms_floats(h, w, cmid);
    if (per_img * 4 >= (int64_t)1 << 31 || (int64_t)32 * cmid * NB * 2 >= (int64_t)1 << 31 || (int64_t)1024 * h * w >= (int64_t)1 << 31) return AWSEG_ERANGE;   // 32-bit byte offsets
    ws_args a;
    a.x = nullptr; a.U = u_split; a.shift = shift2; a.residual = nullptr; a.w2 = w2; a.b2 = b2; a.out = out;
    a.H = 32 * h; a.W = 32 * w; a.Cin = cmid; a.Cout = NB; a.dil = 1; a.act = AWSEG_ACT_RELU;
    a.nbx = a.W / (2 * TB); a.nby = a.H / (2 * TB); a.ngroups = 1; a.batch = batch;
    a.u_halfs = (int64_t)16 * cmid * NB * 2;
    a.span = 1; a.nblocks = 0; a.tpb = 1;
    a.forms = forms; a.fh = h; a.fw = w;
    return u_is_bf16 ? launch_gen<true>(a, awseg_s(stream)) : launch_gen<false>(a, awseg_s(stream));
}

AWSEG_API int awseg_conv3x3_winograd_split_nhwc(const float* x, int batch, int height, int width, int cin, int cout, int dilation,
                                                const uint16_t* u_split, const float* shift, const float* residual, int act,
                                                const float* w2, const float* b2, float* out, awseg_stream_t stream)
{
    return ws_entry(false, x, batch, height, width, cin, cout, dilation, u_split, shift, residual, act, w2, b2, out, stream);
}

AWSEG_API int awseg_conv3x3_winograd_bf16_nhwc(const float* x, int batch, int height, int width, int cin, int cout, int dilation,
                                               const uint16_t* u_bf16, const float* shift, const float* residual, int act,
                                               const float* w2, const float* b2, float* out, awseg_stream_t stream)
{
    return ws_entry(true, x, batch, height, width, cin, cout, dilation, u_bf16, shift, residual, act, w2, b2, out, stream);
}
