// gemm_split3.hip — the operator of gemm_split.hip (out[M,N] = act(x[M,K] . w[N,K]^T + bias (+ residual)), float32 grade on
// the f16 matrix cores with split operands; the 1x1 convolutions behind PKG/models/model.py:349 and the MiT Linear layers
// behind :193-197) rebuilt around the LDS-DMA pipeline of cdna_hip_programming.md §5 for the shapes that fill 256-wide tiles
// (N % 256 == 0, K % 32 == 0).  Round 2's kernel staged both operands through registers (global load -> split -> ds_write)
// with every wave of the block in the same phase: cycle stamps put a third of a K tile into the barrier and a third into
// staging, with the matrix pipe idle in both.  Here NO operand passes through a staging phase:
//   * x (raw float32) and the weights (a k-blocked image [N][K/32][32 hi | 32 lo] f16, unscaled low parts, written once by
//     awseg_gemm_split_weights) go global -> LDS by `buffer_load_dwordx4 ... lds`, one K tile (32 deep: 128-byte rows of both
//     operands) ahead, two stages of 64 KB; 16-byte chunks XOR-swizzled on the SOURCE address (chunk ^ ((row >> 1) & 7)), so
//     the linear LDS-DMA image is conflict-free for ds_read_b128 fragment reads;
//   * ONE barrier per K tile; a wave's only vector work in the loop is splitting the activation fragment it has just read
//     (4 v_fma_mix per pair: hi = f16(x s), lo = f16(x s - hi); s = 2^4 as in gemm_split.hip's single-accumulator form) and
//     the running max|x| of the range guard;
//   * block = 256 x 256, 8 waves as 8 x 1: a wave owns 32 ROWS x all 256 columns (1 x 8 MFMA tiles, ONE accumulator each:
//     128 registers), so every activation fragment is split by exactly one wave — 20 vector instructions beside 24 MFMAs
//     per 16-deep step.  (The first build had 2 x 4 waves of 128 x 64: each activation fragment was split by the four waves
//     that share its rows, 80 vector instructions per 24 MFMAs, and the ablation build without the split ran 25 % faster:
//     v_fma_mix does not hide behind MFMAs at that density.)  The weight fragments — no vector work — are what the waves
//     share: 36 ds_read_b128 per wave and K tile, 37 % of the LDS read rate at the MFMA-bound pace;
//   * the product is computed TRANSPOSED (weights as the MFMA's row operand): a lane owns one output ROW and its registers
//     4 consecutive COLUMNS, so the epilogue moves 16 bytes per lane and instruction (bias / residual / ReLU / rescale as
//     before; raw-buffer accesses, rows past M dropped by the hardware range check) — a quarter of the store instructions;
//   * persistent blocks, XCD-aware tile walk, the next tile's first K tile in flight during the epilogue (as before).
// Operand range: as gemm_split.hip — weights normalised when they are split, activations split optimistically (x 2^4) while
// the block tracks max|x|; a tile that met |x| >= 2^11 is recomputed with x 2^-e and rescaled in the epilogue.
#include "awseg_common.h"

namespace {

typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef _Float16 h8 __attribute__((ext_vector_type(8)));
typedef unsigned u32x4 __attribute__((ext_vector_type(4)));

constexpr int G3M_MAX = 256, G3K = 32;     // the tile has 32 WV rows (WV = 8 or 4 waves); one operand's stage: rows x 128 B
// two stages of two operands + sMax[2]; the four-wave blocks pack the weight stages (64-column tiles: 48 KB a block, THREE blocks per CU)
constexpr int g3_lds(int wv, int nt) { return 2 * (32 * wv * 128) + 2 * (wv == 8 ? 32 * wv * 128 : 32 * nt * 128) + 64; }
constexpr float kActScale0 = 16.0f;        // optimistic-pass activation scale (gemm_split.hip: split_pair_unscaled)
constexpr int kActExp0 = -4;
constexpr float kSplitLimit3 = 2048.0f;
#ifndef G3_TRANSPOSED
#define G3_TRANSPOSED 0                    // 1: weights as the MFMA's row operand, lane = output row, 16-byte epilogue accesses (measured slower)
#endif    // |x| below this splits without loss at scale 2^4 (|x s| < 2^15)

__device__ __forceinline__ float pow2f3(int e) { return __builtin_bit_cast(float, (unsigned)(127 + e) << 23); }

__device__ __forceinline__ uint32_t lds_addr3(const void* p)
{
    return (uint32_t)(uintptr_t)(__attribute__((address_space(3))) const void*)p;
}
// 16 bytes per lane global -> LDS through a buffer descriptor: LDS address = m0 + 16 * lane (wave-uniform base), the
// source address per lane (vector offset; out-of-range lanes read zeros) + a scalar offset
__device__ __forceinline__ void dma16(__amdgpu_buffer_rsrc_t rsrc, uint32_t voff, uint32_t soff, uint32_t lds_base)
{
    asm volatile("s_mov_b32 m0, %3\n\ts_nop 0\n\tbuffer_load_dwordx4 %0, %1, %2 offen lds" : : "v"(voff), "s"(rsrc), "s"(soff), "s"(lds_base) : "m0", "memory");
}

// two float32 -> packed f16 high parts and packed f16 low parts of (a s, b s): hi = f16(x s) (round to nearest), lo = f16(x s - hi)
// (exact in float32, then exactly representable while normal).  Four mixed-precision FMAs, no packed-float32 instruction
// (those cost extra beside MFMAs: MI355X_MICROARCH.md cycle constants).
__device__ __forceinline__ void split2(float a, float b, float s, unsigned& hi, unsigned& lo)
{
    asm("v_fma_mixlo_f16 %0, %2, %4, 0\n\t"
        "v_fma_mixhi_f16 %0, %3, %4, 0\n\t"
        "v_fma_mixlo_f16 %1, %2, %4, -%0 op_sel_hi:[0,0,1]\n\t"
        "v_fma_mixhi_f16 %1, %3, %4, -%0 op_sel:[0,0,1] op_sel_hi:[0,0,1]\n\t"
        "s_nop 1"          // a vector write needs two wait states before an MFMA reads the register as an operand; hipcc pads only its own instructions
        : "=&v"(hi), "=&v"(lo) : "v"(a), "v"(b), "s"(s));
}

#ifdef AWSEG_G3_STAMP
// tools/scratch/g3_stamps.py: s_memtime stamps of block 0, waves 0 and 4, summed over K tiles:
// [wait + barrier, DMA issue at the top, first 16-deep step, (late DMA issue +) second step, K tiles, whole tile loop incl. epilogue]
__device__ unsigned long long g_g3_stamp[2][8];
#define G3_T(v) const unsigned long long v = __builtin_readcyclecounter()
#else
#define G3_T(v)
#endif

struct g3_args {
    const float* x; const uint16_t* w3; const float* bias; const float* residual; float* out;
    const unsigned* trailer;               // {max|w| bits, weight exponent ew, 0, 0}
    int64_t M; int N, K, act, ntm, ntm8, ntn;
    int img_bn;                            // rows of an n-tile of the weight image (>= the block tile's width)
    int rot;                               // measurement switch (AWSEG_G3_ROT): block b starts its K loop at K tile (b * rot) % nkt
    int prio;                              // AWSEG_G3_PRIO (default 0): waves 4-7 at raised priority during the first 16-deep step of a K tile
    int stagger;                           // AWSEG_G3_STAGGER (default 1): waves 4-7 issue their LDS-DMA between the two 16-deep steps of a K tile
    // CONV: x is an NHWC image batch [B, cH, cW, cC] (cC % 32 == 0: a K tile lies inside one tap) and row m = (b, oy, ox) of the
    // A operand is gathered from it by the LDS-DMA's per-lane source address — column k = (ky * ckw + kx) * cC + c is
    // x[b, oy * cs - cp + ky * cd, ox * cs - cp + kx * cd, c], zero outside (awseg_conv_gemm_split_bias_act)
    int cH, cW, cC, cHo, cWo, ckw, cs, cp, cd;
    // ROWS form (awseg_conv_rows_gemm_split_bias_act: the 7x7 stems on 3 input channels): the image has cpitch floats a pixel
    // (< cC), a 'tap' is a run of cC = 32 consecutive floats of one image row (8 pixels x 4 channels) starting at pixel
    // ox * cs - cpx, one tap per kernel row (ckw = 1); the image is zero-padded on the left / right so that a run never leaves its row
    int cpitch, cpx;
    int64_t x_bytes;
    // DUAL: the A operand is [x | x2] along K — columns k < K1 from x (rows of K1 floats), the other K - K1 from x2: rows of
    // K - K1 floats (x2s == 0), or the pixels (b, oy * x2s, ox * x2s) of an NHWC image [B, x2H, x2W, K - K1] for output row
    // m = (b, oy, ox) of an x2Ho x x2Wo grid — a 1x1 convolution of stride x2s on the block's input (awseg_gemm_split_dual_bias_act)
    const float* x2; int K1, x2H, x2W, x2s, x2Ho, x2Wo;
    int64_t x2_bytes;
    // ... and, for plain rows, up to two more pieces of the same width as x2 behind it (x3, x4: [x | x2 | x3 | x4], K2 floats each)
    const float* x3; const float* x4; int K2;
};

// ABL != 0: ablation builds for measurements (wrong results, valid times; AWSEG_G3_ABL): 1 no LDS-DMA in the K loop, 2 no MFMAs,
// 3 no operand split, 4 no activation fragment reads, 5 no weight fragment reads
// BF16: BASELINE config 5 — ONE v_mfma_f32_32x32x16_bf16 per product tile: the activation fragment is rounded to bf16 (RNE) by the
// wave that multiplies it, the weights come as a bf16 image [N/256][KB][256][32] (64-byte rows: a 16 KB stage), float32
// accumulation and epilogue; bf16 has float32's exponent range: no range guard, no scaling.
typedef __bf16 bf8_3 __attribute__((ext_vector_type(8)));
typedef __bf16 bf2_3 __attribute__((ext_vector_type(2)));
typedef float f2_3 __attribute__((ext_vector_type(2)));
__device__ __forceinline__ unsigned pack_bf16_3(float x, float y) { return __builtin_bit_cast(unsigned, __builtin_convertvector(f2_3{x, y}, bf2_3)); }

// NT: MFMA tiles of 32 columns per wave: block tile 256 x (32 NT) — 8 for N % 256 == 0, 4 for N % 128 == 0, 2 for N % 64 == 0 (the
// narrow tiles serve HBM-bound shapes: l1 / l2 conv1, MiT projections; the weight image is cut into n-tiles of the same width)
// WV: waves of a block = 32-row strips of its tile.  8: the 256-row tile, one block per CU (128 KB of LDS).  4: a 128-row tile on
// 64 KB — TWO blocks per CU, so that one block's epilogue (its stores leave a CU at ~15 B/clk: 17 k cycles for a 256 x 256 tile,
// a third of a K = 256 tile's time) runs beside the other block's K loop; the price is NT <= 4 (every activation
// fragment is split once per 128 columns instead of 256) and twice the weight traffic from L2 per product.
template <bool CONV, int ABL = 0, bool BF16 = false, int NT = 8, int WV = 8, bool DUAL = false>
__global__ __launch_bounds__(64 * WV, 2)
void gemm_split3_kernel(g3_args a)
{
    static_assert(NT <= WV, "a weight stage must fit an activation stage");
    static_assert(!DUAL || (!CONV && !BF16), "two A sources: the plain split-operand form only");
    constexpr int G3M = 32 * WV, G3_STAGE = G3M * 128, G3_A0 = 0, G3_B0 = 2 * G3_STAGE;    // A stage s at s * G3_STAGE, B stage s at 2 G3_STAGE + s * G3_BSTAGE
    constexpr int G3_BSTAGE = WV == 8 ? G3_STAGE : 32 * NT * 128;                          // (eight waves: the round's layout; four: the tile's own width)
    constexpr int BN = 32 * NT;
    constexpr int ROWB = BF16 ? 64 : 128;                          // bytes of a weight row in a K tile
    constexpr int NBI_ALL = BN * ROWB / 1024;                      // LDS-DMA instructions per weight K tile (1 KB each)
    constexpr int NBI = NBI_ALL >= WV ? NBI_ALL / WV : 1;          // per wave (fewer than WV in all: the first NBI_ALL waves issue one each)
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    unsigned* sMax = reinterpret_cast<unsigned*>(smem + 2 * G3_STAGE + 2 * G3_BSTAGE);
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int hk = lane >> 5, li = lane & 31;
    const int K = a.K, nkt = (K + G3K - 1) / G3K, kb = nkt;         // K % 32 != 0 (K % 8 == 0): the last K tile is zero-filled past K
    const int ktail = K % G3K;
    const int ntiles = a.ntm8 * a.ntn;

    auto tile_of = [&](int slot, int64_t& m0, int& n0) -> bool {   // gemm_split.hip: XCD-aware persistent walk
        const int xcd = slot & 7, jj = slot >> 3;
        const int nt_i = jj % a.ntn, mt_i = (jj / a.ntn) * 8 + xcd;
        m0 = (int64_t)mt_i * G3M; n0 = nt_i * BN;
        return mt_i < a.ntm;
    };

    // ---- LDS-DMA roles.  One instruction covers 8 rows x 128 B (1 KB, linear in LDS); wave w fills row groups 4w .. 4w+3 of
    // each operand's 256-row stage.  Lane l -> row 8 q + (l >> 3), slot l & 7, which holds source chunk slot ^ ((row >> 1) & 7).
    const int rl = lane >> 3, sl = lane & 7;
    uint32_t a_voff[4], a_voff_last[4], b_voff[4];
    uint32_t a_voff2[4];                                           // DUAL: this lane's rows in the second source
    const int KA = DUAL ? a.K1 : a.K;                              // row length of x
    const int kt1 = KA / G3K;                                      // DUAL: K tiles kt >= kt1 come from x2
    int cby[4], cy0[4], cx0[4];                                    // CONV: image row base b * cH, first tap's input row / column of this lane's A rows (< 0: row past M)
    const uint32_t lds0 = __builtin_amdgcn_readfirstlane(lds_addr3(smem));
    const uint32_t wave_u = __builtin_amdgcn_readfirstlane(wave);   // scalar register: the LDS-DMA base goes through m0
    __amdgpu_buffer_rsrc_t x_rsrc, w_rsrc, x2_rsrc, x3_rsrc, x4_rsrc;
    auto point = [&](int64_t m0, int n0) {
        const int64_t rows_left = a.M - m0;                      // rows past M: out-of-range source -> zeros, never stored
        const int64_t xbytes = rows_left * (int64_t)KA * 4;
        if (CONV) x_rsrc = __builtin_amdgcn_make_buffer_rsrc((void*)a.x, 0, (int)a.x_bytes, 0x00020000);
        else x_rsrc = __builtin_amdgcn_make_buffer_rsrc((void*)(a.x + m0 * KA), 0, (int)(xbytes > 0x7fffffff ? 0x7fffffff : xbytes), 0x00020000);
        if (DUAL) {
            x2_rsrc = __builtin_amdgcn_make_buffer_rsrc((void*)a.x2, 0, (int)a.x2_bytes, 0x00020000);
            x3_rsrc = __builtin_amdgcn_make_buffer_rsrc((void*)(a.x3 ? a.x3 : a.x2), 0, (int)a.x2_bytes, 0x00020000);
            x4_rsrc = __builtin_amdgcn_make_buffer_rsrc((void*)(a.x4 ? a.x4 : a.x2), 0, (int)a.x2_bytes, 0x00020000);
        }
        // the weight image is [n-tile of IB rows][kb K tiles][IB rows][ROWB bytes]; this block's BN rows start at row n0 % IB of n-tile n0 / IB
        const int IB = a.img_bn, nr = n0 % IB;
        w_rsrc = __builtin_amdgcn_make_buffer_rsrc((void*)(a.w3 + ((int64_t)(n0 - nr) * kb + nr) * (ROWB / 2)), 0, (IB * kb - nr) * ROWB, 0x00020000);
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            const int row = 8 * (4 * wave + j) + rl;
            const int c = sl ^ ((row >> 1) & 7);
            a_voff[j] = (uint32_t)(row * KA * 4 + c * 16);
            a_voff_last[j] = (ktail == 0 || c * 4 < ktail) ? a_voff[j] : 0x80000000u;     // chunks past K read zeros (the weight image is zero there too)
            if (DUAL) {
                const int64_t m = m0 + row;
                const int k2 = a.K2;
                int64_t pix = m;                                   // plain rows
                if (a.x2s > 0) {
                    const int b = (int)(m / ((int64_t)a.x2Ho * a.x2Wo));
                    const int rem = (int)(m - (int64_t)b * a.x2Ho * a.x2Wo);
                    const int oy = rem / a.x2Wo, ox = rem - oy * a.x2Wo;
                    pix = ((int64_t)b * a.x2H + oy * a.x2s) * a.x2W + ox * a.x2s;
                }
                a_voff2[j] = m < a.M ? (uint32_t)(pix * k2 * 4 + c * 16) : 0x80000000u;
            }
            {   // weights: instruction wave * NBI + j covers 1 KB = 8 (split: 128-byte rows) or 16 (bf16: 64-byte rows) rows of the K tile
                const int qi = wave * NBI + j;
                if (BF16) {
                    const int brow = 16 * qi + (lane >> 2);
                    b_voff[j] = (uint32_t)(brow * 64 + (((lane & 3) ^ ((brow >> 2) & 3)) * 16));
                } else {
                    const int brow = 8 * qi + rl;
                    b_voff[j] = (uint32_t)(brow * 128 + ((sl ^ ((brow >> 1) & 7)) * 16));
                }
            }
            if (CONV) {
                a_voff[j] = (uint32_t)(c * 16);
                const int64_t m = m0 + row;
                const int b = (int)(m / ((int64_t)a.cHo * a.cWo));
                const int rem = (int)(m - (int64_t)b * a.cHo * a.cWo);
                const int oy = rem / a.cWo, ox = rem - oy * a.cWo;
                cby[j] = m < a.M ? b * a.cH : -1; cy0[j] = oy * a.cs - a.cp; cx0[j] = ox * a.cs - a.cpx;
            }
        }
    };
    auto issue = [&](int kt, int stage) {                          // 8 LDS-DMA instructions per wave and K tile
        const uint32_t la = lds0 + (uint32_t)(G3_A0 + stage * G3_STAGE) + wave_u * 4096u;
        if (CONV) {
            const int k0 = kt * G3K, tap = k0 / a.cC, c0 = k0 - tap * a.cC;     // block-uniform
            const int ky = tap / a.ckw, kx = tap - ky * a.ckw;
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                const int iy = cy0[j] + ky * a.cd, ix = cx0[j] + kx * a.cd;
                const bool ok = cby[j] >= 0 && (unsigned)iy < (unsigned)a.cH && (unsigned)ix < (unsigned)a.cW;
                const uint32_t vo = ok ? (uint32_t)((((cby[j] + iy) * a.cW + ix) * a.cpitch + c0) * 4) + a_voff[j] : 0x80000000u;
                dma16(x_rsrc, vo, 0u, la + (uint32_t)(j * 1024));
            }
        } else {
        if (DUAL && kt >= kt1) {                                   // (block-uniform)
            const int kt2 = a.K2 / G3K;
            const int piece = __builtin_amdgcn_readfirstlane((kt - kt1) / kt2);
            const uint32_t ko = (uint32_t)__builtin_amdgcn_readfirstlane(((kt - kt1) - piece * kt2) * 128);
            if (piece == 0) {
#pragma unroll
                for (int j = 0; j < 4; ++j) dma16(x2_rsrc, a_voff2[j], ko, la + (uint32_t)(j * 1024));
            } else if (piece == 1) {
#pragma unroll
                for (int j = 0; j < 4; ++j) dma16(x3_rsrc, a_voff2[j], ko, la + (uint32_t)(j * 1024));
            } else {
#pragma unroll
                for (int j = 0; j < 4; ++j) dma16(x4_rsrc, a_voff2[j], ko, la + (uint32_t)(j * 1024));
            }
        } else {
#pragma unroll
            for (int j = 0; j < 4; ++j) dma16(x_rsrc, kt == nkt - 1 ? a_voff_last[j] : a_voff[j], (uint32_t)(kt * 128), la + (uint32_t)(j * 1024));
        }
        }
        {
            const uint32_t lbw = lds0 + (uint32_t)(G3_B0 + stage * G3_BSTAGE) + wave_u * (uint32_t)(NBI * 1024);
            if (NBI_ALL >= WV || (int)wave_u < NBI_ALL) {
#pragma unroll
                for (int j = 0; j < NBI; ++j) dma16(w_rsrc, b_voff[j], (uint32_t)(kt * a.img_bn * ROWB), lbw + (uint32_t)(j * 1024));
            }
        }
    };

    // ---- fragment addresses: row li of a 32-row MFMA tile, k = 16 ks + 8 hk .. + 7.  Activations (rows 32 wave + li): float32,
    // chunks 4 ks + 2 hk and + 1; weights (rows 32 j + li): hi chunk 2 ks + hk, lo chunk 4 + 2 ks + hk.  Swizzle key
    // (row >> 1) & 7 = (li >> 1) & 7 for every tile row base (multiples of 32).
    const int sw = (li >> 1) & 7;
    int fa[2][2], fb[2][2];
#pragma unroll
    for (int ks = 0; ks < 2; ++ks) {
#pragma unroll
        for (int e = 0; e < 2; ++e) fa[ks][e] = G3_A0 + (wave * 32 + li) * 128 + (((4 * ks + 2 * hk + e) ^ sw) * 16);
        fb[ks][0] = BF16 ? G3_B0 + li * 64 + (((2 * ks + hk) ^ ((li >> 2) & 3)) * 16) : G3_B0 + li * 128 + (((2 * ks + hk) ^ sw) * 16);
        fb[ks][1] = G3_B0 + li * 128 + (((4 + 2 * ks + hk) ^ sw) * 16);
    }

    int slot = blockIdx.x;
    int64_t m0 = 0; int n0 = 0;
    while (slot < ntiles && !tile_of(slot, m0, n0)) slot += gridDim.x;
    if (slot >= ntiles) return;
    if (tid < 2) sMax[tid] = 0u;
    int par = 0;
    const int we = (int)a.trailer[1];
    float amax = 0.f, sx = kActScale0;
    int xe = kActExp0;
    bool scaled = false;
    bool stores_pending = false;                                  // an epilogue's stores may still be in flight
    int g = 0;                                                    // running K-tile counter: stage = g & 1 across output tiles
    const int kt0 = a.rot ? (int)(((unsigned)blockIdx.x * (unsigned)a.rot) % (unsigned)nkt) : 0;
    auto ktile = [&](int t) { const int u = t + kt0; return u >= nkt ? u - nkt : u; };
#ifdef AWSEG_G3_STAMP
    unsigned long long g3s[6] = {0, 0, 0, 0, 0, 0};
    G3_T(sb0);
#endif
    point(m0, n0);
    issue(ktile(0), 0);

    while (true) {
        int nslot = slot + gridDim.x;
        int64_t nm0 = 0; int nn0 = 0;
        while (nslot < ntiles && !tile_of(nslot, nm0, nn0)) nslot += gridDim.x;
        const bool has_next = nslot < ntiles;

        f32x16 acc[NT];
#pragma unroll
        for (int j = 0; j < NT; ++j)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[j][r] = 0.f;
        amax = 0.f;

        for (int t = 0; t < nkt; ++t, ++g) {
            const int st = g & 1;
            // K tile t (issued one tile ago) has landed for this wave; behind the barrier for every wave — and every wave has
            // finished the MFMAs of tile t-1, whose fragments came from the other stage: it may be refilled now.
            // (first K tile behind an epilogue: the epilogue's stores — 128, or 32 in the transposed form — are the youngest
            // vector-memory operations and the 8 LDS-DMA of this K tile are older than all of them: "at most 63 (32)
            // outstanding" means the DMA have landed, without waiting out the stores' write latency)
            G3_T(s0);
            // (per template: the epilogue issues 16 NT stores — 128 / 64 / 32 — so "all but the min(63, 16 NT) youngest" covers the DMA
            // explicitly; with vmcnt(63) the 32-store tiles would lean on the compiler's wait for the bias loads issued behind the DMA)
            if (t == 0 && stores_pending) asm volatile("s_waitcnt vmcnt(%0)" : : "n"(G3_TRANSPOSED ? 32 : (16 * NT < 63 ? 16 * NT : 63)) : "memory");
            else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            __builtin_amdgcn_s_barrier();
            // (the next output tile's first K tile is issued behind the guard check.)  Issuing its eight LDS-DMA instructions keeps a
            // wave away from its MFMAs for ~800 cycles (tools/probe_wino_stamps.hip: ~100 per instruction): waves 0-3 do it here, their
            // SIMD partners 4-7 between the two 16-deep steps, so a SIMD always has one wave multiplying (AWSEG_G3_STAGGER=0: all here)
            const bool issue_late = a.stagger && wave_u >= 4;
            G3_T(s1);
            if (t + 1 < nkt && ABL != 1 && !issue_late) issue(ktile(t + 1), st ^ 1);
            G3_T(s2);
            if (t == 0 && tid == 0) sMax[par ^ 1] = 0u;
            const unsigned char* sa = smem + st * G3_STAGE;
            const unsigned char* sbw = smem + st * G3_BSTAGE;         // weight fragments (fb carries G3_B0)
#pragma unroll
            for (int ks = 0; ks < 2; ++ks) {
                // The arbiter favours the older wave of a SIMD: the stamps show waves 0-3 done with a K tile after 2 400 cycles and waiting
                // 1 100 at the barrier for their partners, which finish the last third alone (one wave cannot keep the matrix pipe
                // busy).  AWSEG_G3_PRIO=1 runs the FIRST 16-deep step of waves 4-7 at raised priority: 1-4 % on the long-K shapes in
                // isolation (l4 conv1 0.381 -> 0.367 ms), -0.3 % on the whole step in an A/B on one box: off by default.
                if (a.prio && wave_u >= 4) { if (ks == 0) __builtin_amdgcn_s_setprio(1); else __builtin_amdgcn_s_setprio(0); }
#ifdef AWSEG_G3_STAMP
                if (ks == 1) { G3_T(s3); g3s[2] += s3 - s2; g3s[5] = s3; }
#endif
                if (ks == 1 && t + 1 < nkt && ABL != 1 && issue_late) issue(ktile(t + 1), st ^ 1);
                f32x4 p, q;
                if (ABL == 4) { p = f32x4{(float)t, 1.f, 2.f, 3.f}; q = f32x4{(float)ks, 1.f, 2.f, 3.f}; }
                else {
                    p = *reinterpret_cast<const f32x4*>(sa + fa[ks][0]);
                    q = *reinterpret_cast<const f32x4*>(sa + fa[ks][1]);
                }
                h8 Bh[NT], Bl[NT];
#pragma unroll
                for (int j = 0; j < NT; ++j) {
                    if (ABL == 5) { Bh[j] = h8{(_Float16)t, (_Float16)j, 1, 2, 3, 4, 5, 6}; Bl[j] = h8{(_Float16)ks, (_Float16)j, 1, 2, 3, 4, 5, 6}; continue; }
                    if (BF16) { Bh[j] = *reinterpret_cast<const h8*>(sbw + fb[ks][0] + j * 2048); Bl[j] = Bh[j]; continue; }
                    Bh[j] = *reinterpret_cast<const h8*>(sbw + fb[ks][0] + j * 4096);
                    Bl[j] = *reinterpret_cast<const h8*>(sbw + fb[ks][1] + j * 4096);
                }
                u32x4 H, L; unsigned h, l;
                if (BF16) {
                    H[0] = pack_bf16_3(p[0], p[1]); H[1] = pack_bf16_3(p[2], p[3]); H[2] = pack_bf16_3(q[0], q[1]); H[3] = pack_bf16_3(q[2], q[3]);
                    const bf8_3 Ab = __builtin_bit_cast(bf8_3, H);
#pragma unroll
                    for (int j = 0; j < NT; ++j) acc[j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(Ab, __builtin_bit_cast(bf8_3, Bh[j]), acc[j], 0, 0, 0);
                    continue;
                }
                amax = __builtin_fmaxf(__builtin_fmaxf(amax, __builtin_fabsf(p[0])), __builtin_fabsf(p[1]));
                amax = __builtin_fmaxf(__builtin_fmaxf(amax, __builtin_fabsf(p[2])), __builtin_fabsf(p[3]));
                amax = __builtin_fmaxf(__builtin_fmaxf(amax, __builtin_fabsf(q[0])), __builtin_fabsf(q[1]));
                amax = __builtin_fmaxf(__builtin_fmaxf(amax, __builtin_fabsf(q[2])), __builtin_fabsf(q[3]));
                if (ABL == 3) { H = __builtin_bit_cast(u32x4, p); L = __builtin_bit_cast(u32x4, q); }
                else {
                    split2(p[0], p[1], sx, h, l); H[0] = h; L[0] = l;
                    split2(p[2], p[3], sx, h, l); H[1] = h; L[1] = l;
                    split2(q[0], q[1], sx, h, l); H[2] = h; L[2] = l;
                    split2(q[2], q[3], sx, h, l); H[3] = h; L[3] = l;
                }
                const h8 Ah = __builtin_bit_cast(h8, H), Al = __builtin_bit_cast(h8, L);
#pragma unroll
                for (int j = 0; j < NT; ++j) {
                    if (ABL == 2) { asm volatile("" : : "v"(Ah), "v"(Al), "v"(Bh[j]), "v"(Bl[j])); continue; }
                    if (G3_TRANSPOSED) {
                        // transposed product: rows of the accumulator tile = weight rows n, columns (lanes) = activation rows m
                        acc[j] = __builtin_amdgcn_mfma_f32_32x32x16_f16(Bh[j], Ah, acc[j], 0, 0, 0);
                        acc[j] = __builtin_amdgcn_mfma_f32_32x32x16_f16(Bl[j], Ah, acc[j], 0, 0, 0);
                        acc[j] = __builtin_amdgcn_mfma_f32_32x32x16_f16(Bh[j], Al, acc[j], 0, 0, 0);
                    } else {
                        acc[j] = __builtin_amdgcn_mfma_f32_32x32x16_f16(Ah, Bh[j], acc[j], 0, 0, 0);
                        acc[j] = __builtin_amdgcn_mfma_f32_32x32x16_f16(Ah, Bl[j], acc[j], 0, 0, 0);
                        acc[j] = __builtin_amdgcn_mfma_f32_32x32x16_f16(Al, Bh[j], acc[j], 0, 0, 0);
                    }
                }
            }
#ifdef AWSEG_G3_STAMP
            { G3_T(s4); g3s[0] += s1 - s0; g3s[1] += s2 - s1; g3s[3] += s4 - g3s[5]; g3s[4] += 1; }
#endif
        }

        // ---- range guard (gemm_split.hip): the block's max|x| of this pass, through LDS
        if (!BF16 && !scaled && amax >= kSplitLimit3) atomicMax(&sMax[par], __builtin_bit_cast(unsigned, amax));
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        __builtin_amdgcn_s_barrier();                              // also: every wave is done with the last K tile's stage
        const unsigned mx = sMax[par];
        par ^= 1;
        bool again = false;
        if (mx != 0u && !scaled) {
            const int ex = (int)(mx >> 23) & 0xff;
            if (ex != 0xff) {                                      // Inf / NaN: nothing to rescue, let them propagate
                xe = ex - 127 - 13;                                // max|x| * 2^-xe in [2^13, 2^14)
                sx = pow2f3(-xe);
                scaled = true; again = true;
            }
        }
        if (again) {                                               // same tile again, scaled: its first K tile into the free stage
            stores_pending = false;
            issue(ktile(0), g & 1);
            continue;
        }
        // the next output tile's first K tile travels during the epilogue (stage g & 1 was last read two K tiles ago)
        const int64_t em0 = m0; const int en0 = n0;
        if (has_next) { point(nm0, nn0); issue(ktile(0), g & 1); }

        const int oe = BF16 ? 0 : we + xe;
        const int oe1 = oe / 2, oe2 = oe - oe1;
        const float os1 = pow2f3(oe1 < -126 ? -126 : (oe1 > 127 ? 127 : oe1)), os2 = pow2f3(oe2 < -126 ? -126 : (oe2 > 127 ? 127 : oe2));

#if G3_TRANSPOSED
        // ---- epilogue: lane = output row m (32 wave + li), registers 4 g4 .. 4 g4 + 3 of tile j = four consecutive columns
        // 32 j + 8 g4 + 4 hk ..: 16-byte accesses.  BRANCH-FREE on purpose: a missing bias / residual is read through a
        // zero-record descriptor (the hardware returns zeros).  With `has_res ? load : 0` the loads sat in their own basic
        // blocks and hipcc (clang-22) placed a `v_mov v, 0` of a store-data register directly behind a 16-byte store across
        // the block boundary — the store then wrote the zero (seen as exact zeros in column 8 g4 + 2 of some rows, run-dependent).
        {
            const int64_t tile_off = em0 * a.N + en0;
            const int64_t rem = ((int64_t)a.M * a.N - tile_off) * 4;
            const int nrec = rem > 0x7fffffff ? 0x7fffffff : (int)rem;
            const __amdgpu_buffer_rsrc_t o_rsrc = __builtin_amdgcn_make_buffer_rsrc((void*)(a.out + tile_off), 0, nrec, 0x00020000);
            const __amdgpu_buffer_rsrc_t r_rsrc = __builtin_amdgcn_make_buffer_rsrc((void*)((a.residual ? a.residual : a.out) + tile_off), 0, a.residual ? nrec : 0, 0x00020000);
            const __amdgpu_buffer_rsrc_t b_rsrc = __builtin_amdgcn_make_buffer_rsrc((void*)((a.bias ? a.bias : a.out) + en0), 0, a.bias ? BN * 4 : 0, 0x00020000);
            const int voff = ((wave * 32 + li) * a.N + 4 * hk) * 4;
            const float relu_floor = a.act == 1 ? 0.f : -__builtin_inff();
#pragma unroll
            for (int j = 0; j < NT; ++j) {
                f32x4 bv[4], rv[4];
#pragma unroll
                for (int g4 = 0; g4 < 4; ++g4) {
                    bv[g4] = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(b_rsrc, 16 * hk + (j * 32 + 8 * g4) * 4, 0, 0));
                    rv[g4] = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(r_rsrc, voff + (j * 32 + 8 * g4) * 4, 0, 0));
                }
#pragma unroll
                for (int g4 = 0; g4 < 4; ++g4) {
                    f32x4 v;
#pragma unroll
                    for (int qq = 0; qq < 4; ++qq) {
                        float vv = acc[j][4 * g4 + qq] * os1 * os2;
                        vv = vv + bv[g4][qq] + rv[g4][qq];
                        v[qq] = __builtin_fmaxf(vv, relu_floor);
                    }
                    // the column offset rides in the instruction's IMMEDIATE offset (vector offset + constant, scalar offset 0), not in
                    // a scalar register: with an SGPR offset hipcc pads no wait state between a 16-byte store and the next vector
                    // write of its data registers (LLVM's rule: that hazard exists only without an soffset register) — on gfx950
                    // the store then picked up the NEXT group's value in its first dword (wrong, run-dependent outputs)
                    __builtin_amdgcn_raw_buffer_store_b128(__builtin_bit_cast(u32x4, v), o_rsrc, voff + (j * 32 + 8 * g4) * 4, 0, 0);
                }
            }
        }
#else
        // ---- epilogue: lane = output column n (32 j + li), registers = rows m (32 wave + 8 (r >> 2) + 4 hk + (r & 3)): a store
        // instruction writes two full 128-byte lines.  (The transposed form — lane = row, 16-byte accesses, a quarter of the
        // instructions — was measured SLOWER wherever the epilogue dominates (K = 64: 0.62 against 0.48 ms): each of its
        // instructions touches 32 rows x 32 bytes, four times the write requests at the L2 for the same bytes.)
        {
            const int64_t tile_off = em0 * a.N + en0;
            const int64_t rem = ((int64_t)a.M * a.N - tile_off) * 4;
            const int nrec = rem > 0x7fffffff ? 0x7fffffff : (int)rem;
            const __amdgpu_buffer_rsrc_t o_rsrc = __builtin_amdgcn_make_buffer_rsrc((void*)(a.out + tile_off), 0, nrec, 0x00020000);
            const __amdgpu_buffer_rsrc_t r_rsrc = __builtin_amdgcn_make_buffer_rsrc((void*)((a.residual ? a.residual : a.out) + tile_off), 0, a.residual ? nrec : 0, 0x00020000);
            const int ncols = a.N - en0 < BN ? a.N - en0 : BN;     // columns of this tile inside the matrix (N < 64: fewer than BN)
            const __amdgpu_buffer_rsrc_t b_rsrc = __builtin_amdgcn_make_buffer_rsrc((void*)((a.bias ? a.bias : a.out) + en0), 0, a.bias ? ncols * 4 : 0, 0x00020000);
            const int voff = ((wave * 32 + 4 * hk) * a.N + li) * 4;
            int vcol[NT];                                          // a lane's column of tile j: past N -> an offset no descriptor covers (loads 0, stores dropped)
#pragma unroll
            for (int j = 0; j < NT; ++j) vcol[j] = j * 32 + li < ncols ? voff + j * 128 : (int)0x80000000;
            const float relu_floor = a.act == 1 ? 0.f : -__builtin_inff();
            const int rowb = a.N * 4;                              // bytes per output row
            const bool has_res = a.residual != nullptr;           // without one the 16 residual loads per column tile are not issued (-3 % at K = 256)
            // (Measured and dropped: the tile through this wave's 8 KB of the free stage to FULL-ROW 16-byte stores — a quarter of the
            // store instructions, whole 128-byte lines per instruction: 5-25 % SLOWER on every epilogue-heavy shape (K = 64: 0.47 ->
            // 0.59 ms).  The stamps put ~23 k cycles of a 56 k-cycle K = 256 tile into the epilogue: that is 256 KB at the ~10 B/clk a CU
            // gets of the HBM write rate when every CU writes, drained inside the next tile's first two K-tile waits — vmcnt counts
            // loads and stores together on gfx9, so a wave cannot wait for its LDS-DMA without waiting for its stores.)
            // gfx9's vmcnt counts loads and stores in ONE in-order counter: a load whose value is needed after some stores were issued
            // makes the wave wait for those stores' write acknowledgements.  The first build fetched the bias of a column tile behind
            // the previous tile's 16 stores (eight such waits per output tile) — all NT bias values are fetched before the first
            // store now, and handed on through an asm move so that hipcc stops tying later uses to the load; the residual of row
            // group g + 1 is requested BEFORE the stores of group g, so its wait covers loads only.
            float bv[NT], bld[NT];
#pragma unroll
            for (int j = 0; j < NT; ++j) bld[j] = __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(b_rsrc, (j * 32 + li) * 4, 0, 0));
#pragma unroll
            for (int j = 0; j < NT; ++j) asm volatile("v_mov_b32 %0, %1" : "=v"(bv[j]) : "v"(bld[j]));
            auto res_load = [&](int g, float (&rv)[8]) {             // row group g = (column tile g >> 1, rows 8 (g & 1) .. + 7 of the register file)
                const int j = g >> 1, half = g & 1;
#pragma unroll
                for (int r8 = 0; r8 < 8; ++r8) {
                    const int r = 8 * half + r8, rowc = (r & 3) + 8 * (r >> 2);
                    rv[r8] = __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(r_rsrc, vcol[j], rowc * rowb, 0));
                }
            };
            auto group_store = [&](int g, const float (&rv)[8], bool with_res) {
                const int j = g >> 1, half = g & 1;
#pragma unroll
                for (int r8 = 0; r8 < 8; ++r8) {
                    const int r = 8 * half + r8, rowc = (r & 3) + 8 * (r >> 2);
                    float vv = acc[j][r] * os1 * os2;
                    vv = vv + bv[j];
                    if (with_res) vv = vv + rv[r8];
                    vv = __builtin_fmaxf(vv, relu_floor);
                    __builtin_amdgcn_raw_buffer_store_b32(__builtin_bit_cast(uint32_t, vv), o_rsrc, vcol[j], rowc * rowb, 0);
                }
            };
            if (has_res) {                                          // block-uniform (dword accesses: none of the 16-byte store hazards of DESIGN.md 10a)
                float rva[8], rvb[8];
                res_load(0, rva);
#pragma unroll
                for (int g = 0; g < 2 * NT; g += 2) {
                    res_load(g + 1, rvb);
                    group_store(g, rva, true);
                    if (g + 2 < 2 * NT) res_load(g + 2, rva);
                    group_store(g + 1, rvb, true);
                }
            } else {
                const float none[8] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
#pragma unroll
                for (int g = 0; g < 2 * NT; ++g) group_store(g, none, false);
            }
        }
#endif
        if (!has_next) break;
        stores_pending = true;
        slot = nslot; m0 = nm0; n0 = nn0;
        scaled = false; sx = kActScale0; xe = kActExp0;
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
#ifdef AWSEG_G3_STAMP
    if (blockIdx.x == 0 && (tid == 0 || tid == 256)) {
        G3_T(sb1);
        for (int i = 0; i < 5; ++i) g_g3_stamp[tid >> 8][i] += g3s[i];
        g_g3_stamp[tid >> 8][5] += sb1 - sb0;
    }
#endif
}

// weights float32 [N][K] (normalised by 2^-e, the trailer's exponent) -> the k-blocked image [N/256][KB][256][32 hi | 32 lo],
// KB = ceil(K / 32), zeros past K: the K tile of a 256-row n-tile is one contiguous 32 KB block (a tile's rows at stride K * 4 B
// would all fall into the same few L2 channels)
__global__ __launch_bounds__(256)
void split3_weights_kernel(const float* __restrict__ w, int n_rows, int k_dim, int kb, int bn, uint16_t* __restrict__ out, const unsigned* __restrict__ trailer)
{
    const int e = (int)trailer[1];
    const int e1 = e / 2, e2 = e - e1;
    const float s1 = pow2f3(-e1), s2 = pow2f3(-e2);
    const int64_t i = ((int64_t)blockIdx.x * 256 + threadIdx.x) * 2;        // padded element pair (n, k), (n, k + 1), k < 32 KB
    const int kp = kb * 32;
    if (i >= (int64_t)n_rows * kp) return;
    const int64_t n = i / kp;
    const int k = (int)(i - n * kp);
    const float x0 = k < k_dim ? w[n * k_dim + k] * s1 * s2 : 0.f, x1 = k + 1 < k_dim ? w[n * k_dim + k + 1] * s1 * s2 : 0.f;
    const _Float16 h0 = (_Float16)x0, h1 = (_Float16)x1;
    const _Float16 l0 = (_Float16)(x0 - (float)h0), l1 = (_Float16)(x1 - (float)h1);
    uint16_t* d = out + (((n / bn) * kb + (k >> 5)) * bn + (n % bn)) * 64 + (k & 31);
    d[0] = __builtin_bit_cast(uint16_t, h0); d[1] = __builtin_bit_cast(uint16_t, h1);
    d[32] = __builtin_bit_cast(uint16_t, l0); d[33] = __builtin_bit_cast(uint16_t, l1);
}

__global__ __launch_bounds__(256)
void bf16_3_weights_kernel(const float* __restrict__ w, int n_rows, int k_dim, int kb, int bn, uint16_t* __restrict__ out)
{
    const int64_t i = ((int64_t)blockIdx.x * 256 + threadIdx.x) * 2;
    const int kp = kb * 32;
    if (i >= (int64_t)n_rows * kp) return;
    const int64_t n = i / kp;
    const int k = (int)(i - n * kp);
    const unsigned pr = pack_bf16_3(k < k_dim ? w[n * k_dim + k] : 0.f, k + 1 < k_dim ? w[n * k_dim + k + 1] : 0.f);
    uint16_t* d = out + (((n / bn) * kb + (k >> 5)) * bn + (n % bn)) * 32 + (k & 31);
    d[0] = (uint16_t)pr; d[1] = (uint16_t)(pr >> 16);
}

}  // namespace

// width of the block tile (and of the weight image's n-tiles) for an N-column problem; 0: not served
// (N < 64, any width from 8 up: ONE 64-column tile whose columns past N are never stored — the 48-channel skip projection
// and the 19-class head of DeepLabV3+, MiT stage 1's 32-channel projections: HBM-bound launches of 10^6 rows)
// (any other N >= 8: 64-column tiles, the last one masked the same way — MiT stage 3's 160 channels are 2.5 tiles)
int awseg_gemm_split3_bn(int n) { return n % 256 == 0 ? 256 : (n % 128 == 0 ? 128 : (n >= 8 ? 64 : 0)); }
// rows of the weight image: N rounded up to whole n-tiles (the rows past N are zero)
int64_t awseg_gemm_split3_image_rows(int n) { const int bn = awseg_gemm_split3_bn(n); return bn ? (int64_t)(n + bn - 1) / bn * bn : 0; }

int awseg_gemm_bf16_3_weights(const float* w, int n, int k, uint16_t* w3, hipStream_t stream)
{
    const int kb = (k + 31) / 32;
    const int64_t ne = (int64_t)n * kb * 32;
    hipLaunchKernelGGL(bf16_3_weights_kernel, dim3((unsigned)((ne / 2 + 255) / 256)), dim3(256), 0, stream, w, n, k, kb, awseg_gemm_split3_bn(n), w3);
    AWSEG_LAUNCH_CHECK();
    return 0;
}

// Called by awseg_gemm_split_weights once the classic image and the trailer {max|w| bits, ew} are written (same stream).
int awseg_gemm_split3_weights(const float* w, int n, int k, uint16_t* w3, const unsigned* trailer, hipStream_t stream)
{
    const int kb = (k + 31) / 32;
    const int64_t ne = (int64_t)n * kb * 32;
    const int64_t rows = awseg_gemm_split3_image_rows(n);
    if (rows != n) {                                               // a last n-tile with rows past N: they stay zero
        hipError_t e = hipMemsetAsync(w3, 0, (size_t)rows * kb * 32 * 2 * sizeof(uint16_t), stream);
        if (e != hipSuccess) return (int)e;
    }
    hipLaunchKernelGGL(split3_weights_kernel, dim3((unsigned)((ne / 2 + 255) / 256)), dim3(256), 0, stream, w, n, k, kb, awseg_gemm_split3_bn(n), w3, trailer);
    AWSEG_LAUNCH_CHECK();
    return 0;
}

bool awseg_gemm_split3_eligible(int64_t m, int n, int k, const void* x, const void* out, const void* residual, const void* bias)
{
    if (awseg_gemm_split3_bn(n) == 0 || k % 8 || k < 32 || m < 1) return false;
    if (((uintptr_t)x | (uintptr_t)out | (uintptr_t)residual | (uintptr_t)bias) & 15) return false;
    if ((int64_t)G3M_MAX * n * 4 > 0x7fffffff || (int64_t)G3M_MAX * (k + 32) * 4 > 0x7fffffff) return false;
    return true;
}

int awseg_gemm_split3_launch(const float* x, const uint16_t* w3, const unsigned* trailer, const float* bias, const float* residual,
                             int act, float* out, int64_t m, int n, int k, int cus, hipStream_t stream, const int* conv, bool bf16,
                             const awseg_g3_dual* dual)
{
    g3_args a;
    a.x2 = nullptr; a.K1 = k; a.x2H = a.x2W = a.x2Ho = a.x2Wo = 1; a.x2s = 0; a.x2_bytes = 0; a.x3 = a.x4 = nullptr; a.K2 = 0;
    if (dual) {
        const int pieces = 1 + (dual->x3 ? 1 : 0) + (dual->x4 ? 1 : 0);
        if (conv || bf16 || dual->k1 % G3K || dual->k1 < G3K || (k - dual->k1) % pieces || dual->bytes > 0x7fffffff) return AWSEG_ERANGE;
        const int k2 = (k - dual->k1) / pieces;
        if (k2 % G3K || k2 < G3K || (pieces > 1 && dual->stride > 0) || (dual->x4 && !dual->x3)) return AWSEG_ERANGE;
        a.x2 = dual->x2; a.K1 = dual->k1; a.x2H = dual->h; a.x2W = dual->w; a.x2s = dual->stride; a.x2Ho = dual->ho; a.x2Wo = dual->wo;
        a.x2_bytes = dual->bytes; a.x3 = dual->x3; a.x4 = dual->x4; a.K2 = k2;
    }
    a.cH = a.cW = a.cC = a.cHo = a.cWo = a.ckw = a.cs = 1; a.cp = 0; a.cd = 1; a.x_bytes = 0; a.cpitch = 1; a.cpx = 0;
    if (conv) {                                                   // {H, W, C, Ho, Wo, kw, stride, pad, dil, batch, pixel pitch (0: C), pad x (-1: pad)}
        a.cH = conv[0]; a.cW = conv[1]; a.cC = conv[2]; a.cHo = conv[3]; a.cWo = conv[4]; a.ckw = conv[5]; a.cs = conv[6]; a.cp = conv[7]; a.cd = conv[8];
        a.cpitch = conv[10] > 0 ? conv[10] : a.cC; a.cpx = conv[11] >= 0 ? conv[11] : a.cp;
        a.x_bytes = (int64_t)conv[9] * a.cH * a.cW * a.cpitch * 4;
        if (a.cC % G3K || a.x_bytes > 0x7fffffff) return AWSEG_ERANGE;    // checked by the caller (eligibility)
    }
    static int rot = -1;
    if (rot < 0) { const char* e = getenv("AWSEG_G3_ROT"); rot = e ? atoi(e) : 0; }
    a.rot = rot;
    static int stagger = -1;
    if (stagger < 0) { const char* e = getenv("AWSEG_G3_STAGGER"); stagger = e ? atoi(e) : 1; }
    a.stagger = stagger;
    static int prio = -1;
    if (prio < 0) { const char* e = getenv("AWSEG_G3_PRIO"); prio = e ? atoi(e) : 0; }
    a.prio = prio;
    a.x = x; a.w3 = w3; a.bias = bias; a.residual = residual; a.out = out; a.trailer = trailer;
    a.M = m; a.N = n; a.K = k; a.act = act;
    const int img_bn = awseg_gemm_split3_bn(n);
    a.img_bn = img_bn;
    // half-height tiles, two blocks per CU (WV = 4): where a tile's epilogue weighs against its K loop and the narrower tile costs
    // little — measured per shape (kernel_bench, one box, 256-row -> 128-row): K = 64 N = 256 0.498 -> 0.442 ms, K = 128 N = 512
    // 0.269 -> 0.245, N = 64 K = 256 0.253 -> 0.238, N = 128 K = 512 0.152 -> 0.142; against it K = 256 / 304 N = 256 0.513 -> 0.59 /
    // 0.625 -> 0.70 (the activation split per 128 instead of 256 columns and twice the weight reads outweigh the overlap) and every
    // long-K shape (K = 2048: +20 %).  AWSEG_G3_HALF=0 turns it off, =2 forces it on every shape (tests).
    static int half_mode = -1;
    if (half_mode < 0) { const char* e = getenv("AWSEG_G3_HALF"); half_mode = e ? atoi(e) : 1; }
    const int bn_half = img_bn >= 128 ? 128 : 64;
    // ... and, whatever K, where 256-row tiles would leave CUs without a block (M = 16 384: MiT stage 4, the key / value projections)
    const bool few = ((m + 255) / 256) * (int64_t)((n + img_bn - 1) / img_bn) < (int64_t)cus;
    const bool half = !bf16 && (half_mode == 2 || (half_mode == 1 && (few || ((k <= 128 || (img_bn <= 128 && k <= 512)) &&
                                                   ((m + 127) / 128) * (int64_t)((n + bn_half - 1) / bn_half) >= 4 * (int64_t)cus))));
    const int rows = half ? 128 : 256;
    const int bn = half ? bn_half : img_bn;
    const int64_t ntm = (m + rows - 1) / rows;
    a.ntn = (n + bn - 1) / bn;                                     // N < 64: one tile, its columns past N masked in the epilogue
    const int64_t ntm8 = (ntm + 7) / 8 * 8;
    if (ntm8 * a.ntn > 0x7fffffff) return AWSEG_ERANGE;
    a.ntm = (int)ntm; a.ntm8 = (int)ntm8;
    const int64_t slots = ntm8 * a.ntn;
    static int three = -1;                                          // AWSEG_G3_THREE=0: two four-wave blocks per CU for the 64-column tiles too
    if (three < 0) { const char* e = getenv("AWSEG_G3_THREE"); three = e ? atoi(e) : 1; }
    int64_t blocks = (int64_t)cus * (half ? ((bn == 64 && three) ? 3 : 2) : 1) / 8 * 8;
    if (blocks < 8) blocks = 8;
    if (blocks > slots) blocks = slots;
    static int abl = -1;
    if (abl < 0) { const char* e = getenv("AWSEG_G3_ABL"); abl = e ? atoi(e) : 0; }
    if (abl && !conv && !bf16 && !dual && bn == 256) {
#define G3_ABL(n) case n: { auto kf = gemm_split3_kernel<false, n>; (void)hipFuncSetAttribute(reinterpret_cast<const void*>(kf), hipFuncAttributeMaxDynamicSharedMemorySize, g3_lds(8, 8)); hipLaunchKernelGGL(kf, dim3((unsigned)blocks), dim3(512), g3_lds(8, 8), stream, a); break; }
        switch (abl) { G3_ABL(1) G3_ABL(2) G3_ABL(3) G3_ABL(4) G3_ABL(5) default: break; }
#undef G3_ABL
        AWSEG_LAUNCH_CHECK();
        return 0;
    }
#define G3_GO(CONV_, BF_, NT_, WV_)                                                                                                    \
    do {                                                                                                                              \
        auto kf = gemm_split3_kernel<CONV_, 0, BF_, NT_, WV_>;                                                                        \
        static bool attr = false;                                                                                                     \
        if (!attr) {                                                                                                                  \
            hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(kf), hipFuncAttributeMaxDynamicSharedMemorySize, g3_lds(WV_, NT_)); \
            if (e != hipSuccess) return (int)e;                                                                                       \
            attr = true;                                                                                                              \
        }                                                                                                                             \
        hipLaunchKernelGGL(kf, dim3((unsigned)blocks), dim3(64 * WV_), g3_lds(WV_, NT_), stream, a);                                       \
    } while (0)
#define G3_BY_NT(CONV_, BF_) do { if (bn == 256) G3_GO(CONV_, BF_, 8, 8); else if (bn == 128) G3_GO(CONV_, BF_, 4, 8); else G3_GO(CONV_, BF_, 2, 8); } while (0)
#define G3_HALF(CONV_) do { if (bn == 128) G3_GO(CONV_, false, 4, 4); else G3_GO(CONV_, false, 2, 4); } while (0)
#define G3_GO_DUAL(NT_, WV_)                                                                                                            \
    do {                                                                                                                              \
        auto kf = gemm_split3_kernel<false, 0, false, NT_, WV_, true>;                                                                \
        static bool attr = false;                                                                                                     \
        if (!attr) {                                                                                                                  \
            hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(kf), hipFuncAttributeMaxDynamicSharedMemorySize, g3_lds(WV_, NT_)); \
            if (e != hipSuccess) return (int)e;                                                                                       \
            attr = true;                                                                                                              \
        }                                                                                                                             \
        hipLaunchKernelGGL(kf, dim3((unsigned)blocks), dim3(64 * WV_), g3_lds(WV_, NT_), stream, a);                                       \
    } while (0)
    if (dual) {
        if (half) { if (bn == 128) G3_GO_DUAL(4, 4); else G3_GO_DUAL(2, 4); }
        else if (bn == 256) G3_GO_DUAL(8, 8); else if (bn == 128) G3_GO_DUAL(4, 8); else G3_GO_DUAL(2, 8);
    }
    else if (bf16) G3_BY_NT(false, true);
    else if (half && conv) G3_HALF(true);
    else if (half) G3_HALF(false);
    else if (conv) G3_BY_NT(true, false);
    else G3_BY_NT(false, false);
#undef G3_GO_DUAL
#undef G3_HALF
#undef G3_BY_NT
#undef G3_GO
    AWSEG_LAUNCH_CHECK();
    return 0;
}

#ifdef AWSEG_G3_STAMP
AWSEG_API int awseg_debug_g3_stamps(unsigned long long* out16, int reset)
{
    hipError_t e = hipMemcpyFromSymbol(out16, HIP_SYMBOL(g_g3_stamp), sizeof(unsigned long long) * 16);
    if (e == hipSuccess && reset) { unsigned long long z[16] = {}; e = hipMemcpyToSymbol(HIP_SYMBOL(g_g3_stamp), z, sizeof z); }
    return (int)e;
}
#endif
