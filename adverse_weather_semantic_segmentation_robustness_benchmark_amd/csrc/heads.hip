// heads.hip — A8 SegFormer head with the x32 upsample fused away, A9 ASPP depthwise fusion.
//
// A8 (PKG/models/model.py:209-214): F.interpolate(bilinear, align_corners=False) ->
// Conv3x3(pad 1) -> BatchNorm(eval) -> ReLU -> Conv1x1.  Upsample and 3x3 are linear, so
//     conv3x3(up(f))[o,y,x] = sum_{tap} [tap inside image] sum_{4 cells} wy*wx * G[tap][cell][o]
// with G = the nine 1x1 products W_tap . f at the encoder's resolution (a 2.4 GFLOP GEMM the
// caller runs once).  The 2.15 GB / image full-resolution 256-channel tensor and 2.47 TFLOP /
// image of the as-written op never exist; per pixel we spend 36 gathers x Cmid + Cmid x Cout.
#include "awseg_common.h"
#include <cstdlib>
#include <stdlib.h>

namespace {

constexpr int kThreads = 256;
constexpr int HPX = 32;            // pixels of one row handled per block

// torch area_pixel_compute_source_index(align_corners=False): src = max(scale*(dst+.5)-.5, 0)
struct src_idx { int i0, i1; float l0, l1; };
__device__ __forceinline__ src_idx bilinear_src(int dst, float scale, int in_size)
{
    float s = scale * ((float)dst + 0.5f) - 0.5f;
    if (s < 0.f) s = 0.f;
    src_idx r;
    r.i0 = (int)s;
    if (r.i0 > in_size - 1) r.i0 = in_size - 1;
    r.i1 = r.i0 + (r.i0 < in_size - 1 ? 1 : 0);
    r.l1 = s - (float)r.i0;
    r.l0 = 1.f - r.l1;
    return r;
}

// v1 (VALU): block = one row segment of HPX pixels; threads sweep the mid channels with
// coalesced G reads, stage relu(bn(mid)) in LDS, then (pixel, class) dot products.
__global__ __launch_bounds__(kThreads)
void segformer_head_kernel(const float* __restrict__ g9, int cmid, int h, int w, int H, int W,
                           const float* __restrict__ scale, const float* __restrict__ shift,
                           const float* __restrict__ w2, const float* __restrict__ b2, int cout,
                           float* __restrict__ out)
{
    extern __shared__ float s_mid[];            // [HPX][cmid]
    __shared__ int s_cell[HPX][36];
    __shared__ float s_coef[HPX][36];
    const int b = blockIdx.z, y = blockIdx.y, x0 = blockIdx.x * HPX;
    const float sh = (float)h / (float)H, sw = (float)w / (float)W;
    const float* g = g9 + (int64_t)b * h * w * 9 * cmid;
    // per (pixel, tap, neighbour): flat offset into g (cell*9+tap)*cmid and weight (0 when padded)
    for (int i = threadIdx.x; i < HPX * 36; i += kThreads) {
        int px = i / 36, r = i - px * 36;
        int tap = r >> 2, nb = r & 3;
        int ky = tap / 3, kx = tap - ky * 3;
        int yy = y + ky - 1, xx = x0 + px + kx - 1;
        float cf = 0.f; int cell = 0;
        if (yy >= 0 && yy < H && xx >= 0 && xx < W && x0 + px < W) {
            src_idx sy = bilinear_src(yy, sh, h), sx = bilinear_src(xx, sw, w);
            int ci = (nb & 2) ? sy.i1 : sy.i0, cj = (nb & 1) ? sx.i1 : sx.i0;
            cf = ((nb & 2) ? sy.l1 : sy.l0) * ((nb & 1) ? sx.l1 : sx.l0);
            cell = (ci * w + cj) * 9 + tap;
        }
        s_cell[px][r] = cell; s_coef[px][r] = cf;
    }
    __syncthreads();
    for (int o = threadIdx.x; o < cmid; o += kThreads) {
        const float sc = scale[o], sf = shift[o];
        for (int px = 0; px < HPX; ++px) {
            float acc = 0.f;
#pragma unroll 4
            for (int r = 0; r < 36; ++r) acc = fmaf(s_coef[px][r], g[(int64_t)s_cell[px][r] * cmid + o], acc);
            float v = fmaf(acc, sc, sf);
            s_mid[px * cmid + o] = v > 0.f ? v : 0.f;
        }
    }
    __syncthreads();
    const int64_t HW = (int64_t)H * W;
    for (int i = threadIdx.x; i < HPX * cout; i += kThreads) {
        int k = i / HPX, px = i - k * HPX;
        if (x0 + px >= W) continue;
        float acc = b2[k];
        const float* wk = w2 + (int64_t)k * cmid;
        const float* m = s_mid + px * cmid;
        for (int o = 0; o < cmid; ++o) acc = fmaf(wk[o], m[o], acc);
        out[((int64_t)b * cout + k) * HW + (int64_t)y * W + x0 + px] = acc;
    }
}

// ---------------------------------------------------------------------------------------
// v2 (MFMA): the same arithmetic as two chained fp32 GEMMs per 32-pixel row segment, on
// v_mfma_f32_32x32x2_f32 (exact fp32 = fmaf chain, 157 TF peak).  Block = 32 px x R rows,
// 8 waves; every wave owns whole row segments.
//
//   stage 0  the <= 3 x 4 low-res cells the tile can touch (x 9 taps x Cmid) are copied once
//            into LDS (108 KB for Cmid = 256) — all later gathers are conflict-free ds_reads.
//   GEMM 1   mid^T[o, px] = sum_k T[o, k] * cx[k, px],  k = (kx, cell column) in 12 slots:
//            T[o,k] = sum_ky wy * G[cell row][cell col k][tap ky,kx][o]  (vertical taps folded on
//            the VALU into the A operand), cx = horizontal bilinear weights (B operand, fixed
//            per block).  Zero padding of the 3x3 = dropped taps.
//   epilogue BatchNorm(eval)+ReLU on the accumulator registers.
//   GEMM 2   logits^T[cls, px] = sum_o W2[cls, o] * mid^T[o, px]: the accumulator tile of GEMM 1
//            IS the B operand (lane = pixel column, register = k), so nothing moves through LDS;
//            the k order is the accumulator's row order and W2 is pre-permuted into registers.
// Per 32 pixels: 6*OT + 16*OT MFMAs (OT = Cmid/32), i.e. 176 x 64 cycles for Cmid = 256.
// ---------------------------------------------------------------------------------------
typedef float f32x16 __attribute__((ext_vector_type(16)));
template <int OT, bool CLASSIFY, int NW>
__global__ __launch_bounds__(NW * 64, NW / 4)
void head_mfma_kernel(const float* __restrict__ g9, int h, int w, int H, int W, int R,
                      const float* __restrict__ scale, const float* __restrict__ shift,
                      const float* __restrict__ w2, const float* __restrict__ b2, int cout,
                      float* __restrict__ out, int out_nhwc)
{
    constexpr int CM = OT * 32;
    constexpr int kHeadThreads = NW * 64;
    extern __shared__ float smem[];
    float* Gl = smem;                       // [3 cell rows][4 cell cols][9 taps][CM]
    float* s_scale = smem + 108 * CM;
    float* s_shift = s_scale + CM;
    const int tid = threadIdx.x, lane = tid & 63, wv = tid >> 6;
    const int lo = lane & 31, hh = lane >> 5;
    const int b = blockIdx.z, y0 = blockIdx.y * R, x0 = blockIdx.x * 32;
    const float sh = (float)h / (float)H, sw = (float)w / (float)W;
    const int ibase = bilinear_src(y0 > 0 ? y0 - 1 : 0, sh, h).i0;
    const int jbase = bilinear_src(x0 > 0 ? x0 - 1 : 0, sw, w).i0;

    const float* g = g9 + (int64_t)b * h * w * 9 * CM;
    constexpr int CELL4 = 9 * CM / 4;       // float4 per cell
    for (int i = tid; i < 12 * CELL4; i += kHeadThreads) {
        int cell = i / CELL4, q = i - cell * CELL4;
        int ci = ibase + (cell >> 2), cj = jbase + (cell & 3);
        if (ci > h - 1) ci = h - 1;
        if (cj > w - 1) cj = w - 1;
        reinterpret_cast<float4*>(Gl)[i] = reinterpret_cast<const float4*>(g + ((int64_t)ci * w + cj) * 9 * CM)[q];
    }
    const bool has_scale = (scale != nullptr);
    for (int i = tid; i < CM; i += kHeadThreads) { s_scale[i] = has_scale ? scale[i] : 1.f; s_shift[i] = shift[i]; }

    // B operand of GEMM 1: horizontal weights, lane (hh, lo = pixel), k = 2s + hh -> (kx, cell col)
    float cx[6];
#pragma unroll
    for (int s = 0; s < 6; ++s) {
        const int k = 2 * s + hh, kx = k >> 2, c = k & 3;
        const int xx = x0 + lo + kx - 1;
        float v = 0.f;
        if (xx >= 0 && xx < W && x0 + lo < W) {
            src_idx sx = bilinear_src(xx, sw, w);
            if (sx.i0 - jbase == c) v += sx.l0;
            if (sx.i1 - jbase == c) v += sx.l1;
        }
        cx[s] = v;
    }
    // A operand of GEMM 2: W2 in the accumulator's row order, one float per (o-tile, k-step, lane),
    // lane = (kk = hh, cls = lo); kept in LDS (32 KB at Cmid = 256) to stay under 256 VGPRs
    float* s_w2 = s_shift + CM;             // [OT][16][64]
    if (CLASSIFY) {
        for (int i = tid; i < OT * 16 * 64; i += kHeadThreads) {
            const int ln = i & 63, s2 = (i >> 6) & 15, ot = i >> 10;
            const int cls = ln & 31, kk = ln >> 5;
            const int o = ot * 32 + (s2 & 3) + 8 * (s2 >> 2) + 4 * kk;
            s_w2[i] = (cls < cout) ? w2[(int64_t)cls * CM + o] : 0.f;
        }
    }
    __syncthreads();

    // classifier bias of the 16 classes this lane's accumulator rows hold
    float b2v[16];
#pragma unroll
    for (int r = 0; r < 16; ++r) {
        const int cls = (r & 3) + 8 * (r >> 2) + 4 * hh;
        b2v[r] = (CLASSIFY && cls < cout) ? b2[cls] : 0.f;
    }
    // lane part of the A-operand gather address: (cell col c, tap column kx) of k-slot 2s+hh, + o_local
    int lbase[6];
#pragma unroll
    for (int s = 0; s < 6; ++s) {
        const int k = 2 * s + hh;
        lbase[s] = ((k & 3) * 9 + (k >> 2)) * CM + lo;
    }
    const int64_t HW = (int64_t)H * W;
    for (int ry = wv; ry < R; ry += kHeadThreads / 64) {
        const int y = y0 + ry;
        if (y >= H) break;
        // vertical taps of this row: wave-uniform scalars; an out-of-image tap keeps a valid address
        // and gets zero weights, so the gather below is branch-free
        int ub[6]; float lw[6];
#pragma unroll
        for (int ky = 0; ky < 3; ++ky) {
            const int yy = y + ky - 1;
            const bool ok = (yy >= 0 && yy < H);
            src_idx sy = bilinear_src(ok ? yy : 0, sh, h);
            const int r0 = __builtin_amdgcn_readfirstlane(sy.i0 - ibase), r1 = __builtin_amdgcn_readfirstlane(sy.i1 - ibase);
            ub[2 * ky] = (r0 * 36 + ky * 3) * CM;
            ub[2 * ky + 1] = (r1 * 36 + ky * 3) * CM;
            lw[2 * ky] = __builtin_bit_cast(float, __builtin_amdgcn_readfirstlane(__builtin_bit_cast(int, ok ? sy.l0 : 0.f)));
            lw[2 * ky + 1] = __builtin_bit_cast(float, __builtin_amdgcn_readfirstlane(__builtin_bit_cast(int, ok ? sy.l1 : 0.f)));
        }
        auto gather = [&](int ot, float* t) {
#pragma unroll
            for (int s = 0; s < 6; ++s) {
                const float* base = Gl + lbase[s] + ot * 32;
                float v = 0.f;
#pragma unroll
                for (int j = 0; j < 6; ++j) v = fmaf(lw[j], base[ub[j]], v);
                t[s] = v;
            }
        };
        f32x16 acc2;
#pragma unroll
        for (int r = 0; r < 16; ++r) acc2[r] = 0.f;
        // Software pipeline over the o-tiles with two accumulator tiles: while tile `ot` goes through
        // BatchNorm/ReLU (VALU) and GEMM 2, GEMM 1 of tile ot+1 is already in the matrix pipe and the
        // A operands of tile ot+2 are being gathered from LDS.
        // accumulators start at the folded BatchNorm shift of their rows (the C input of the first MFMA)
        auto gemm1 = [&](int ot, const float* t) {
            f32x16 a;
#pragma unroll
            for (int r = 0; r < 16; ++r) a[r] = has_scale ? 0.f : s_shift[ot * 32 + (r & 3) + 8 * (r >> 2) + 4 * hh];
#pragma unroll
            for (int s = 0; s < 6; ++s) a = __builtin_amdgcn_mfma_f32_32x32x2f32(t[s], cx[s], a, 0, 0, 0);
            return a;
        };
        auto finish = [&](int ot, f32x16& acc) {
            float wf[16];
            if (CLASSIFY) {
#pragma unroll
                for (int r = 0; r < 16; ++r) wf[r] = s_w2[(ot * 16 + r) * 64 + lane];
            }
            const bool relu = !(out_nhwc & 2);                  // bit 1 of the layout flag: linear output (training forward)
            if (has_scale) {
#pragma unroll
                for (int r = 0; r < 16; ++r) {
                    const int o = ot * 32 + (r & 3) + 8 * (r >> 2) + 4 * hh;
                    float v = fmaf(acc[r], s_scale[o], s_shift[o]);
                    acc[r] = (v > 0.f || !relu) ? v : 0.f;
                }
            } else if (relu) {
#pragma unroll
                for (int r = 0; r < 16; ++r) acc[r] = acc[r] > 0.f ? acc[r] : 0.f;
            }
            if (CLASSIFY) {
#pragma unroll
                for (int s2 = 0; s2 < 16; ++s2) acc2 = __builtin_amdgcn_mfma_f32_32x32x2f32(wf[s2], acc[s2], acc2, 0, 0, 0);
            } else if (x0 + lo < W) {
                if (out_nhwc & 1) {
                    // a lane holds channels {8q+4hh .. 8q+4hh+3} of its pixel: four 16-byte stores; the two
                    // half-waves fill adjacent halves of every 32-byte run
                    float* px = out + (((int64_t)b * H + y) * W + x0 + lo) * CM + ot * 32 + 4 * hh;
#pragma unroll
                    for (int q = 0; q < 4; ++q)
                        *reinterpret_cast<float4*>(px + 8 * q) = make_float4(acc[4 * q], acc[4 * q + 1], acc[4 * q + 2], acc[4 * q + 3]);
                } else {
#pragma unroll
                    for (int r = 0; r < 16; ++r) {
                        const int o = ot * 32 + (r & 3) + 8 * (r >> 2) + 4 * hh;
                        out[((int64_t)b * CM + o) * HW + (int64_t)y * W + x0 + lo] = acc[r];
                    }
                }
            }
        };
        float ta[6], tb[6];
        gather(0, ta);
        f32x16 accA = gemm1(0, ta), accB;
        if (OT > 1) gather(1, tb);
#pragma unroll 1
        for (int ot = 0; ot < OT; ot += 2) {
            if (ot + 1 < OT) accB = gemm1(ot + 1, tb);         // GEMM 1 of tile ot+1
            if (ot + 2 < OT) gather(ot + 2, ta);
            finish(ot, accA);                                  // BN/ReLU + GEMM 2 of tile ot
            if (ot + 1 < OT) {
                if (ot + 2 < OT) accA = gemm1(ot + 2, ta);     // GEMM 1 of tile ot+2
                if (ot + 3 < OT) gather(ot + 3, tb);
                finish(ot + 1, accB);
            }
        }
        if (CLASSIFY && x0 + lo < W) {
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const int cls = (r & 3) + 8 * (r >> 2) + 4 * hh;
                if (cls < cout) out[((int64_t)b * cout + cls) * HW + (int64_t)y * W + x0 + lo] = acc2[r] + b2v[r];
            }
        }
    }
}

// ---------------------------------------------------------------------------------------
// v3: the classifier form with 16-byte LDS accesses.  Row i of o-tile `ot` is channel
//        o(ot, i) = (ot >> 2) * 128 + 4 * i + (ot & 3)
// instead of ot*32 + i: a lane then needs the SAME four consecutive floats of a staged cell for four
// consecutive o-tiles, so the A-operand gather is one ds_read_b128 per (k-slot, tap row) per group
// of four tiles (lane stride 16 B: conflict-free) — a quarter of the ds_read_b32 + address
// instructions of v2.  BatchNorm shift (the accumulators' initial value) and the pre-permuted
// classifier fragments are laid out so that they, too, load as b128.  Everything else is v2.
// ---------------------------------------------------------------------------------------
template <int OT>
__global__ __launch_bounds__(512, 2)
void head_mfma_classify_kernel(const float* __restrict__ g9, int h, int w, int H, int W, int R,
                               const float* __restrict__ shift, const float* __restrict__ w2,
                               const float* __restrict__ b2, int cout, float* __restrict__ out)
{
    static_assert(OT % 4 == 0, "groups of four o-tiles");
    constexpr int CM = OT * 32, NG = OT / 4, kT = 512;
    extern __shared__ float smem[];
    float* Gl = smem;                           // [3][4][9][CM]
    float* s_sh = smem + 108 * CM;              // [OT][2 (hh)][16 (acc reg)]
    float* s_w2 = s_sh + OT * 32;               // [OT][4 (s2 >> 2)][64 lanes][4 (s2 & 3)]
    const int tid = threadIdx.x, lane = tid & 63, wv = tid >> 6;
    const int lo = lane & 31, hh = lane >> 5;
    const int b = blockIdx.z, y0 = blockIdx.y * R, x0 = blockIdx.x * 32;
    const float sh = (float)h / (float)H, sw = (float)w / (float)W;
    const int ibase = bilinear_src(y0 > 0 ? y0 - 1 : 0, sh, h).i0;
    const int jbase = bilinear_src(x0 > 0 ? x0 - 1 : 0, sw, w).i0;
    auto chan = [](int ot, int i) { return (ot >> 2) * 128 + 4 * i + (ot & 3); };

    const float* g = g9 + (int64_t)b * h * w * 9 * CM;
    constexpr int CELL4 = 9 * CM / 4;
    for (int i = tid; i < 12 * CELL4; i += kT) {
        int cell = i / CELL4, q = i - cell * CELL4;
        int ci = ibase + (cell >> 2), cj = jbase + (cell & 3);
        if (ci > h - 1) ci = h - 1;
        if (cj > w - 1) cj = w - 1;
        reinterpret_cast<float4*>(Gl)[i] = reinterpret_cast<const float4*>(g + ((int64_t)ci * w + cj) * 9 * CM)[q];
    }
    for (int i = tid; i < OT * 32; i += kT) {
        const int r = i & 15, h2 = (i >> 4) & 1, ot = i >> 5;
        s_sh[i] = shift[chan(ot, (r & 3) + 8 * (r >> 2) + 4 * h2)];
    }
    for (int i = tid; i < OT * 16 * 64; i += kT) {
        const int e = i & 3, ln = (i >> 2) & 63, sg = (i >> 8) & 3, ot = i >> 10;
        const int s2 = sg * 4 + e, cls = ln & 31, kk = ln >> 5;
        s_w2[i] = (cls < cout) ? w2[(int64_t)cls * CM + chan(ot, (s2 & 3) + 8 * (s2 >> 2) + 4 * kk)] : 0.f;
    }
    float cx[6];
    int lbase[6];
#pragma unroll
    for (int s = 0; s < 6; ++s) {
        const int k = 2 * s + hh, kx = k >> 2, c = k & 3;
        const int xx = x0 + lo + kx - 1;
        float v = 0.f;
        if (xx >= 0 && xx < W && x0 + lo < W) {
            src_idx sx = bilinear_src(xx, sw, w);
            if (sx.i0 - jbase == c) v += sx.l0;
            if (sx.i1 - jbase == c) v += sx.l1;
        }
        cx[s] = v;
        lbase[s] = (c * 9 + kx) * CM + lo * 4;
    }
    float* s_b2 = s_w2 + OT * 16 * 64;          // [32]
    if (tid < 32) s_b2[tid] = tid < cout ? b2[tid] : 0.f;
    __syncthreads();

    const int64_t HW = (int64_t)H * W;
    for (int ry = wv; ry < R; ry += kT / 64) {
        const int y = y0 + ry;
        if (y >= H) break;
        int ub[6]; float lw[6];
#pragma unroll
        for (int ky = 0; ky < 3; ++ky) {
            const int yy = y + ky - 1;
            const bool ok = (yy >= 0 && yy < H);
            src_idx sy = bilinear_src(ok ? yy : 0, sh, h);
            const int r0 = __builtin_amdgcn_readfirstlane(sy.i0 - ibase), r1 = __builtin_amdgcn_readfirstlane(sy.i1 - ibase);
            ub[2 * ky] = (r0 * 36 + ky * 3) * CM;
            ub[2 * ky + 1] = (r1 * 36 + ky * 3) * CM;
            lw[2 * ky] = __builtin_bit_cast(float, __builtin_amdgcn_readfirstlane(__builtin_bit_cast(int, ok ? sy.l0 : 0.f)));
            lw[2 * ky + 1] = __builtin_bit_cast(float, __builtin_amdgcn_readfirstlane(__builtin_bit_cast(int, ok ? sy.l1 : 0.f)));
        }
        // A operands of the four tiles of group gi: t[q][s]
        auto gather4 = [&](int gi, float (*t)[6]) {
#pragma unroll
            for (int s = 0; s < 6; ++s) {
                const float* base = Gl + lbase[s] + gi * 128;
                float4 v = make_float4(0.f, 0.f, 0.f, 0.f);
#pragma unroll
                for (int j = 0; j < 6; ++j) {
                    const float4 a = *reinterpret_cast<const float4*>(base + ub[j]);
                    v.x = fmaf(lw[j], a.x, v.x); v.y = fmaf(lw[j], a.y, v.y);
                    v.z = fmaf(lw[j], a.z, v.z); v.w = fmaf(lw[j], a.w, v.w);
                }
                t[0][s] = v.x; t[1][s] = v.y; t[2][s] = v.z; t[3][s] = v.w;
            }
        };
        auto gemm1 = [&](int ot, const float* t) {
            f32x16 a;
            const float4* sp = reinterpret_cast<const float4*>(s_sh + (ot * 2 + hh) * 16);
#pragma unroll
            for (int q = 0; q < 4; ++q) { const float4 v = sp[q]; a[4 * q] = v.x; a[4 * q + 1] = v.y; a[4 * q + 2] = v.z; a[4 * q + 3] = v.w; }
#pragma unroll
            for (int s = 0; s < 6; ++s) a = __builtin_amdgcn_mfma_f32_32x32x2f32(t[s], cx[s], a, 0, 0, 0);
            return a;
        };
        f32x16 acc2;
#pragma unroll
        for (int r = 0; r < 16; ++r) acc2[r] = 0.f;
        auto finish = [&](int ot, f32x16& acc) {
            float wf[16];
            const float4* wp = reinterpret_cast<const float4*>(s_w2 + (int64_t)ot * 1024 + lane * 4);
#pragma unroll
            for (int q = 0; q < 4; ++q) { const float4 v = wp[q * 64]; wf[4 * q] = v.x; wf[4 * q + 1] = v.y; wf[4 * q + 2] = v.z; wf[4 * q + 3] = v.w; }
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[r] = acc[r] > 0.f ? acc[r] : 0.f;
#pragma unroll
            for (int s2 = 0; s2 < 16; ++s2) acc2 = __builtin_amdgcn_mfma_f32_32x32x2f32(wf[s2], acc[s2], acc2, 0, 0, 0);
        };
        // One A-operand buffer: the four tiles' GEMM 1 chains are issued first (their operands are dead
        // once issued), then the next group's gather runs on the VALU/LDS while the matrix pipe
        // works through the remaining GEMM 2 chains of this group.
        float ta[4][6];
        gather4(0, ta);
#pragma unroll 1
        for (int gi = 0; gi < NG; ++gi) {
            f32x16 accA = gemm1(4 * gi, ta[0]);
            f32x16 accB = gemm1(4 * gi + 1, ta[1]);
            finish(4 * gi, accA);
            accA = gemm1(4 * gi + 2, ta[2]);
            finish(4 * gi + 1, accB);
            accB = gemm1(4 * gi + 3, ta[3]);
            if (gi + 1 < NG) gather4(gi + 1, ta);
            finish(4 * gi + 2, accA);
            finish(4 * gi + 3, accB);
        }
        if (x0 + lo < W) {
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const int cls = (r & 3) + 8 * (r >> 2) + 4 * hh;
                if (cls < cout) out[((int64_t)b * cout + cls) * HW + (int64_t)y * W + x0 + lo] = acc2[r] + s_b2[cls];
            }
        }
    }
}

// ---------------------------------------------------------------------------------------
// v4: the classifier form on the f16 matrix cores with SPLIT operands.  v_mfma_f32_32x32x16_f16 issues 16x the
// multiply-adds of v_mfma_f32_32x32x2_f32 per cycle; an operand x is carried as hi = f16(x) and lo = f16(x - hi) and a
// product as hi*hi + lo*hi + hi*lo (float32 accumulation, the lo*lo term is 2^-22 of the product).
//   GEMM 1   K = 12 slots padded to one 16-deep step: lane (hh, row) holds slots {2j + hh, j < 6} in halves 0..5 of its
//            operand vector, zeros in 6..7, for A (the vertically blended cells T) and for B (the horizontal weights) alike.
//            Horizontal weights that are exact in f16 (power-of-two upsampling ratios) make the hi*lo product vanish:
//            block-uniform test, two MFMAs instead of three.
//   GEMM 2   the accumulator tile of GEMM 1 is the B operand again: registers 0..7 / 8..15 of a lane are the k = 8 hh + j
//            values of the two 16-deep steps; W2 is split once per block into LDS in that k order.
// Per 32 pixels: OT * (2..3 + 6) MFMAs of 32 cycles instead of OT * 22 of 64.
// Operand range: every |T| and every activation a wave converts is tracked (v_max3); a row whose maximum reaches 2^15 — f16
// would overflow — is recomputed by the same wave on the float32 instruction (the v3 arithmetic, W2 read from memory).
// ---------------------------------------------------------------------------------------
typedef _Float16 hd8 __attribute__((ext_vector_type(8)));
typedef unsigned hu4 __attribute__((ext_vector_type(4)));
typedef float hf2 __attribute__((ext_vector_type(2)));
typedef float hf4 __attribute__((ext_vector_type(4)));
__device__ __forceinline__ void head_split_pair(float a, float b, unsigned& hi, unsigned& lo)
{
    hi = __builtin_bit_cast(unsigned, __builtin_amdgcn_cvt_pkrtz(a, b));
    asm("v_fma_mixlo_f16 %0, %1, 1.0, -%3 op_sel_hi:[0,0,1]\n\t"
        "v_fma_mixhi_f16 %0, %2, 1.0, -%3 op_sel:[0,0,1] op_sel_hi:[0,0,1]"
        : "=&v"(lo) : "v"(a), "v"(b), "v"(hi));
}
__device__ __forceinline__ float head_max3(float m, float a, float b)
{
    float r;
    asm("v_max3_f32 %0, %1, |%2|, |%3|" : "=v"(r) : "v"(m), "v"(a), "v"(b));
    return r;
}

template <int OT>
__global__ __launch_bounds__(512, 1)
void head_split_classify_kernel(const float* __restrict__ g9, int h, int w, int H, int W, int R,
                                const float* __restrict__ shift, const float* __restrict__ w2,
                                const float* __restrict__ b2, int cout, float* __restrict__ out)
{
    static_assert(OT % 4 == 0, "groups of four o-tiles");
    constexpr int CM = OT * 32, NG = OT / 4, kT = 512;
    extern __shared__ float smem[];
    float* Gl = smem;                           // [3][4][9][CM]
    float* s_sh = smem + 108 * CM;              // [OT][2 (hh)][16 (acc reg)]
    hu4* s_w2 = reinterpret_cast<hu4*>(s_sh + OT * 32);   // [OT][2 (k step)][2 (hi, lo)][64 lanes] x 8 halves
    float* s_b2 = reinterpret_cast<float*>(s_w2 + OT * 256);   // [32]
    const int tid = threadIdx.x, lane = tid & 63, wv = tid >> 6;
    const int lo = lane & 31, hh = lane >> 5;
    const int b = blockIdx.z, y0 = blockIdx.y * R, x0 = blockIdx.x * 32;
    const float sh = (float)h / (float)H, sw = (float)w / (float)W;
    const int ibase = bilinear_src(y0 > 0 ? y0 - 1 : 0, sh, h).i0;
    const int jbase = bilinear_src(x0 > 0 ? x0 - 1 : 0, sw, w).i0;
    auto chan = [](int ot, int i) { return (ot >> 2) * 128 + 4 * i + (ot & 3); };

    const float* g = g9 + (int64_t)b * h * w * 9 * CM;
    constexpr int CELL4 = 9 * CM / 4;
    for (int i = tid; i < 12 * CELL4; i += kT) {
        int cell = i / CELL4, q = i - cell * CELL4;
        int ci = ibase + (cell >> 2), cj = jbase + (cell & 3);
        if (ci > h - 1) ci = h - 1;
        if (cj > w - 1) cj = w - 1;
        reinterpret_cast<float4*>(Gl)[i] = reinterpret_cast<const float4*>(g + ((int64_t)ci * w + cj) * 9 * CM)[q];
    }
    for (int i = tid; i < OT * 32; i += kT) {
        const int r = i & 15, h2 = (i >> 4) & 1, ot = i >> 5;
        s_sh[i] = shift[chan(ot, (r & 3) + 8 * (r >> 2) + 4 * h2)];
    }
    for (int i = tid; i < OT * 2 * 64; i += kT) {
        const int ln = i & 63, ks = (i >> 6) & 1, ot = i >> 7;
        const int cls = ln & 31, kk = ln >> 5;
        float v[8];
#pragma unroll
        for (int j = 0; j < 8; ++j) {
            const int row = (j & 3) + 8 * (j >> 2) + 16 * ks + 4 * kk;       // accumulator row of register 8 ks + j
            v[j] = (cls < cout) ? w2[(int64_t)cls * CM + chan(ot, row)] : 0.f;
        }
        hu4 Hh, Ll; unsigned a, c;
        head_split_pair(v[0], v[1], a, c); Hh[0] = a; Ll[0] = c;
        head_split_pair(v[2], v[3], a, c); Hh[1] = a; Ll[1] = c;
        head_split_pair(v[4], v[5], a, c); Hh[2] = a; Ll[2] = c;
        head_split_pair(v[6], v[7], a, c); Hh[3] = a; Ll[3] = c;
        s_w2[((ot * 2 + ks) * 2 + 0) * 64 + ln] = Hh;
        s_w2[((ot * 2 + ks) * 2 + 1) * 64 + ln] = Ll;
    }
    float cx[6];
    int lbase[6];
#pragma unroll
    for (int s = 0; s < 6; ++s) {
        const int k = 2 * s + hh, kx = k >> 2, c = k & 3;
        const int xx = x0 + lo + kx - 1;
        float v = 0.f;
        if (xx >= 0 && xx < W && x0 + lo < W) {
            src_idx sx = bilinear_src(xx, sw, w);
            if (sx.i0 - jbase == c) v += sx.l0;
            if (sx.i1 - jbase == c) v += sx.l1;
        }
        cx[s] = v;
        lbase[s] = (c * 9 + kx) * CM + lo * 4;
    }
    hu4 cxh, cxl;
    {
        unsigned a, c;
        head_split_pair(cx[0], cx[1], a, c); cxh[0] = a; cxl[0] = c;
        head_split_pair(cx[2], cx[3], a, c); cxh[1] = a; cxl[1] = c;
        head_split_pair(cx[4], cx[5], a, c); cxh[2] = a; cxl[2] = c;
        // the two spare K slots of the first half-wave carry the BatchNorm shift: B = (1, 1) against A = (shift hi, shift lo)
        cxh[3] = hh == 0 ? 0x3C003C00u : 0u; cxl[3] = 0u;
    }
    unsigned shp[OT];                           // this lane's row of every o-tile: f16 (hi, lo) of its shift; zero in half-wave 1
#pragma unroll
    for (int ot = 0; ot < OT; ++ot) {
        const float sv = shift[chan(ot, lo)];
        const _Float16 sh_hi = (_Float16)sv, sh_lo = (_Float16)(sv - (float)sh_hi);
        const unsigned pk = (unsigned)__builtin_bit_cast(unsigned short, sh_hi) | ((unsigned)__builtin_bit_cast(unsigned short, sh_lo) << 16);
        shp[ot] = hh == 0 ? pk : 0u;
    }
    // lo halves are +0 or -0 when a weight is exact in f16
    const bool cx_exact = __builtin_amdgcn_ballot_w64(((cxl[0] | cxl[1] | cxl[2]) & 0x7FFF7FFFu) != 0u) == 0ull;
    if (tid < 32) s_b2[tid] = tid < cout ? b2[tid] : 0.f;
    __syncthreads();
#define HMFMA(A, B, C) __builtin_amdgcn_mfma_f32_32x32x16_f16(__builtin_bit_cast(hd8, A), __builtin_bit_cast(hd8, B), C, 0, 0, 0)

    const int64_t HW = (int64_t)H * W;
    for (int ry = wv; ry < R; ry += kT / 64) {
        const int y = y0 + ry;
        if (y >= H) break;
        int ub[6]; float lw[6];
#pragma unroll
        for (int ky = 0; ky < 3; ++ky) {
            const int yy = y + ky - 1;
            const bool ok = (yy >= 0 && yy < H);
            src_idx sy = bilinear_src(ok ? yy : 0, sh, h);
            const int r0 = __builtin_amdgcn_readfirstlane(sy.i0 - ibase), r1 = __builtin_amdgcn_readfirstlane(sy.i1 - ibase);
            ub[2 * ky] = (r0 * 36 + ky * 3) * CM;
            ub[2 * ky + 1] = (r1 * 36 + ky * 3) * CM;
            lw[2 * ky] = __builtin_bit_cast(float, __builtin_amdgcn_readfirstlane(__builtin_bit_cast(int, ok ? sy.l0 : 0.f)));
            lw[2 * ky + 1] = __builtin_bit_cast(float, __builtin_amdgcn_readfirstlane(__builtin_bit_cast(int, ok ? sy.l1 : 0.f)));
        }
        // packed float32 FMAs on the register pairs a 16-byte LDS read delivers (v_pk_fma_f32, no operand shuffles)
        auto gather4 = [&](int gi, float (*t)[6]) {
#pragma unroll
            for (int s = 0; s < 6; ++s) {
                const float* base = Gl + lbase[s] + gi * 128;
                hf2 v01 = { 0.f, 0.f }, v23 = { 0.f, 0.f };
#pragma unroll
                for (int j = 0; j < 6; ++j) {
                    const hf4 a = *reinterpret_cast<const hf4*>(base + ub[j]);
                    const hf2 wj = { lw[j], lw[j] };
                    v01 = __builtin_elementwise_fma(wj, __builtin_shufflevector(a, a, 0, 1), v01);
                    v23 = __builtin_elementwise_fma(wj, __builtin_shufflevector(a, a, 2, 3), v23);
                }
                t[0][s] = v01[0]; t[1][s] = v01[1]; t[2][s] = v23[0]; t[3][s] = v23[1];
            }
        };
        auto shift_init = [&](int ot) {
            f32x16 a;
            const float4* sp = reinterpret_cast<const float4*>(s_sh + (ot * 2 + hh) * 16);
#pragma unroll
            for (int q = 0; q < 4; ++q) { const float4 v = sp[q]; a[4 * q] = v.x; a[4 * q + 1] = v.y; a[4 * q + 2] = v.z; a[4 * q + 3] = v.w; }
            return a;
        };
        float amax = 0.f;
        f32x16 acc2;
#pragma unroll
        for (int r = 0; r < 16; ++r) acc2[r] = 0.f;
        // A operands of the four tiles of a group, already split: th[q] / tl[q] halves 0..5 = slots, 6..7 = shift / zero.
        // A group's operands are produced in three steps (slot pairs) so that the LDS reads and packed FMAs of the NEXT
        // group can sit between the matrix instructions of this one instead of in one exposed block behind them.
        auto gather_step = [&](int gi, int sp, hu4* th, hu4* tl) {
            float t[4][2];
#pragma unroll
            for (int e = 0; e < 2; ++e) {
                const int s = 2 * sp + e;
                const float* base = Gl + lbase[s] + gi * 128;
                hf2 v01 = { 0.f, 0.f }, v23 = { 0.f, 0.f };
#pragma unroll
                for (int j = 0; j < 6; ++j) {
                    const hf4 a = *reinterpret_cast<const hf4*>(base + ub[j]);
                    const hf2 wj = { lw[j], lw[j] };
                    v01 = __builtin_elementwise_fma(wj, __builtin_shufflevector(a, a, 0, 1), v01);
                    v23 = __builtin_elementwise_fma(wj, __builtin_shufflevector(a, a, 2, 3), v23);
                }
                t[0][e] = v01[0]; t[1][e] = v01[1]; t[2][e] = v23[0]; t[3][e] = v23[1];
            }
#pragma unroll
            for (int q = 0; q < 4; ++q) {
                unsigned p, r;
                head_split_pair(t[q][0], t[q][1], p, r);
                th[q][sp] = p; tl[q][sp] = r;
                amax = head_max3(amax, t[q][0], t[q][1]);
            }
        };
        auto gemm1 = [&](unsigned shpk, hu4 th, hu4 tl) {
            f32x16 a;
#pragma unroll
            for (int r = 0; r < 16; ++r) a[r] = 0.f;
            th[3] = shpk; tl[3] = 0u;
            a = HMFMA(th, cxh, a);
            a = HMFMA(tl, cxh, a);
            if (!cx_exact) a = HMFMA(th, cxl, a);
            return a;
        };
        auto finish = [&](int ot, f32x16& acc) {
            const hu4* wp = s_w2 + ot * 256 + lane;
            const hu4 wh0 = wp[0], wl0 = wp[64], wh1 = wp[128], wl1 = wp[192];
#pragma unroll
            for (int r = 0; r < 16; ++r) {                           // one v_max_f32 (fmaxf() adds a canonicalising v_max x, x)
                float v;
                asm("v_max_f32 %0, 0, %1" : "=v"(v) : "v"(acc[r]));
                acc[r] = v;
            }
            hu4 mh0, ml0, mh1, ml1; unsigned p, q;
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                head_split_pair(acc[2 * j], acc[2 * j + 1], p, q); mh0[j] = p; ml0[j] = q;
                head_split_pair(acc[8 + 2 * j], acc[9 + 2 * j], p, q); mh1[j] = p; ml1[j] = q;
                amax = head_max3(amax, acc[2 * j], acc[2 * j + 1]);
                amax = head_max3(amax, acc[8 + 2 * j], acc[9 + 2 * j]);
            }
            acc2 = HMFMA(wh0, mh0, acc2);
            acc2 = HMFMA(wl0, mh0, acc2);
            acc2 = HMFMA(wh0, ml0, acc2);
            acc2 = HMFMA(wh1, mh1, acc2);
            acc2 = HMFMA(wl1, mh1, acc2);
            acc2 = HMFMA(wh1, ml1, acc2);
        };
        hu4 oh[2][4], ol[2][4];                                      // operands of the current / next group
#pragma unroll
        for (int sp = 0; sp < 3; ++sp) gather_step(0, sp, oh[0], ol[0]);
#pragma unroll
        for (int gi = 0; gi < NG; ++gi) {
            const int c = gi & 1, n = c ^ 1;
            const bool more = gi + 1 < NG;
            f32x16 accA = gemm1(shp[4 * gi], oh[c][0], ol[c][0]);
            f32x16 accB = gemm1(shp[4 * gi + 1], oh[c][1], ol[c][1]);
            finish(4 * gi, accA);
            if (more) gather_step(gi + 1, 0, oh[n], ol[n]);
            accA = gemm1(shp[4 * gi + 2], oh[c][2], ol[c][2]);
            finish(4 * gi + 1, accB);
            if (more) gather_step(gi + 1, 1, oh[n], ol[n]);
            accB = gemm1(shp[4 * gi + 3], oh[c][3], ol[c][3]);
            finish(4 * gi + 2, accA);
            if (more) gather_step(gi + 1, 2, oh[n], ol[n]);
            finish(4 * gi + 3, accB);
        }
        // range guard: a value at or beyond 2^15 (or an infinity) anywhere in this row -> the row again, in float32
        if (__builtin_amdgcn_ballot_w64(!(amax < 32768.f)) != 0ull) {
#pragma unroll
            for (int r = 0; r < 16; ++r) acc2[r] = 0.f;
            float ta[4][6];
            gather4(0, ta);
#pragma unroll 1
            for (int gi = 0; gi < NG; ++gi) {
#pragma unroll 1
                for (int q = 0; q < 4; ++q) {
                    const int ot = 4 * gi + q;
                    f32x16 a = shift_init(ot);
#pragma unroll
                    for (int s = 0; s < 6; ++s) a = __builtin_amdgcn_mfma_f32_32x32x2f32(q == 0 ? ta[0][s] : (q == 1 ? ta[1][s] : (q == 2 ? ta[2][s] : ta[3][s])), cx[s], a, 0, 0, 0);
#pragma unroll
                    for (int r = 0; r < 16; ++r) a[r] = a[r] > 0.f ? a[r] : 0.f;
#pragma unroll
                    for (int s2 = 0; s2 < 16; ++s2) {
                        const int o = chan(ot, (s2 & 3) + 8 * (s2 >> 2) + 4 * hh);
                        const float wf = lo < cout ? w2[(int64_t)lo * CM + o] : 0.f;
                        acc2 = __builtin_amdgcn_mfma_f32_32x32x2f32(wf, a[s2], acc2, 0, 0, 0);
                    }
                }
                if (gi + 1 < NG) gather4(gi + 1, ta);
            }
        }
        if (x0 + lo < W) {
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const int cls = (r & 3) + 8 * (r >> 2) + 4 * hh;
                if (cls < cout) out[((int64_t)b * cout + cls) * HW + (int64_t)y * W + x0 + lo] = acc2[r] + s_b2[cls];
            }
        }
    }
#undef HMFMA
}

template <int OT>
static int launch_head_split_classify(const float* g9, int64_t batch, int h, int w, int H, int W, int R, const float* shift,
                                      const float* w2, const float* b2, int cout, float* out, hipStream_t s)
{
    constexpr int CM = OT * 32;
    const size_t lds = (size_t)(108 * CM + OT * 32 + OT * 1024 + 32) * sizeof(float);
    auto kern = head_split_classify_kernel<OT>;
    hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
    if (e != hipSuccess) return (int)e;
    dim3 grid((W + 31) / 32, (H + R - 1) / R, (unsigned)batch);
    hipLaunchKernelGGL(kern, grid, dim3(512), lds, s, g9, h, w, H, W, R, shift, w2, b2, cout, out);
    e = hipGetLastError();
    return e == hipSuccess ? 0 : (int)e;
}

template <int OT>
static int launch_head_classify(const float* g9, int64_t batch, int h, int w, int H, int W, int R, const float* shift,
                                const float* w2, const float* b2, int cout, float* out, hipStream_t s)
{
    constexpr int CM = OT * 32;
    const size_t lds = (size_t)(108 * CM + OT * 32 + OT * 16 * 64 + 32) * sizeof(float);
    auto kern = head_mfma_classify_kernel<OT>;
    hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
    if (e != hipSuccess) return (int)e;
    dim3 grid((W + 31) / 32, (H + R - 1) / R, (unsigned)batch);
    hipLaunchKernelGGL(kern, grid, dim3(512), lds, s, g9, h, w, H, W, R, shift, w2, b2, cout, out);
    e = hipGetLastError();
    return e == hipSuccess ? 0 : (int)e;
}

// host mirror of bilinear_src's i0 / i1 (same float expressions; file is built -ffp-contract=off)
static void host_src(int dst, float scale, int in_size, int* i0, int* i1)
{
    float s = scale * ((float)dst + 0.5f) - 0.5f;
    if (s < 0.f) s = 0.f;
    int a = (int)s;
    if (a > in_size - 1) a = in_size - 1;
    *i0 = a; *i1 = a + (a < in_size - 1 ? 1 : 0);
}

// Largest row count R <= 32 for which every tile touches <= 3 cell rows, and whether every
// 32-pixel column tile touches <= 4 cell columns.  0 = geometry not supported by the MFMA path.
static int mfma_tile_rows(int h, int w, int H, int W)
{
    const float sh = (float)h / (float)H, sw = (float)w / (float)W;
    for (int x0 = 0; x0 < W; x0 += 32) {
        int a, b, c, d;
        host_src(x0 > 0 ? x0 - 1 : 0, sw, w, &a, &b);
        int xe = x0 + 32 < W ? x0 + 32 : W - 1;
        host_src(xe, sw, w, &c, &d);
        if (d - a > 3) return 0;
    }
    for (int R = 32; R >= 1; R >>= 1) {
        bool good = true;
        for (int y0 = 0; y0 < H && good; y0 += R) {
            int a, b, c, d;
            host_src(y0 > 0 ? y0 - 1 : 0, sh, h, &a, &b);
            int ye = y0 + R < H ? y0 + R : H - 1;
            host_src(ye, sh, h, &c, &d);
            if (d - a > 2) good = false;
        }
        if (good) return R;
    }
    return 0;
}

static int head_waves()
{
    const char* e = getenv("AWSEG_HEAD_WAVES");
    return (e && e[0] == '4') ? 4 : 8;
}

template <int OT, bool CLASSIFY, int NW>
static int launch_head_mfma_nw(const float* g9, int64_t batch, int h, int w, int H, int W, int R, const float* scale,
                            const float* shift, const float* w2, const float* b2, int cout, float* out, int out_nhwc,
                            hipStream_t s)
{
    constexpr int CM = OT * 32;
    const size_t lds = (size_t)(110 * CM + (CLASSIFY ? OT * 16 * 64 : 0)) * sizeof(float);
    auto kern = head_mfma_kernel<OT, CLASSIFY, NW>;
    hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
    if (e != hipSuccess) return (int)e;
    dim3 grid((W + 31) / 32, (H + R - 1) / R, (unsigned)batch);
    hipLaunchKernelGGL(kern, grid, dim3(NW * 64), lds, s, g9, h, w, H, W, R, scale, shift, w2, b2, cout, out, out_nhwc);
    e = hipGetLastError();
    return e == hipSuccess ? 0 : (int)e;
}

// A9: depthwise atrous 3x3 for the three ASPP rates in one pass over x (NHWC, float4 over C).
// (A strip-mined variant that hoists the weights was measured 20 % slower: the dilated taps of
// neighbouring pixels share nothing, and the longer per-lane loop only costs parallelism.)
__global__ __launch_bounds__(kThreads)
void aspp_dw3_kernel(const float* __restrict__ x, int64_t batch, int h, int w, int C,
                     const float* __restrict__ wdw, int r0, int r1, int r2, float* __restrict__ out)
{
    const int c4n = C / 4;
    const int64_t total = batch * h * w * c4n;
    const int64_t plane = batch * (int64_t)h * w * C;
    const int rates[3] = { r0, r1, r2 };
    for (int64_t i = (int64_t)blockIdx.x * kThreads + threadIdx.x; i < total; i += (int64_t)gridDim.x * kThreads) {
        int c4 = (int)(i % c4n);
        int64_t p = i / c4n;
        int xx = (int)(p % w); int64_t t = p / w;
        int yy = (int)(t % h); int64_t b = t / h;
        const float* xb = x + b * (int64_t)h * w * C;
#pragma unroll
        for (int r = 0; r < 3; ++r) {
            const int d = rates[r];
            float4 acc = make_float4(0.f, 0.f, 0.f, 0.f);
#pragma unroll
            for (int ky = 0; ky < 3; ++ky) {
                int sy = yy + (ky - 1) * d;
                if (sy < 0 || sy >= h) continue;
#pragma unroll
                for (int kx = 0; kx < 3; ++kx) {
                    int sx = xx + (kx - 1) * d;
                    if (sx < 0 || sx >= w) continue;
                    float4 v = *reinterpret_cast<const float4*>(xb + ((int64_t)sy * w + sx) * C + c4 * 4);
                    float4 k = *reinterpret_cast<const float4*>(wdw + ((int64_t)r * 9 + ky * 3 + kx) * C + c4 * 4);
                    acc.x = fmaf(v.x, k.x, acc.x); acc.y = fmaf(v.y, k.y, acc.y);
                    acc.z = fmaf(v.z, k.z, acc.z); acc.w = fmaf(v.w, k.w, acc.w);
                }
            }
            *reinterpret_cast<float4*>(out + (int64_t)r * plane + p * C + c4 * 4) = acc;
        }
    }
}

// Same arithmetic, XCD-aware work order.  The dilated taps reach 36 rows away, so sweeping pixels with all 2048
// channels at once has a 76 MB reuse distance and every tap is served by the Infinity Cache (measured: 13 GB
// through it per launch = its bandwidth).  Here a work unit is (image, slice of 64 channels) = 2 MB of input,
// which stays in ONE XCD's 4 MB L2: consecutive block ids alternate over the 8 XCDs, so unit u is given to the
// blocks whose id is congruent to u mod 8.
constexpr int kAsppSlice = 16;                       // float4 quads per slice (64 channels)
__global__ __launch_bounds__(kThreads)
void aspp_dw3_sliced_kernel(const float* __restrict__ x, int64_t batch, int h, int w, int C,
                            const float* __restrict__ wdw, int r0, int r1, int r2, float* __restrict__ out,
                            int n_units, int blocks_per_unit)
{
    const int xcd = blockIdx.x & 7, j = blockIdx.x >> 3;
    const int unit = (j / blocks_per_unit) * 8 + xcd, blk = j % blocks_per_unit;
    if (unit >= n_units) return;
    const int slices = (C / 4) / kAsppSlice;
    const int b = unit / slices, sl = unit - b * slices;
    const int item = blk * kThreads + threadIdx.x;                 // (pixel, quad within the slice)
    const int q = item % kAsppSlice, p = item / kAsppSlice;
    if (p >= h * w) return;
    const int yy = p / w, xx = p - yy * w;
    const int c = (sl * kAsppSlice + q) * 4;
    const float* xb = x + (int64_t)b * h * w * C + c;
    const int64_t plane = batch * (int64_t)h * w * C;
    const int rates[3] = { r0, r1, r2 };
#pragma unroll
    for (int r = 0; r < 3; ++r) {
        const int d = rates[r];
        float4 acc = make_float4(0.f, 0.f, 0.f, 0.f);
#pragma unroll
        for (int ky = 0; ky < 3; ++ky) {
            const int sy = yy + (ky - 1) * d;
            if (sy < 0 || sy >= h) continue;
#pragma unroll
            for (int kx = 0; kx < 3; ++kx) {
                const int sx = xx + (kx - 1) * d;
                if (sx < 0 || sx >= w) continue;
                const float4 v = *reinterpret_cast<const float4*>(xb + ((int64_t)sy * w + sx) * C);
                const float4 k = *reinterpret_cast<const float4*>(wdw + ((int64_t)r * 9 + ky * 3 + kx) * C + c);
                acc.x = fmaf(v.x, k.x, acc.x); acc.y = fmaf(v.y, k.y, acc.y);
                acc.z = fmaf(v.z, k.z, acc.z); acc.w = fmaf(v.w, k.w, acc.w);
            }
        }
        *reinterpret_cast<float4*>(out + (int64_t)r * plane + ((int64_t)b * h * w + p) * C + c) = acc;
    }
}

// 16-byte store with the non-temporal hint: the three output planes (1.5 GB per launch) stream through the same L2 the
// taps re-read their 2 MB unit from; without the hint they evict it (PMC: the 0.54 GB input came from HBM 2.6 times).
__device__ __forceinline__ void st_stream(float* p, float4 v)
{
    typedef float vf4 __attribute__((ext_vector_type(4)));
    __builtin_nontemporal_store(vf4{ v.x, v.y, v.z, v.w }, reinterpret_cast<vf4*>(p));
}
// Same units and XCD order; a lane owns (rate, row class j mod d, column, channel quad) and walks the rows j, j+d, j+2d, ...
// of its class.  An input row s feeds the outputs s-d, s, s+d of the SAME class (tap rows 2, 1, 0), so the lane loads
// each input row once — 3 loads (the three tap columns) per output instead of 9 — and carries two partial outputs:
//   out[s-d] = (h0(s-2d) + h1(s-d)) + h2(s),   h_ky(s) = sum_kx k[ky][kx] * x[s][xx + (kx-1)d]   (zero outside the map).
// The next row's loads are issued before the current row's arithmetic.
__global__ __launch_bounds__(kThreads)
void aspp_dw3_walk_kernel(const float* __restrict__ x, int64_t batch, int h, int w, int C,
                          const float* __restrict__ wdw, int r0, int r1, int r2, float* __restrict__ out,
                          int n_units, int blocks_per_unit, int slice_q)
{
    const int xcd = blockIdx.x & 7, jb = blockIdx.x >> 3;
    const int unit = (jb / blocks_per_unit) * 8 + xcd, blk = jb % blocks_per_unit;
    if (unit >= n_units) return;
    const int slices = (C / 4) / slice_q;
    const int b = unit / slices, sl = unit - b * slices;
    const int n0 = r0 < h ? r0 : h, n1 = r1 < h ? r1 : h, n2 = r2 < h ? r2 : h;
    const int item = blk * kThreads + threadIdx.x;                 // (class, column, quad within the slice)
    const int q = item % slice_q, t = item / slice_q;
    const int xx = t % w, cls = t / w;
    if (cls >= n0 + n1 + n2) return;
    const int r = cls < n0 ? 0 : (cls < n0 + n1 ? 1 : 2);
    const int j = cls - (r == 0 ? 0 : (r == 1 ? n0 : n0 + n1));
    const int d = r == 0 ? r0 : (r == 1 ? r1 : r2);
    const int c = (sl * slice_q + q) * 4;
    const float* xb = x + (int64_t)b * h * w * C + c;
    float* ob = out + (int64_t)r * (batch * (int64_t)h * w * C) + (int64_t)b * h * w * C + c;
    float4 k[9];
#pragma unroll
    for (int i = 0; i < 9; ++i) k[i] = *reinterpret_cast<const float4*>(wdw + ((int64_t)r * 9 + i) * C + c);
    const float4 zero = make_float4(0.f, 0.f, 0.f, 0.f);
    const bool okl = xx - d >= 0, okr = xx + d < w;
    auto row = [&](int s, float4* v) {
        const float* p = xb + ((int64_t)s * w + xx) * C;
        v[0] = okl ? *reinterpret_cast<const float4*>(p - (int64_t)d * C) : zero;
        v[1] = *reinterpret_cast<const float4*>(p);
        v[2] = okr ? *reinterpret_cast<const float4*>(p + (int64_t)d * C) : zero;
    };
    auto hsum = [&](int ky, const float4* v) {
        float4 a;
        a.x = v[0].x * k[ky * 3].x; a.y = v[0].y * k[ky * 3].y; a.z = v[0].z * k[ky * 3].z; a.w = v[0].w * k[ky * 3].w;
        a.x = fmaf(v[1].x, k[ky * 3 + 1].x, a.x); a.y = fmaf(v[1].y, k[ky * 3 + 1].y, a.y);
        a.z = fmaf(v[1].z, k[ky * 3 + 1].z, a.z); a.w = fmaf(v[1].w, k[ky * 3 + 1].w, a.w);
        a.x = fmaf(v[2].x, k[ky * 3 + 2].x, a.x); a.y = fmaf(v[2].y, k[ky * 3 + 2].y, a.y);
        a.z = fmaf(v[2].z, k[ky * 3 + 2].z, a.z); a.w = fmaf(v[2].w, k[ky * 3 + 2].w, a.w);
        return a;
    };
    // rows in groups of three: the nine loads of a group are in flight together (a lane's rows are megabytes apart, every
    // load is a separate trip to L2 / the Infinity Cache, and one row at a time left the kernel latency-bound)
    float4 accA = zero, accB = zero;
    for (int s0 = j; s0 < h; s0 += 3 * d) {
        float4 v[3][3];
#pragma unroll
        for (int g = 0; g < 3; ++g)
            if (s0 + g * d < h) row(s0 + g * d, v[g]);
#pragma unroll
        for (int g = 0; g < 3; ++g) {
            const int s = s0 + g * d;
            if (s >= h) break;
            const float4 h0 = hsum(0, v[g]), h1 = hsum(1, v[g]), h2 = hsum(2, v[g]);
            if (s != j)
                st_stream(ob + ((int64_t)(s - d) * w + xx) * C, make_float4(accA.x + h2.x, accA.y + h2.y, accA.z + h2.z, accA.w + h2.w));
            accA = make_float4(accB.x + h1.x, accB.y + h1.y, accB.z + h1.z, accB.w + h1.w);
            accB = h0;
            if (s + d >= h) st_stream(ob + ((int64_t)s * w + xx) * C, accA);
        }
    }
}

}  // namespace

template <int OT, bool CLASSIFY>
static int launch_head_mfma(const float* g9, int64_t batch, int h, int w, int H, int W, int R, const float* scale,
                            const float* shift, const float* w2, const float* b2, int cout, float* out, int out_nhwc,
                            hipStream_t s)
{
    if (head_waves() == 4)
        return launch_head_mfma_nw<OT, CLASSIFY, 4>(g9, batch, h, w, H, W, R, scale, shift, w2, b2, cout, out, out_nhwc, s);
    return launch_head_mfma_nw<OT, CLASSIFY, 8>(g9, batch, h, w, H, W, R, scale, shift, w2, b2, cout, out, out_nhwc, s);
}

static bool force_v1()
{
    const char* e = getenv("AWSEG_HEAD_V1");
    return e && e[0] == '1';
}

static bool force_v2()
{
    const char* e = getenv("AWSEG_HEAD_V2");
    return e && e[0] == '1';
}

static int head_dispatch(bool classify, const float* g9, int64_t batch, int cmid, int h, int w, int height, int width,
                         const float* scale, const float* shift, const float* w2, const float* b2, int cout,
                         float* out, int out_nhwc, hipStream_t s)
{
    if (!g9 || !shift || !out) return AWSEG_EINVAL;
    if (classify && (!w2 || !b2 || cout < 1 || cout > 32)) return AWSEG_EINVAL;
    if (batch < 1 || cmid < 1 || h < 1 || w < 1 || height < 1 || width < 1) return AWSEG_EINVAL;
    if (batch > 65535 || height > 65535) return AWSEG_ERANGE;
    if ((int64_t)h * w * 9 * cmid > 0x7fffffffLL) return AWSEG_ERANGE;
    const int R = ((cmid % 32) == 0 && cmid <= 256 && (((uintptr_t)g9 & 15) == 0) && !force_v1())
                      ? mfma_tile_rows(h, w, height, width) : 0;
    if (R > 0 && classify && !scale && !force_v2()) {
        if (cmid == 256) return launch_head_classify<8>(g9, batch, h, w, height, width, R, shift, w2, b2, cout, out, s);
        if (cmid == 128) return launch_head_classify<4>(g9, batch, h, w, height, width, R, shift, w2, b2, cout, out, s);
    }
    if (R > 0) {
#define AWSEG_HEAD(OTV)                                                                                              \
    case OTV:                                                                                                         \
        return classify ? launch_head_mfma<OTV, true>(g9, batch, h, w, height, width, R, scale, shift, w2, b2, cout, out, 0, s) \
                        : launch_head_mfma<OTV, false>(g9, batch, h, w, height, width, R, scale, shift, w2, b2, cout, out, out_nhwc, s);
        switch (cmid / 32) {
            AWSEG_HEAD(1) AWSEG_HEAD(2) AWSEG_HEAD(4) AWSEG_HEAD(8)
            default: break;
        }
#undef AWSEG_HEAD
    }
    if (!classify || !scale) return AWSEG_ERANGE;   // the VALU path implements the classifier form with explicit scale
    const size_t lds = (size_t)HPX * cmid * sizeof(float);
    if (lds > 60 * 1024) return AWSEG_ERANGE;
    dim3 grid((width + HPX - 1) / HPX, height, (unsigned)batch);
    hipLaunchKernelGGL(segformer_head_kernel, grid, dim3(kThreads), lds, s, g9, cmid, h, w, height, width,
                       scale, shift, w2, b2, cout, out);
    AWSEG_LAUNCH_CHECK();
    return 0;
}

AWSEG_API int awseg_segformer_head_fused(const float* g9, int64_t batch, int cmid, int h, int w, int height, int width,
                                         const float* scale, const float* shift, const float* w2, const float* b2,
                                         int cout, float* out, awseg_stream_t stream)
{
    return head_dispatch(true, g9, batch, cmid, h, w, height, width, scale, shift, w2, b2, cout, out, 0, awseg_s(stream));
}

AWSEG_API int awseg_segformer_head_fused_split(const float* g9, int64_t batch, int cmid, int h, int w, int height, int width,
                                               const float* scale, const float* shift, const float* w2, const float* b2,
                                               int cout, float* out, awseg_stream_t stream)
{
    if (!g9 || !shift || !out || !w2 || !b2 || cout < 1 || cout > 32) return AWSEG_EINVAL;
    if (batch < 1 || cmid < 1 || h < 1 || w < 1 || height < 1 || width < 1) return AWSEG_EINVAL;
    if (batch > 65535 || height > 65535) return AWSEG_ERANGE;
    if ((int64_t)h * w * 9 * cmid > 0x7fffffffLL) return AWSEG_ERANGE;
    if (scale || (cmid != 128 && cmid != 256)) return AWSEG_ERANGE;             // scale folded into g9 by the caller
    if (((uintptr_t)g9 & 15)) return AWSEG_EALIGN;
    const int R = mfma_tile_rows(h, w, height, width);
    if (R <= 0) return AWSEG_ERANGE;                                             // geometry outside the 3 x 4 cell tile
    hipStream_t s = awseg_s(stream);
    if (cmid == 256) return launch_head_split_classify<8>(g9, batch, h, w, height, width, R, shift, w2, b2, cout, out, s);
    return launch_head_split_classify<4>(g9, batch, h, w, height, width, R, shift, w2, b2, cout, out, s);
}

AWSEG_API int awseg_upconv3x3_bn_relu(const float* g9, int64_t batch, int cmid, int h, int w, int height, int width,
                                      const float* scale, const float* shift, float* out, int channels_last,
                                      awseg_stream_t stream)
{
    if (channels_last && ((uintptr_t)out & 15)) return AWSEG_EALIGN;
    return head_dispatch(false, g9, batch, cmid, h, w, height, width, scale, shift, nullptr, nullptr, 0, out,
                         channels_last ? 1 : 0, awseg_s(stream));
}

AWSEG_API int awseg_upconv3x3_linear(const float* g9, int64_t batch, int cmid, int h, int w, int height, int width,
                                     const float* bias, float* out, int channels_last, awseg_stream_t stream)
{
    if (channels_last && ((uintptr_t)out & 15)) return AWSEG_EALIGN;
    if ((cmid % 32) || cmid > 256) return AWSEG_ERANGE;
    return head_dispatch(false, g9, batch, cmid, h, w, height, width, nullptr, bias, nullptr, nullptr, 0, out,
                         (channels_last ? 1 : 0) | 2, awseg_s(stream));
}

namespace {
// Adjoint of awseg_upconv3x3_linear with respect to g9 (the backward pass of conv3x3(interpolate(f)) in training):
//   dG[b, ci, cj, tap, c] = sum over pixels (y, x) whose tap (ky, kx) lands on (y + ky - 1, x + kx - 1) inside the image
//                           and whose bilinear stencil there touches cell (ci, cj):  l_i * l_j * dz[b, y, x, c]
// Block = one 32 x 32 pixel tile of one image, thread = one channel (blockDim = Cmid): the shifted rows / columns of the
// tile touch at most 3 cell rows / columns when H/h >= 32 (checked by the launcher), so a thread keeps 3 x 3 x 9
// accumulators in registers; per-row and per-column stencil weights of the 34 shifted coordinates come from LDS tables
// (zero outside the image).  dz is read exactly once, coalesced over channels; the 81 sums per channel are added to dG
// with float atomics (<= 9 tiles meet in a cell).
__global__ __launch_bounds__(256)
void upconv3x3_adjoint_kernel(const float* __restrict__ dz, int h, int w, int H, int W, int C, float* __restrict__ dg)
{
    __shared__ float s_wy[34][3], s_wx[34][3];
    const int b = blockIdx.z, y0 = blockIdx.y * 32, x0 = blockIdx.x * 32;
    const float sh = (float)h / (float)H, sw = (float)w / (float)W;
    const int ibase = bilinear_src(y0 > 0 ? y0 - 1 : 0, sh, h).i0;
    const int jbase = bilinear_src(x0 > 0 ? x0 - 1 : 0, sw, w).i0;
    for (int i = threadIdx.x; i < 34 * 2; i += blockDim.x) {
        const bool isx = i >= 34;
        const int k = isx ? i - 34 : i;
        const int p = (isx ? x0 : y0) + k - 1;
        float wv[3] = {0.f, 0.f, 0.f};
        if (p >= 0 && p < (isx ? W : H)) {
            const src_idx sidx = bilinear_src(p, isx ? sw : sh, isx ? w : h);
            const int r0 = sidx.i0 - (isx ? jbase : ibase), r1 = sidx.i1 - (isx ? jbase : ibase);
#pragma unroll
            for (int a = 0; a < 3; ++a) wv[a] = (a == r0 ? sidx.l0 : 0.f) + (a == r1 ? sidx.l1 : 0.f);
        }
#pragma unroll
        for (int a = 0; a < 3; ++a) (isx ? s_wx : s_wy)[k][a] = wv[a];
    }
    __syncthreads();
    const int c = threadIdx.x;
    float acc[3][3][9];
#pragma unroll
    for (int a = 0; a < 3; ++a)
#pragma unroll
        for (int bb = 0; bb < 3; ++bb)
#pragma unroll
            for (int t = 0; t < 9; ++t) acc[a][bb][t] = 0.f;
    const float* src = dz + (int64_t)b * H * W * C + c;
    const int ny = (H - y0) < 32 ? (H - y0) : 32, nx = (W - x0) < 32 ? (W - x0) : 32;
    for (int r = 0; r < ny; ++r) {
        for (int q = 0; q < nx; ++q) {
            const float v = src[((int64_t)(y0 + r) * W + x0 + q) * C];
#pragma unroll
            for (int ky = 0; ky < 3; ++ky) {
                const float wy0 = s_wy[r + ky][0], wy1 = s_wy[r + ky][1], wy2 = s_wy[r + ky][2];
#pragma unroll
                for (int kx = 0; kx < 3; ++kx) {
                    const float t0 = s_wx[q + kx][0] * v, t1 = s_wx[q + kx][1] * v, t2 = s_wx[q + kx][2] * v;
                    float (&ac)[3][3][9] = acc;
                    const int t = ky * 3 + kx;
                    ac[0][0][t] = fmaf(wy0, t0, ac[0][0][t]); ac[0][1][t] = fmaf(wy0, t1, ac[0][1][t]); ac[0][2][t] = fmaf(wy0, t2, ac[0][2][t]);
                    ac[1][0][t] = fmaf(wy1, t0, ac[1][0][t]); ac[1][1][t] = fmaf(wy1, t1, ac[1][1][t]); ac[1][2][t] = fmaf(wy1, t2, ac[1][2][t]);
                    ac[2][0][t] = fmaf(wy2, t0, ac[2][0][t]); ac[2][1][t] = fmaf(wy2, t1, ac[2][1][t]); ac[2][2][t] = fmaf(wy2, t2, ac[2][2][t]);
                }
            }
        }
    }
    float* dst = dg + (int64_t)b * h * w * 9 * C + c;
#pragma unroll
    for (int a = 0; a < 3; ++a) {
        const int ci = ibase + a;
        if (ci >= h) continue;
#pragma unroll
        for (int bb = 0; bb < 3; ++bb) {
            const int cj = jbase + bb;
            if (cj >= w) continue;
#pragma unroll
            for (int t = 0; t < 9; ++t) {
                const float s = acc[a][bb][t];
                if (s != 0.f) atomicAdd(dst + (((int64_t)ci * w + cj) * 9 + t) * C, s);
            }
        }
    }
}
// The same walk with a lane per (rate, column, channel quad) that goes through ALL row classes of its rate one after the other — every
// lane visits each of the h input rows exactly once (h x 3 loads, h stores), the nine filter taps are loaded once per lane (in the
// kernel above a lane lives for one class: 1-2 loop iterations behind 9 weight loads and the index arithmetic — half of its load
// instructions were weights).  Same expressions in the same order per output: bit-identical.  AWSEG_ASPP_ROWS=0 selects the kernel above.
__global__ __launch_bounds__(kThreads)
void aspp_dw3_rows_kernel(const float* __restrict__ x, int64_t batch, int h, int w, int C,
                          const float* __restrict__ wdw, int r0, int r1, int r2, float* __restrict__ out,
                          int n_units, int blocks_per_unit, int slice_q)
{
    const int xcd = blockIdx.x & 7, jb = blockIdx.x >> 3;
    const int unit = (jb / blocks_per_unit) * 8 + xcd, blk = jb % blocks_per_unit;
    if (unit >= n_units) return;
    const int slices = (C / 4) / slice_q;
    const int b = unit / slices, sl = unit - b * slices;
    const int item = blk * kThreads + threadIdx.x;                 // (rate, column, quad within the slice)
    const int q = item % slice_q, t = item / slice_q;
    const int xx = t % w, r = t / w;
    if (r >= 3) return;
    const int d = r == 0 ? r0 : (r == 1 ? r1 : r2);
    const int ncl = d < h ? d : h;
    const int c = (sl * slice_q + q) * 4;
    const float* xb = x + (int64_t)b * h * w * C + c;
    float* ob = out + (int64_t)r * (batch * (int64_t)h * w * C) + (int64_t)b * h * w * C + c;
    float4 k[9];
#pragma unroll
    for (int i = 0; i < 9; ++i) k[i] = *reinterpret_cast<const float4*>(wdw + ((int64_t)r * 9 + i) * C + c);
    const float4 zero = make_float4(0.f, 0.f, 0.f, 0.f);
    const bool okl = xx - d >= 0, okr = xx + d < w;
    auto row = [&](int s, float4* v) {
        const float* p = xb + ((int64_t)s * w + xx) * C;
        v[0] = okl ? *reinterpret_cast<const float4*>(p - (int64_t)d * C) : zero;
        v[1] = *reinterpret_cast<const float4*>(p);
        v[2] = okr ? *reinterpret_cast<const float4*>(p + (int64_t)d * C) : zero;
    };
    auto hsum = [&](int ky, const float4* v) {
        float4 a;
        a.x = v[0].x * k[ky * 3].x; a.y = v[0].y * k[ky * 3].y; a.z = v[0].z * k[ky * 3].z; a.w = v[0].w * k[ky * 3].w;
        a.x = fmaf(v[1].x, k[ky * 3 + 1].x, a.x); a.y = fmaf(v[1].y, k[ky * 3 + 1].y, a.y);
        a.z = fmaf(v[1].z, k[ky * 3 + 1].z, a.z); a.w = fmaf(v[1].w, k[ky * 3 + 1].w, a.w);
        a.x = fmaf(v[2].x, k[ky * 3 + 2].x, a.x); a.y = fmaf(v[2].y, k[ky * 3 + 2].y, a.y);
        a.z = fmaf(v[2].z, k[ky * 3 + 2].z, a.z); a.w = fmaf(v[2].w, k[ky * 3 + 2].w, a.w);
        return a;
    };
    for (int j = 0; j < ncl; ++j) {
        float4 accA = zero, accB = zero;
        for (int s0 = j; s0 < h; s0 += 3 * d) {
            float4 v[3][3];
#pragma unroll
            for (int g = 0; g < 3; ++g)
                if (s0 + g * d < h) row(s0 + g * d, v[g]);
#pragma unroll
            for (int g = 0; g < 3; ++g) {
                const int s = s0 + g * d;
                if (s >= h) break;
                const float4 h0 = hsum(0, v[g]), h1 = hsum(1, v[g]), h2 = hsum(2, v[g]);
                if (s != j)
                    st_stream(ob + ((int64_t)(s - d) * w + xx) * C, make_float4(accA.x + h2.x, accA.y + h2.y, accA.z + h2.z, accA.w + h2.w));
                accA = make_float4(accB.x + h1.x, accB.y + h1.y, accB.z + h1.z, accB.w + h1.w);
                accB = h0;
                if (s + d >= h) st_stream(ob + ((int64_t)s * w + xx) * C, accA);
            }
        }
    }
}

// The walk with the input row staged in LDS.  The kernels above fetch every input element three times per rate from L2 (the three
// column taps are three different lanes): 4.8 GB of L2 -> CU loads + 1.6 GB of stores per launch at 8 x 64 x 128 x 2048, and the
// launch takes exactly what the ~8 TB/s L2 <-> CU path allows (0.8 ms) whatever the unit size.  Here a block of w x 8 threads is one
// (image, 32-channel slice, rate): thread = (column, channel quad); per step of the class walk every thread loads ITS pixel's quad
// of input row s (one full 128-byte line per pixel), the row goes through LDS (two buffers, one barrier per row), and the taps at
// columns x - d and x + d come from there — each input element crosses L2 -> CU once per rate: 1.6 GB + 1.6 GB.  The three rates
// of a unit are consecutive blocks of one XCD.  Same expressions in the same order per output: bit-identical to the kernels above.
// 0.88 -> 0.68 ms at 8 x 64 x 128 x 2048 (39 % of the HBM rate for the 0.54 GB in + 1.6 GB out).  What is left is the write pattern the
// NHWC layout forces on a channel-sliced kernel: 128 contiguous bytes per pixel, 8 KB apart (slice widths of 32-256 channels all
// land within 5 % of each other; the 512-byte-per-pixel writes of awseg_upconv3x3_bn_relu reach 5.7 TB/s).
constexpr int kAsppLdsQuads = 8;                                     // quads per slice: 128 bytes per pixel
// MEAN: the blocks of rate 0 also leave the per-channel mean of their slice over the image in mean_out [batch, C] — every thread of such
// a block meets each row of its column exactly once, so ASPPPooling's global average (a 0.5 GB pass of its own otherwise) is a
// running sum beside the walk and one reduction over the columns through the row buffer at the end (fixed order: deterministic).
template <int AHEAD, bool NT, bool MEAN = false>
__global__ __launch_bounds__(1024)
void aspp_dw3_lds_kernel(const float* __restrict__ x, int64_t batch, int h, int w, int C,
                         const float* __restrict__ wdw, int r0, int r1, int r2, float* __restrict__ out, int n_units,
                         float* __restrict__ mean_out = nullptr)
{
    extern __shared__ __attribute__((aligned(16))) float4 srow[];    // [2][w][8]
    const int xcd = blockIdx.x & 7, jb = blockIdx.x >> 3;
    const int unit = (jb / 3) * 8 + xcd, r = jb % 3;
    if (unit >= n_units) return;
    const int slices = (C / 4) / kAsppLdsQuads;
    const int b = unit / slices, sl = unit - b * slices;
    const int q = threadIdx.x % kAsppLdsQuads, xx = threadIdx.x / kAsppLdsQuads;     // blockDim.x == w * 8
    const int d = r == 0 ? r0 : (r == 1 ? r1 : r2);
    const int ncl = d < h ? d : h;
    const int c = (sl * kAsppLdsQuads + q) * 4;
    const float* xb = x + (int64_t)b * h * w * C + c + (int64_t)xx * C;
    float* ob = out + (int64_t)r * (batch * (int64_t)h * w * C) + (int64_t)b * h * w * C + c + (int64_t)xx * C;
    float4 k[9];
#pragma unroll
    for (int i = 0; i < 9; ++i) k[i] = *reinterpret_cast<const float4*>(wdw + ((int64_t)r * 9 + i) * C + c);
    const float4 zero = make_float4(0.f, 0.f, 0.f, 0.f);
    const bool okl = xx - d >= 0, okr = xx + d < w;
    const int il = (xx - d) * kAsppLdsQuads + q, ir = (xx + d) * kAsppLdsQuads + q, rowq = w * kAsppLdsQuads;
    auto hsum = [&](int ky, const float4& v0, const float4& v1, const float4& v2) {
        float4 a;
        a.x = v0.x * k[ky * 3].x; a.y = v0.y * k[ky * 3].y; a.z = v0.z * k[ky * 3].z; a.w = v0.w * k[ky * 3].w;
        a.x = fmaf(v1.x, k[ky * 3 + 1].x, a.x); a.y = fmaf(v1.y, k[ky * 3 + 1].y, a.y);
        a.z = fmaf(v1.z, k[ky * 3 + 1].z, a.z); a.w = fmaf(v1.w, k[ky * 3 + 1].w, a.w);
        a.x = fmaf(v2.x, k[ky * 3 + 2].x, a.x); a.y = fmaf(v2.y, k[ky * 3 + 2].y, a.y);
        a.z = fmaf(v2.z, k[ky * 3 + 2].z, a.z); a.w = fmaf(v2.w, k[ky * 3 + 2].w, a.w);
        return a;
    };
    // the rows in walk order: class 0: 0, d, 2d, ...; class 1: 1, 1 + d, ...  (every row of the map once).  A block sees one row at a
    // time, so the loads run AHEAD rows ahead of the arithmetic (one row ahead left the block waiting a memory round trip per row)
    auto put = [&](float* p, float4 v) { if (NT) st_stream(p, v); else *reinterpret_cast<float4*>(p) = v; };
    auto step = [&](int& ss, int& jj) { ss += d; if (ss >= h) { ++jj; ss = jj; } };
    float4 ring[AHEAD];
    int ls = 0, lj = 0;                                           // the row the next load fetches
#pragma unroll
    for (int i = 0; i < AHEAD; ++i) {
        ring[i] = lj < ncl ? *reinterpret_cast<const float4*>(xb + (int64_t)ls * w * C) : zero;
        step(ls, lj);
    }
    int j = 0, s = 0, buf = 0;
    float4 accA = zero, accB = zero;
    float4 msum = zero;
    while (j < ncl) {
#pragma unroll
        for (int i = 0; i < AHEAD; ++i) {                         // (unrolled so that the ring stays in registers)
            if (j >= ncl) break;
            const float4 cur = ring[i];
            if (MEAN && r == 0) { msum.x += cur.x; msum.y += cur.y; msum.z += cur.z; msum.w += cur.w; }
            ring[i] = lj < ncl ? *reinterpret_cast<const float4*>(xb + (int64_t)ls * w * C) : zero;
            step(ls, lj);
            srow[buf * rowq + threadIdx.x] = cur;
            __syncthreads();
            const float4 vl = okl ? srow[buf * rowq + il] : zero;
            const float4 vr = okr ? srow[buf * rowq + ir] : zero;
            const float4 h0 = hsum(0, vl, cur, vr), h1 = hsum(1, vl, cur, vr), h2 = hsum(2, vl, cur, vr);
            if (s != j)
                put(ob + (int64_t)(s - d) * w * C, make_float4(accA.x + h2.x, accA.y + h2.y, accA.z + h2.z, accA.w + h2.w));
            accA = make_float4(accB.x + h1.x, accB.y + h1.y, accB.z + h1.z, accB.w + h1.w);
            accB = h0;
            if (s + d >= h) { put(ob + (int64_t)s * w * C, accA); accA = zero; accB = zero; }
            step(s, j);
            buf ^= 1;
        }
    }
    if (MEAN && r == 0) {
        __syncthreads();                                          // the last row's neighbours have been read
        srow[threadIdx.x] = msum;
        __syncthreads();
        if (threadIdx.x < kAsppLdsQuads) {
            float4 t = zero;
            for (int col = 0; col < w; ++col) {
                const float4 v = srow[col * kAsppLdsQuads + threadIdx.x];
                t.x += v.x; t.y += v.y; t.z += v.z; t.w += v.w;
            }
            const float inv = 1.0f / (float)((int64_t)h * w);
            *reinterpret_cast<float4*>(mean_out + (int64_t)b * C + (sl * kAsppLdsQuads + threadIdx.x) * 4) = make_float4(t.x * inv, t.y * inv, t.z * inv, t.w * inv);
        }
    }
}

}  // namespace

AWSEG_API int awseg_upconv3x3_adjoint(const float* dz, int64_t batch, int cmid, int h, int w, int height, int width,
                                      float* dg9, awseg_stream_t stream)
{
    if (!dz || !dg9 || batch < 1 || cmid < 1 || h < 1 || w < 1 || height < 1 || width < 1) return AWSEG_EINVAL;
    if (cmid > 256 || batch > 65535) return AWSEG_ERANGE;
    // a 32-pixel tile (+1 halo) may touch at most 3 cells per axis: true when the upsampling factor is >= 32
    if ((int64_t)h * 32 > height || (int64_t)w * 32 > width) return AWSEG_ERANGE;
    hipError_t e = hipMemsetAsync(dg9, 0, (size_t)batch * h * w * 9 * cmid * sizeof(float), awseg_s(stream));
    if (e != hipSuccess) return (int)e;
    dim3 grid((width + 31) / 32, (height + 31) / 32, (unsigned)batch);
    hipLaunchKernelGGL(upconv3x3_adjoint_kernel, grid, dim3(cmid), 0, awseg_s(stream), dz, h, w, height, width, cmid, dg9);
    AWSEG_LAUNCH_CHECK();
    return 0;
}

AWSEG_API int awseg_aspp_depthwise3_mean(const float* x, int64_t batch, int h, int w, int channels, const float* wdw,
                                         int rate0, int rate1, int rate2, float* out, float* mean_out, awseg_stream_t stream)
{
    if (!x || !wdw || !out || !mean_out || batch < 1 || h < 1 || w < 1 || channels < 4 || (channels & 3)) return AWSEG_EINVAL;
    if (((uintptr_t)x & 15) || ((uintptr_t)wdw & 15) || ((uintptr_t)out & 15) || ((uintptr_t)mean_out & 15)) return AWSEG_EALIGN;
    if (rate0 < 1 || rate1 < 1 || rate2 < 1) return AWSEG_EINVAL;
    // the LDS-staged walk only: its rate-0 blocks see every pixel of their channel slice once
    if ((channels / 4) % kAsppLdsQuads || w * kAsppLdsQuads > 1024 || w * kAsppLdsQuads < 64) return AWSEG_ERANGE;
    const int n_units = (int)batch * ((channels / 4) / kAsppLdsQuads);
    const int64_t grid = (int64_t)((n_units + 7) / 8) * 3 * 8;
    if (grid >= ((int64_t)1 << 31)) return AWSEG_ERANGE;
    const size_t lds_bytes = (size_t)2 * w * kAsppLdsQuads * sizeof(float4);
    auto kern = aspp_dw3_lds_kernel<4, true, true>;
    hipLaunchKernelGGL(kern, dim3((unsigned)grid), dim3((unsigned)(w * kAsppLdsQuads)), lds_bytes, awseg_s(stream), x, batch, h, w,
                       channels, wdw, rate0, rate1, rate2, out, n_units, mean_out);
    AWSEG_LAUNCH_CHECK();
    return 0;
}

AWSEG_API int awseg_aspp_depthwise3(const float* x, int64_t batch, int h, int w, int channels, const float* wdw,
                                    int rate0, int rate1, int rate2, float* out, awseg_stream_t stream)
{
    if (!x || !wdw || !out || batch < 1 || h < 1 || w < 1 || channels < 4 || (channels & 3)) return AWSEG_EINVAL;
    if (((uintptr_t)x & 15) || ((uintptr_t)wdw & 15) || ((uintptr_t)out & 15)) return AWSEG_EALIGN;
    static const bool flat = getenv("AWSEG_ASPP_FLAT") != nullptr;
    static const bool taps = getenv("AWSEG_ASPP_TAPS") != nullptr;
    if (rate0 < 1 || rate1 < 1 || rate2 < 1) return AWSEG_EINVAL;
    const int64_t ncls = (int64_t)(rate0 < h ? rate0 : h) + (rate1 < h ? rate1 : h) + (rate2 < h ? rate2 : h);
    static const int slice_q = getenv("AWSEG_ASPP_SLICE") ? atoi(getenv("AWSEG_ASPP_SLICE")) : kAsppSlice;
    // the LDS-staged walk: a block is (image, 32-channel slice, rate) with w x 8 threads (AWSEG_ASPP_LDS=0: the kernels below)
    static const bool lds = !(getenv("AWSEG_ASPP_LDS") && atoi(getenv("AWSEG_ASPP_LDS")) == 0);
    if (lds && !flat && !taps && (channels / 4) % kAsppLdsQuads == 0 && w * kAsppLdsQuads <= 1024 && w * kAsppLdsQuads >= 64) {
        const int n_units = (int)batch * ((channels / 4) / kAsppLdsQuads);
        const int64_t grid = (int64_t)((n_units + 7) / 8) * 3 * 8;
        if (grid < ((int64_t)1 << 31)) {
            const size_t lds_bytes = (size_t)2 * w * kAsppLdsQuads * sizeof(float4);
            // (measured: 4 rows of loads ahead 0.68 ms, 1 row 0.78, 8 rows 0.84, 12 rows 1.26; plain instead of non-temporal stores +2 %)
            auto kern = aspp_dw3_lds_kernel<4, true>;
            hipLaunchKernelGGL(kern, dim3((unsigned)grid), dim3((unsigned)(w * kAsppLdsQuads)), lds_bytes, awseg_s(stream), x, batch, h, w,
                               channels, wdw, rate0, rate1, rate2, out, n_units, (float*)nullptr);
            AWSEG_LAUNCH_CHECK();
            return 0;
        }
    }
    static const bool rows = !(getenv("AWSEG_ASPP_ROWS") && atoi(getenv("AWSEG_ASPP_ROWS")) == 0);
    if (rows && !flat && !taps && slice_q >= 1 && (channels / 4) % slice_q == 0 && (int64_t)3 * w * slice_q < ((int64_t)1 << 30)) {
        const int n_units = (int)batch * ((channels / 4) / slice_q);
        const int bpu = (int)(((int64_t)3 * w * slice_q + kThreads - 1) / kThreads);
        const int64_t grid = (int64_t)((n_units + 7) / 8) * bpu * 8;
        if (grid < ((int64_t)1 << 31)) {
            hipLaunchKernelGGL(aspp_dw3_rows_kernel, dim3((unsigned)grid), dim3(kThreads), 0, awseg_s(stream), x, batch, h, w, channels,
                               wdw, rate0, rate1, rate2, out, n_units, bpu, slice_q);
            AWSEG_LAUNCH_CHECK();
            return 0;
        }
    }
    if (!flat && !taps && slice_q >= 1 && (channels / 4) % slice_q == 0 && ncls * w * slice_q < ((int64_t)1 << 30)) {
        const int n_units = (int)batch * ((channels / 4) / slice_q);
        const int bpu = (int)((ncls * w * slice_q + kThreads - 1) / kThreads);
        const int64_t grid = (int64_t)((n_units + 7) / 8) * bpu * 8;
        if (grid < ((int64_t)1 << 31)) {
            hipLaunchKernelGGL(aspp_dw3_walk_kernel, dim3((unsigned)grid), dim3(kThreads), 0, awseg_s(stream), x, batch, h, w, channels,
                               wdw, rate0, rate1, rate2, out, n_units, bpu, slice_q);
            AWSEG_LAUNCH_CHECK();
            return 0;
        }
    }
    if (!flat && (channels / 4) % kAsppSlice == 0 && (int64_t)h * w * kAsppSlice < ((int64_t)1 << 30)) {
        const int n_units = (int)batch * ((channels / 4) / kAsppSlice);
        const int bpu = (int)(((int64_t)h * w * kAsppSlice + kThreads - 1) / kThreads);
        const int64_t grid = (int64_t)((n_units + 7) / 8) * bpu * 8;
        if (grid < ((int64_t)1 << 31)) {
            hipLaunchKernelGGL(aspp_dw3_sliced_kernel, dim3((unsigned)grid), dim3(kThreads), 0, awseg_s(stream), x, batch, h, w, channels,
                               wdw, rate0, rate1, rate2, out, n_units, bpu);
            AWSEG_LAUNCH_CHECK();
            return 0;
        }
    }
    const int64_t total = batch * h * w * (channels / 4);
    hipLaunchKernelGGL(aspp_dw3_kernel, dim3(awseg_grid_1d(total, kThreads)), dim3(kThreads), 0, awseg_s(stream), x, batch,
                       h, w, channels, wdw, rate0, rate1, rate2, out);
    AWSEG_LAUNCH_CHECK();
    return 0;
}
