/*
 * awseg_oracle.c — CPU ORACLE (test infrastructure, NOT product code).
 *
 * Plain-C restatement of the reference's hot-path arithmetic, one function per
 * reference function, each citing the file:line it follows
 * (PKG = /root/reference/src/adverse_weather_semantic_segmentation_robustness_benchmark).
 * Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may load
 * this library; the product package never does.
 *
 * Pinning status:
 *   - confusion / IoU, argmax, fog, night, synthetic depth, normalise, combine,
 *     loss: PINNED against golden vectors produced by importing the reference's
 *     own Python (tests/golden/make_golden.py -> tests/golden/ npz files).
 *   - rain / snow (orc_rain, orc_snow and the rasteriser below): PARITY UNPINNED.
 *     The arithmetic lives in OpenCV (cv2.line, cv2.circle, cv2.GaussianBlur;
 *     opencv-python >= 4.8, REF/requirements.txt:5), which is absent from this
 *     container and from /root/reference.  The code restates OpenCV's published
 *     drawing.cpp / smooth algorithms (Bresenham LineIterator, ThickLine ->
 *     FillConvexPoly + end discs, midpoint Circle, separable symmetric float32
 *     Gaussian with BORDER_REFLECT_101) and is anchored on the reference's call
 *     sites PKG/data/preprocessing.py:160-166,194-200 and its shape/dtype/range
 *     tests REF/tests/test_data.py:175-195 only.
 *
 * Build: gcc -O2 -ffp-contract=off -fno-fast-math -shared -fPIC (oracle/Makefile).
 * -ffp-contract=off matters: numpy/torch evaluate these expressions with one
 * rounding per operation.
 */
#include <math.h>
#include <stdint.h>
#include <stdlib.h>
#include <string.h>

#define ORC_API __attribute__((visibility("default")))

/* ------------------------------------------------------------------ A13 --- */
/* IoUMetrics.compute_iou, PKG/evaluation/metrics.py:54-71.
 * valid = target != ignore (:58); idx = targets*C + predictions (:68) — on a
 * uint8 target tensor `targets*C` is uint8 arithmetic and wraps mod 256 before
 * the promotion to int64 (label_wrap_u8 = 1); index_add_ of ones (:70).
 * Returns the number of out-of-range indices (torch raises IndexError). */
ORC_API int64_t orc_confusion(const int64_t* pred, const void* label, int label_is_u8,
                              int64_t n, int C, int ignore_index, int label_wrap_u8,
                              int64_t* counts)
{
    int64_t oob = 0;
    for (int64_t i = 0; i < n; ++i) {
        int64_t t = label_is_u8 ? (int64_t)((const uint8_t*)label)[i]
                                : ((const int64_t*)label)[i];
        if (t == ignore_index) continue;
        int64_t base = label_wrap_u8 ? (int64_t)(uint8_t)(t * C) : t * (int64_t)C;
        int64_t idx = base + pred[i];
        if (idx < 0 || idx >= (int64_t)C * C) { ++oob; continue; }
        counts[idx] += 1;
    }
    return oob;
}

/* ------------------------------------------------------------------ A12 --- */
/* logits.argmax(dim=1), REF/scripts/evaluate.py:179.  torch semantics: first
 * maximal index; NaN is maximal (the first NaN wins). logits [B,C,HW]. */
ORC_API void orc_argmax(const float* logits, int64_t B, int C, int64_t HW, int64_t* pred)
{
    for (int64_t b = 0; b < B; ++b)
        for (int64_t p = 0; p < HW; ++p) {
            const float* x = logits + (b * C) * HW + p;
            float best = x[0];
            int bi = 0;
            for (int c = 1; c < C; ++c) {
                float v = x[(int64_t)c * HW];
                /* torch: update when (v > best) or (v is NaN and best is not) */
                if (!(v <= best) && !(best != best)) { best = v; bi = c; }
            }
            pred[b * HW + p] = bi;
        }
}

/* ------------------------------------------------------------------ A11 --- */
/* EnsembleModel.forward combine, PKG/models/model.py:443-462.
 * mode 0: w0*s1 + w1*s2 (:445-446); mode 2: (s1+s2)/2 (:457-458);
 * mode 1 (max_confidence, :449-455): use = (conf1 > conf2) as float, then
 * use*s1 + (1-use)*s2 — conf = max softmax probability of each member.
 * then / temperature (:462) when has_t.  One float32 rounding per operation. */
static float orc_maxprob(const float* x, int C, int64_t stride)
{
    float m = x[0];
    for (int c = 1; c < C; ++c) { float v = x[c * stride]; if (v > m) m = v; }
    float s = 0.f;
    for (int c = 0; c < C; ++c) s += expf(x[c * stride] - m);
    return 1.0f / s; /* exp(m-m)/sum */
}

ORC_API void orc_combine(const float* s1, const float* s2, int64_t B, int C, int64_t HW,
                         int mode, float w0, float w1, int has_t, float T, float* out)
{
    for (int64_t b = 0; b < B; ++b)
        for (int64_t p = 0; p < HW; ++p) {
            const float* a = s1 + (b * C) * HW + p;
            const float* d = s2 + (b * C) * HW + p;
            float use = 0.f;
            if (mode == 1) use = orc_maxprob(a, C, HW) > orc_maxprob(d, C, HW) ? 1.f : 0.f;
            for (int c = 0; c < C; ++c) {
                float x = a[(int64_t)c * HW], y = d[(int64_t)c * HW], r;
                if (mode == 0) { float u = w0 * x; float v = w1 * y; r = u + v; }
                else if (mode == 1) { float u = use * x; float v = (1.f - use) * y; r = u + v; }
                else { float u = x + y; r = u / 2.f; }
                if (has_t) r = r / T;
                out[(b * C + c) * HW + p] = r;
            }
        }
}

/* ------------------------------------------------------------------- A7 --- */
/* Normalize + ToTensorV2, PKG/data/loader.py:195-198, 275-278, restated as
 * (x/255 - mean)/std in float32 (albumentations absent). HWC->CHW */
ORC_API void orc_normalize(const uint8_t* img, int H, int W, const float* mean,
                           const float* std, float* out)
{
    int64_t hw = (int64_t)H * W;
    for (int64_t p = 0; p < hw; ++p)
        for (int c = 0; c < 3; ++c) {
            float v = (float)img[p * 3 + c] / 255.0f;
            float d = v - mean[c];
            out[c * hw + p] = d / std[c];
        }
}

/* ------------------------------------------------------------------- A2 --- */
/* _generate_synthetic_depth, PKG/data/preprocessing.py:235-246.
 * depth_base = (y/height)*100 (:236) + noise (:240); gaussian_filter(sigma=2)
 * (:243) = scipy.ndimage correlate1d along axis 0 then axis 1 with the 17-tap
 * normalised kernel, mode 'reflect' (d c b a | a b c d | d c b a), float64.
 * scipy's symmetric-kernel loop (ni_filters.c NI_Correlate1D):
 *     o = in[0]*w[0]; for j = -r..-1: o += (in[j] + in[-j]) * w[j]
 * max(depth, 1.0) (:246).  taps[17] are passed in (computed by numpy exactly as
 * scipy's _gaussian_kernel1d does) so the weights are bit-identical. */
static inline int orc_reflect(int i, int n)
{
    /* scipy 'reflect' == numpy 'symmetric': -1 -> 0, -2 -> 1, n -> n-1 */
    if (n == 1) return 0;
    int period = 2 * n;
    i %= period; if (i < 0) i += period;
    return i < n ? i : period - 1 - i;
}

ORC_API void orc_gauss17(const double* in, int H, int W, const double* taps, double* out)
{
    const int R = 8;
    double* tmp = (double*)malloc(sizeof(double) * (size_t)H * W);
    /* axis 0 */
    for (int y = 0; y < H; ++y)
        for (int x = 0; x < W; ++x) {
            double o = in[(int64_t)y * W + x] * taps[R];
            for (int j = -R; j < 0; ++j) {
                double a = in[(int64_t)orc_reflect(y + j, H) * W + x];
                double b = in[(int64_t)orc_reflect(y - j, H) * W + x];
                o += (a + b) * taps[R + j];
            }
            tmp[(int64_t)y * W + x] = o;
        }
    /* axis 1 */
    for (int y = 0; y < H; ++y)
        for (int x = 0; x < W; ++x) {
            const double* row = tmp + (int64_t)y * W;
            double o = row[x] * taps[R];
            for (int j = -R; j < 0; ++j)
                o += (row[orc_reflect(x + j, W)] + row[orc_reflect(x - j, W)]) * taps[R + j];
            out[(int64_t)y * W + x] = o;
        }
    free(tmp);
}

ORC_API void orc_synthetic_depth(const double* noise, int H, int W, const double* taps,
                                 double* depth)
{
    double* d = (double*)malloc(sizeof(double) * (size_t)H * W);
    for (int y = 0; y < H; ++y) {
        double base = ((double)y / (double)H) * 100.0;
        for (int x = 0; x < W; ++x) d[(int64_t)y * W + x] = base + noise[(int64_t)y * W + x];
    }
    orc_gauss17(d, H, W, taps, depth);
    for (int64_t i = 0; i < (int64_t)H * W; ++i) depth[i] = depth[i] > 1.0 ? depth[i] : 1.0;
    free(d);
}

static inline uint8_t orc_quant(double v)
{
    /* (np.clip(v,0,1)*255).astype(np.uint8): truncation toward zero */
    double c = v < 0.0 ? 0.0 : (v > 1.0 ? 1.0 : v);
    return (uint8_t)(c * 255.0);
}

/* ------------------------------------------------------------------- A3 --- */
/* _apply_fog, PKG/data/preprocessing.py:113-123.  image = u8.astype(f32)/255
 * (:81); transmission = exp(-beta*depth) float64 (:117); atmospheric_light =
 * A*ones_like(image) is a float32 array, so A is rounded to float32 (:118);
 * fogged = image*t + A32*(1-t) in float64 (:120-121); quantise (:123). */
ORC_API void orc_fog(const uint8_t* img, const double* depth, int H, int W,
                     double beta, double A, uint8_t* out)
{
    double A32 = (double)(float)A;
    int64_t hw = (int64_t)H * W;
    for (int64_t p = 0; p < hw; ++p) {
        double t = exp(-beta * depth[p]);
        double omt = 1.0 - t;
        for (int c = 0; c < 3; ++c) {
            float v = (float)img[p * 3 + c] / 255.0f;
            double a = (double)v * t;
            double b = A32 * omt;
            out[p * 3 + c] = orc_quant(a + b);
        }
    }
}

/* ------------------------------------------------------------------- A6 --- */
/* _apply_night, PKG/data/preprocessing.py:209-225.  night = image*bf with bf a
 * Python float -> float32 multiply (:213); per-channel *= 0.8/0.85/1.2 in
 * float32 (:217-219); + noise*intensity*0.5 in float64 (:223); quantise. */
ORC_API void orc_night(const uint8_t* img, const double* noise, int H, int W,
                       double brightness, double intensity, const float* gains,
                       uint8_t* out)
{
    float bf = (float)brightness;
    int64_t hw = (int64_t)H * W;
    for (int64_t p = 0; p < hw; ++p)
        for (int c = 0; c < 3; ++c) {
            float v = (float)img[p * 3 + c] / 255.0f;
            v = v * bf;
            v = v * gains[c];
            double n = noise[p * 3 + c] * intensity;
            n = n * 0.5;
            out[p * 3 + c] = orc_quant((double)v + n);
        }
}

/* ------------------------------------------------------- rasteriser (A4/A5) */
/* PARITY UNPINNED — restates OpenCV drawing.cpp from its published algorithm. */
typedef struct { uint8_t* m; int W, H; } orc_mask;

static inline void orc_hline(orc_mask* k, int y, int x0, int x1)
{
    if (y < 0 || y >= k->H) return;
    if (x0 < 0) x0 = 0;
    if (x1 >= k->W) x1 = k->W - 1;
    for (int x = x0; x <= x1; ++x) k->m[(int64_t)y * k->W + x] = 1;
}
static inline void orc_point(orc_mask* k, int x, int y)
{
    if (x >= 0 && x < k->W && y >= 0 && y < k->H) k->m[(int64_t)y * k->W + x] = 1;
}

/* cv::Line / LineIterator(connectivity 8, leftToRight=true): Bresenham from the
 * left endpoint; x-major when dx >= dy.  Minor coordinate after i major steps is
 * floor((2*dmin*i + dmaj - 1) / (2*dmaj))  (closed form of err = dmaj - 2*dmin;
 * err < 0 -> step).  Both endpoints are inside the image on the reference's call
 * path (PKG/data/preprocessing.py:144-156), so no clipping is involved. */
static void orc_line_thin(orc_mask* k, int x0, int y0, int x1, int y1)
{
    if (x1 < x0) { int t = x0; x0 = x1; x1 = t; t = y0; y0 = y1; y1 = t; }
    int dx = x1 - x0, dy = y1 - y0, sy = dy < 0 ? -1 : 1;
    if (dy < 0) dy = -dy;
    if (dx >= dy) {
        if (dx == 0) { orc_point(k, x0, y0); return; }
        for (int i = 0; i <= dx; ++i)
            orc_point(k, x0 + i, y0 + sy * (int)((2LL * dy * i + dx - 1) / (2LL * dx)));
    } else {
        for (int i = 0; i <= dy; ++i)
            orc_point(k, x0 + (int)((2LL * dx * i + dy - 1) / (2LL * dy)), y0 + sy * i);
    }
}

/* cv::Circle(fill): midpoint circle, horizontal spans. */
static void orc_disc(orc_mask* k, int cx, int cy, int r)
{
    int err = 0, dx = r, dy = 0, plus = 1, minus = (r << 1) - 1;
    while (dx >= dy) {
        orc_hline(k, cy - dy, cx - dx, cx + dx);
        orc_hline(k, cy + dy, cx - dx, cx + dx);
        orc_hline(k, cy - dx, cx - dy, cx + dy);
        orc_hline(k, cy + dx, cx - dy, cx + dy);
        dy++;
        err += plus; plus += 2;
        int mask = (err <= 0) - 1;
        err -= minus & mask;
        dx += mask;
        minus -= mask & 2;
    }
}

#define ORC_XY_SHIFT 16
#define ORC_XY_ONE (1 << ORC_XY_SHIFT)

/* cv::Line2: 16.16 fixed-point DDA used for polygon outlines.  Segments here lie
 * within a few pixels of the image, the clip against the scaled image rectangle
 * is applied per plotted point (equivalent for points, the DDA itself is not
 * re-anchored: documented deviation when an outline crosses the border). */
static void orc_line2(orc_mask* k, int64_t x1, int64_t y1, int64_t x2, int64_t y2)
{
    int64_t dx = x2 - x1, dy = y2 - y1;
    int64_t ax = dx < 0 ? -dx : dx, ay = dy < 0 ? -dy : dy;
    orc_point(k, (int)((x2 + (ORC_XY_ONE >> 1)) >> ORC_XY_SHIFT),
                 (int)((y2 + (ORC_XY_ONE >> 1)) >> ORC_XY_SHIFT));
    if (ax > ay) {
        if (dx < 0) { int64_t t = x1; x1 = x2; x2 = t; t = y1; y1 = y2; y2 = t; dy = -dy; }
        int64_t y_step = (dy * ORC_XY_ONE) / (ax | 1);
        int ecount = (int)((x2 - x1) >> ORC_XY_SHIFT);
        x1 += ORC_XY_ONE >> 1; y1 += ORC_XY_ONE >> 1;
        int64_t x = x1 >> ORC_XY_SHIFT;
        while (ecount >= 0) { orc_point(k, (int)x, (int)(y1 >> ORC_XY_SHIFT)); x++; y1 += y_step; ecount--; }
    } else {
        if (dy < 0) { int64_t t = x1; x1 = x2; x2 = t; t = y1; y1 = y2; y2 = t; dx = -dx; }
        int64_t x_step = (dx * ORC_XY_ONE) / (ay | 1);
        int ecount = (int)((y2 - y1) >> ORC_XY_SHIFT);
        x1 += ORC_XY_ONE >> 1; y1 += ORC_XY_ONE >> 1;
        int64_t y = y1 >> ORC_XY_SHIFT;
        while (ecount >= 0) { orc_point(k, (int)(x1 >> ORC_XY_SHIFT), (int)y); y++; x1 += x_step; ecount--; }
    }
}

/* cv::FillConvexPoly (shift = 16, line_type 8) for the 4-point thick-line body. */
static void orc_fill_convex4(orc_mask* k, const int64_t vx[4], const int64_t vy[4])
{
    const int npts = 4, shift = ORC_XY_SHIFT;
    const int64_t delta = (1 << shift) >> 1;
    struct { int idx, di; int64_t x, dx; int ye; } edge[2];
    int imin = 0, edges = npts;
    int64_t xmin = vx[0], xmax = vx[0], ymin = vy[0], ymax = vy[0];
    int64_t px = vx[npts - 1], py = vy[npts - 1];
    for (int i = 0; i < npts; ++i) {
        if (vy[i] < ymin) { ymin = vy[i]; imin = i; }
        if (vy[i] > ymax) ymax = vy[i];
        if (vx[i] > xmax) xmax = vx[i];
        if (vx[i] < xmin) xmin = vx[i];
        orc_line2(k, px, py, vx[i], vy[i]);
        px = vx[i]; py = vy[i];
    }
    xmin = (xmin + delta) >> shift; xmax = (xmax + delta) >> shift;
    ymin = (ymin + delta) >> shift; ymax = (ymax + delta) >> shift;
    if ((int)xmax < 0 || (int)ymax < 0 || (int)xmin >= k->W || (int)ymin >= k->H) return;
    if (ymax > k->H - 1) ymax = k->H - 1;
    int y = (int)ymin;
    edge[0].idx = edge[1].idx = imin;
    edge[0].ye = edge[1].ye = y;
    edge[0].di = 1; edge[1].di = npts - 1;
    edge[0].x = edge[1].x = -ORC_XY_ONE;
    edge[0].dx = edge[1].dx = 0;
    do {
        for (int i = 0; i < 2; ++i) {
            if (y >= edge[i].ye) {
                int idx0 = edge[i].idx, di = edge[i].di;
                int idx = idx0 + di; if (idx >= npts) idx -= npts;
                for (; edges-- > 0;) {
                    int ty = (int)((vy[idx] + delta) >> shift);
                    if (ty > y) {
                        int64_t xs = vx[idx0], xe = vx[idx];
                        edge[i].ye = ty;
                        edge[i].dx = ((xe - xs) * 2 + (ty - y)) / (2 * (ty - y));
                        edge[i].x = xs;
                        edge[i].idx = idx;
                        break;
                    }
                    idx0 = idx; idx += di; if (idx >= npts) idx -= npts;
                }
            }
        }
        if (edges < 0) break;
        if (y >= 0) {
            int left = 0, right = 1;
            if (edge[0].x > edge[1].x) { left = 1; right = 0; }
            int xx1 = (int)((edge[left].x + (ORC_XY_ONE >> 1)) >> ORC_XY_SHIFT);
            int xx2 = (int)((edge[right].x + (ORC_XY_ONE >> 1)) >> ORC_XY_SHIFT);
            if (xx2 >= 0 && xx1 < k->W) orc_hline(k, y, xx1, xx2);
        }
        edge[0].x += edge[0].dx;
        edge[1].x += edge[1].dx;
    } while (++y <= (int)ymax);
}

/* cv::ThickLine for thickness > 1: rotated-rectangle body + a disc of radius
 * (thickness*32768 + 32768) >> 16 at each endpoint. */
static void orc_line_thick(orc_mask* k, int x0, int y0, int x1, int y1, int thickness)
{
    int64_t p0x = (int64_t)x0 << ORC_XY_SHIFT, p0y = (int64_t)y0 << ORC_XY_SHIFT;
    int64_t p1x = (int64_t)x1 << ORC_XY_SHIFT, p1y = (int64_t)y1 << ORC_XY_SHIFT;
    const double INV = 1.0 / ORC_XY_ONE;
    double dx = (double)(p0x - p1x) * INV, dy = (double)(p1y - p0y) * INV;
    double r = dx * dx + dy * dy;
    int odd = thickness & 1;
    int64_t th = (int64_t)thickness << (ORC_XY_SHIFT - 1);
    if (fabs(r) > 2.220446049250313e-16) {
        r = ((double)th + odd * ORC_XY_ONE * 0.5) / sqrt(r);
        int64_t dpx = (int64_t)nearbyint(dy * r), dpy = (int64_t)nearbyint(dx * r);
        int64_t vx[4] = { p0x + dpx, p0x - dpx, p1x - dpx, p1x + dpx };
        int64_t vy[4] = { p0y + dpy, p0y - dpy, p1y - dpy, p1y + dpy };
        orc_fill_convex4(k, vx, vy);
    }
    int rad = (int)((th + (ORC_XY_ONE >> 1)) >> ORC_XY_SHIFT);
    orc_disc(k, x0, y0, rad);
    orc_disc(k, x1, y1, rad);
}

ORC_API void orc_raster_drops(const int32_t* drops, int n, int H, int W, uint8_t* mask)
{
    orc_mask k = { mask, W, H };
    for (int i = 0; i < n; ++i) {
        const int32_t* d = drops + 5 * i;
        if (d[4] <= 1) orc_line_thin(&k, d[0], d[1], d[2], d[3]);
        else orc_line_thick(&k, d[0], d[1], d[2], d[3], d[4]);
    }
}
ORC_API void orc_raster_flakes(const int32_t* flakes, int n, int H, int W, uint8_t* mask)
{
    orc_mask k = { mask, W, H };
    for (int i = 0; i < n; ++i) orc_disc(&k, flakes[3 * i], flakes[3 * i + 1], flakes[3 * i + 2]);
}

/* cv::getGaussianKernel(ksize, sigma>0, CV_32F): exp(-x^2/(2 sigma^2)) computed in
 * float64, normalised by the float64 sum, stored as float32. */
ORC_API void orc_gauss_kernel_f32(int ksize, double sigma, float* k)
{
    double t[16], sum = 0.0, s2 = -0.5 / (sigma * sigma);
    for (int i = 0; i < ksize; ++i) { double x = i - (ksize - 1) * 0.5; t[i] = exp(s2 * x * x); sum += t[i]; }
    sum = 1.0 / sum;
    for (int i = 0; i < ksize; ++i) k[i] = (float)(t[i] * sum);
}

static inline int orc_reflect101(int i, int n)
{
    /* BORDER_REFLECT_101: gfedcb|abcdefgh|gfedcba */
    if (n == 1) return 0;
    while (i < 0 || i >= n) { if (i < 0) i = -i; else i = 2 * n - 2 - i; }
    return i;
}

/* cv::GaussianBlur on CV_32FC3: separable, rows then columns, float32, symmetric
 * form  k[c]*s[0] + sum_j k[c+j]*(s[-j] + s[j]). */
static void orc_blur_f32(float* img, int H, int W, int ksize, double sigma)
{
    float k[16]; orc_gauss_kernel_f32(ksize, sigma, k);
    int r = ksize / 2;
    float* tmp = (float*)malloc(sizeof(float) * (size_t)H * W * 3);
    for (int y = 0; y < H; ++y)
        for (int x = 0; x < W; ++x)
            for (int c = 0; c < 3; ++c) {
                const float* row = img + (int64_t)y * W * 3;
                float s = k[r] * row[x * 3 + c];
                for (int j = 1; j <= r; ++j) {
                    float a = row[orc_reflect101(x - j, W) * 3 + c];
                    float b = row[orc_reflect101(x + j, W) * 3 + c];
                    float ab = a + b;
                    float m = k[r + j] * ab;
                    s = s + m;
                }
                tmp[((int64_t)y * W + x) * 3 + c] = s;
            }
    for (int y = 0; y < H; ++y)
        for (int x = 0; x < W; ++x)
            for (int c = 0; c < 3; ++c) {
                float s = k[r] * tmp[((int64_t)y * W + x) * 3 + c];
                for (int j = 1; j <= r; ++j) {
                    float a = tmp[((int64_t)orc_reflect101(y - j, H) * W + x) * 3 + c];
                    float b = tmp[((int64_t)orc_reflect101(y + j, H) * W + x) * 3 + c];
                    float ab = a + b;
                    float m = k[r + j] * ab;
                    s = s + m;
                }
                img[((int64_t)y * W + x) * 3 + c] = s;
            }
    free(tmp);
}

static inline uint8_t orc_quant_f32(float v)
{
    /* (np.clip(f32,0,1)*255).astype(uint8): float32 multiply then truncation */
    float c = v < 0.f ? 0.f : (v > 1.f ? 1.f : v);
    return (uint8_t)(c * 255.0f);
}

/* ------------------------------------------------------------------- A4 --- */
/* _apply_rain, PKG/data/preprocessing.py:131-168.  haze = intensity*0.3 is a
 * Python float; rain*(1-haze) + haze*0.7 are float32-array x Python-scalar ops:
 * float32 multiply by f32(1-haze), float32 add of f32(haze*0.7) (:134-135).
 * Streak colour (0.8,0.9,1.0) (:159); blur (3,3) sigma 0.5 (:166); quantise. */
ORC_API void orc_rain(const uint8_t* img, int H, int W, double intensity,
                      const int32_t* drops, int n_drops, uint8_t* out)
{
    int64_t hw = (int64_t)H * W;
    double haze = intensity * 0.3;
    float m = (float)(1.0 - haze), a = (float)(haze * 0.7);
    float* f = (float*)malloc(sizeof(float) * (size_t)hw * 3);
    uint8_t* mask = (uint8_t*)calloc((size_t)hw, 1);
    for (int64_t i = 0; i < hw * 3; ++i) { float v = (float)img[i] / 255.0f; v = v * m; f[i] = v + a; }
    orc_raster_drops(drops, n_drops, H, W, mask);
    const float col[3] = { 0.8f, 0.9f, 1.0f };
    for (int64_t p = 0; p < hw; ++p)
        if (mask[p]) for (int c = 0; c < 3; ++c) f[p * 3 + c] = col[c];
    orc_blur_f32(f, H, W, 3, 0.5);
    for (int64_t i = 0; i < hw * 3; ++i) out[i] = orc_quant_f32(f[i]);
    free(f); free(mask);
}

/* ------------------------------------------------------------------- A5 --- */
/* _apply_snow, PKG/data/preprocessing.py:176-202.  clip(img + f32(0.2*I),0,1)
 * (:179-180); filled discs of 1.0 (:194); blur ksize 3|7 sigma 1.0 (:197-200). */
ORC_API void orc_snow(const uint8_t* img, int H, int W, double intensity,
                      const int32_t* flakes, int n_flakes, int ksize, uint8_t* out)
{
    int64_t hw = (int64_t)H * W;
    float boost = (float)(intensity * 0.2);
    float* f = (float*)malloc(sizeof(float) * (size_t)hw * 3);
    uint8_t* mask = (uint8_t*)calloc((size_t)hw, 1);
    for (int64_t i = 0; i < hw * 3; ++i) {
        float v = (float)img[i] / 255.0f; v = v + boost;
        f[i] = v < 0.f ? 0.f : (v > 1.f ? 1.f : v);
    }
    orc_raster_flakes(flakes, n_flakes, H, W, mask);
    for (int64_t p = 0; p < hw; ++p)
        if (mask[p]) for (int c = 0; c < 3; ++c) f[p * 3 + c] = 1.0f;
    orc_blur_f32(f, H, W, ksize, 1.0);
    for (int64_t i = 0; i < hw * 3; ++i) out[i] = orc_quant_f32(f[i]);
    free(f); free(mask);
}

/* ------------------------------------------------------------------ A15 --- */
/* FogDensityAwareLoss.forward, PKG/models/model.py:577-587, 610 and
 * _focal_loss :638-642.  ce = -log_softmax(x)[label] in float32 (torch:
 * x - max - log(sum(exp(x - max)))); focal = (1-exp(-ce))^2 * ce;
 * * (1 + s*density) (:586-587); mean (:610) accumulated in float64 here (the
 * tolerance on the loss is 1e-4 abs).  Returns oob count (label outside [0,C)).
 * pixel_loss may be NULL. */
ORC_API int64_t orc_fog_ce(const float* logits, const void* label, int label_is_u8,
                           const float* density, int64_t B, int C, int64_t HW,
                           int focal, float s, float* pixel_loss, double* mean_out)
{
    double acc = 0.0; int64_t oob = 0;
    for (int64_t b = 0; b < B; ++b)
        for (int64_t p = 0; p < HW; ++p) {
            const float* x = logits + (b * C) * HW + p;
            int64_t t = label_is_u8 ? (int64_t)((const uint8_t*)label)[b * HW + p]
                                    : ((const int64_t*)label)[b * HW + p];
            if (t < 0 || t >= C) { ++oob; if (pixel_loss) pixel_loss[b * HW + p] = 0.f; continue; }
            float m = x[0];
            for (int c = 1; c < C; ++c) { float v = x[(int64_t)c * HW]; if (v > m) m = v; }
            float sum = 0.f;
            for (int c = 0; c < C; ++c) sum += expf(x[(int64_t)c * HW] - m);
            float lse = logf(sum);
            float ce = -((x[t * HW] - m) - lse);
            if (focal) { float pt = expf(-ce); float q = 1.f - pt; ce = (q * q) * ce; }
            if (density) { float w = 1.0f + s * density[b * HW + p]; ce = ce * w; }
            if (pixel_loss) pixel_loss[b * HW + p] = ce;
            acc += (double)ce;
        }
    *mean_out = acc / (double)(B * HW);
    return oob;
}

/* d(mean loss)/d(logits): g/N * w * dce/dx, dce/dx_c = softmax_c - [c==t];
 * focal: d/dce[(1-e^-ce)^2 ce] = (1-pt)^2 + 2 ce pt (1-pt). */
ORC_API void orc_fog_ce_grad(const float* logits, const void* label, int label_is_u8,
                             const float* density, int64_t B, int C, int64_t HW,
                             int focal, float s, float g, float* grad)
{
    double invn = 1.0 / (double)(B * HW);
    for (int64_t b = 0; b < B; ++b)
        for (int64_t p = 0; p < HW; ++p) {
            const float* x = logits + (b * C) * HW + p;
            int64_t t = label_is_u8 ? (int64_t)((const uint8_t*)label)[b * HW + p]
                                    : ((const int64_t*)label)[b * HW + p];
            float m = x[0];
            for (int c = 1; c < C; ++c) { float v = x[(int64_t)c * HW]; if (v > m) m = v; }
            double sum = 0.0;
            for (int c = 0; c < C; ++c) sum += exp((double)x[(int64_t)c * HW] - m);
            double ce = -(((double)x[t * HW] - m) - log(sum));
            double k = 1.0;
            if (focal) { double pt = exp(-ce), q = 1.0 - pt; k = q * q + 2.0 * ce * pt * q; }
            double w = density ? 1.0 + (double)s * density[b * HW + p] : 1.0;
            for (int c = 0; c < C; ++c) {
                double sm = exp((double)x[(int64_t)c * HW] - m) / sum;
                grad[(b * C + c) * HW + p] = (float)((double)g * invn * w * k * (sm - (c == t ? 1.0 : 0.0)));
            }
        }
}

/* _estimate_fog_density_from_depth, PKG/models/model.py:658-677 (float32).
 * depth [B,H,W]; statistics over the whole batch. */
ORC_API void orc_fog_density_from_depth(const float* depth, int64_t B, int H, int W, float* out)
{
    int64_t n = B * (int64_t)H * W;
    float mn = depth[0], mx = depth[0];
    for (int64_t i = 1; i < n; ++i) { if (depth[i] < mn) mn = depth[i]; if (depth[i] > mx) mx = depth[i]; }
    float den = (mx - mn) + 1e-8f;
    float* gm = (float*)malloc(sizeof(float) * (size_t)n);
    double acc = 0.0;
    for (int64_t b = 0; b < B; ++b)
        for (int y = 0; y < H; ++y)
            for (int x = 0; x < W; ++x) {
                const float* d = depth + b * (int64_t)H * W;
                /* forward difference, last column/row replicate the previous gradient */
                int xx = x < W - 1 ? x : (W > 1 ? W - 2 : 0);
                int yy = y < H - 1 ? y : (H > 1 ? H - 2 : 0);
                float gx = W > 1 ? fabsf(d[(int64_t)y * W + xx + 1] - d[(int64_t)y * W + xx]) : 0.f;
                float gy = H > 1 ? fabsf(d[(int64_t)(yy + 1) * W + x] - d[(int64_t)yy * W + x]) : 0.f;
                float a = gx * gx, c = gy * gy;
                float s2 = a + c; s2 = s2 + 1e-8f;
                float g = sqrtf(s2);
                gm[b * (int64_t)H * W + (int64_t)y * W + x] = g;
                acc += (double)g;
            }
    float mean = (float)(acc / (double)n);
    for (int64_t i = 0; i < n; ++i) {
        float dn = (depth[i] - mn) / den;
        float f = dn * 0.7f;
        float e = gm[i] > mean ? 0.3f : 0.0f;
        f = f - e;
        out[i] = f < 0.f ? 0.f : (f > 1.f ? 1.f : f);
    }
    free(gm);
}

/* ------------------------------------------------------------------- A8 --- */
/* SegFormerModel.forward head, PKG/models/model.py:209-214, AS WRITTEN:
 * F.interpolate(size=(H,W), bilinear, align_corners=False) -> Conv3x3(pad 1) ->
 * eval BatchNorm folded to scale/shift -> ReLU -> Conv1x1.  float64 accumulation
 * (reference for a 1e-4 tolerance, small shapes only: O(H W Cin Cmid 9)).
 * feat [Cin,h,w]; w1 [Cmid,Cin,3,3]; scale/shift [Cmid] (conv bias folded in);
 * w2 [Cout,Cmid]; b2 [Cout]; out [Cout,H,W]. */
ORC_API void orc_segformer_head(const float* feat, int Cin, int h, int w, int H, int W,
                                const float* w1, const float* scale, const float* shift,
                                int Cmid, const float* w2, const float* b2, int Cout, float* out)
{
    int64_t HW = (int64_t)H * W;
    float* up = (float*)malloc(sizeof(float) * (size_t)Cin * HW);
    float sh = (float)h / (float)H, sw = (float)w / (float)W;
    for (int y = 0; y < H; ++y) {
        float fy = ((float)y + 0.5f) * sh - 0.5f; if (fy < 0.f) fy = 0.f;
        int y0 = (int)fy; int y1 = y0 + (y0 < h - 1 ? 1 : 0); float ly = fy - (float)y0;
        for (int x = 0; x < W; ++x) {
            float fx = ((float)x + 0.5f) * sw - 0.5f; if (fx < 0.f) fx = 0.f;
            int x0 = (int)fx; int x1 = x0 + (x0 < w - 1 ? 1 : 0); float lx = fx - (float)x0;
            for (int c = 0; c < Cin; ++c) {
                const float* f = feat + (int64_t)c * h * w;
                float v = (1.f - ly) * ((1.f - lx) * f[y0 * w + x0] + lx * f[y0 * w + x1])
                        + ly * ((1.f - lx) * f[y1 * w + x0] + lx * f[y1 * w + x1]);
                up[(int64_t)c * HW + (int64_t)y * W + x] = v;
            }
        }
    }
    double* mid = (double*)malloc(sizeof(double) * (size_t)Cmid);
    for (int y = 0; y < H; ++y)
        for (int x = 0; x < W; ++x) {
            for (int o = 0; o < Cmid; ++o) {
                double acc = 0.0;
                for (int c = 0; c < Cin; ++c)
                    for (int ky = 0; ky < 3; ++ky) {
                        int yy = y + ky - 1; if (yy < 0 || yy >= H) continue;
                        for (int kx = 0; kx < 3; ++kx) {
                            int xx = x + kx - 1; if (xx < 0 || xx >= W) continue;
                            acc += (double)w1[((int64_t)(o * Cin + c) * 3 + ky) * 3 + kx]
                                 * (double)up[(int64_t)c * HW + (int64_t)yy * W + xx];
                        }
                    }
                double v = acc * scale[o] + shift[o];
                mid[o] = v > 0.0 ? v : 0.0;
            }
            for (int k = 0; k < Cout; ++k) {
                double acc = b2[k];
                for (int o = 0; o < Cmid; ++o) acc += (double)w2[(int64_t)k * Cmid + o] * mid[o];
                out[(int64_t)k * HW + (int64_t)y * W + x] = (float)acc;
            }
        }
    free(mid); free(up);
}

/* ------------------------------------------------------------- next #1 --- */
/* ConfidenceCalibration.compute_ece accumulators, PKG/evaluation/metrics.py:
 * 161-176 per pixel: conf = max softmax prob (float32), pred = argmax, skip
 * label == 255 (:170), bin k with edges[k] < conf <= edges[k+1] (:188).
 * cnt[k], sum_conf[k] (float64), sum_correct[k]. */
ORC_API void orc_ece_bins(const float* logits, const void* label, int label_is_u8,
                          int64_t B, int C, int64_t HW, const float* edges, int n_bins,
                          int64_t* cnt, double* sum_conf, int64_t* sum_correct)
{
    for (int64_t b = 0; b < B; ++b)
        for (int64_t p = 0; p < HW; ++p) {
            int64_t t = label_is_u8 ? (int64_t)((const uint8_t*)label)[b * HW + p]
                                    : ((const int64_t*)label)[b * HW + p];
            if (t == 255) continue;
            const float* x = logits + (b * C) * HW + p;
            float m = x[0]; int bi = 0;
            for (int c = 1; c < C; ++c) { float v = x[(int64_t)c * HW]; if (v > m) { m = v; bi = c; } }
            float sum = 0.f;
            for (int c = 0; c < C; ++c) sum += expf(x[(int64_t)c * HW] - m);
            float conf = 1.0f / sum;
            for (int k = 0; k < n_bins; ++k)
                if (conf > edges[k] && conf <= edges[k + 1]) {
                    cnt[k] += 1; sum_conf[k] += (double)conf; sum_correct[k] += (bi == t);
                    break;
                }
        }
}


/* ------------------------------------------------------------- next #2 --- */
/* DepthEstimationPreprocessor._geometric_depth_estimation, PKG/data/
 * preprocessing.py:325-367, on a uint8 RGB frame:
 *   gray  = cv2.cvtColor(image, COLOR_RGB2GRAY)               (:338)
 *   base  = (y/h)*0.8 + 0.2; rows < h//3 -> 1.0; rows >= h//2 -> *0.5   (:340-354)
 *   tex   = cv2.Laplacian(gray, CV_64F)                       (:358)
 *   depth = clip(base - 0.3*|tex|/(max|tex| + 1e-8), 0, 1)    (:359-363)
 *   depth = scipy gaussian_filter(depth, sigma=2)             (:366)
 * cv2 is absent from this image, so the two OpenCV steps restate OpenCV 4.x's published
 * arithmetic and are PARITY UNPINNED: RGB2GRAY on 8-bit data is the 15-bit fixed-point
 * (R*9798 + G*19235 + B*3735 + 16384) >> 15 (imgproc color.hpp RY15/GY15/BY15, gray_shift);
 * Laplacian with the default ksize=1 is the 3x3 aperture [0 1 0; 1 -4 1; 0 1 0] with
 * BORDER_REFLECT_101, exact in float64.  The float64 ladder and the Gaussian are numpy/scipy
 * and are pinned (tests/golden: depth_estimate_* made with scipy.ndimage itself). */
static inline int orc_gray15(const uint8_t* px)
{
    return (px[0] * 9798 + px[1] * 19235 + px[2] * 3735 + (1 << 14)) >> 15;
}

ORC_API void orc_depth_estimate(const uint8_t* img, int H, int W, const double* taps, double* depth)
{
    int64_t hw = (int64_t)H * W;
    int* gray = (int*)malloc(sizeof(int) * (size_t)hw);
    int* lap = (int*)malloc(sizeof(int) * (size_t)hw);
    double* d = (double*)malloc(sizeof(double) * (size_t)hw);
    for (int64_t i = 0; i < hw; ++i) gray[i] = orc_gray15(img + i * 3);
    int mx = 0;
    for (int y = 0; y < H; ++y)
        for (int x = 0; x < W; ++x) {
            int ym = orc_reflect101(y - 1, H), yp = orc_reflect101(y + 1, H);
            int xm = orc_reflect101(x - 1, W), xp = orc_reflect101(x + 1, W);
            int v = gray[(int64_t)ym * W + x] + gray[(int64_t)yp * W + x] + gray[(int64_t)y * W + xm]
                  + gray[(int64_t)y * W + xp] - 4 * gray[(int64_t)y * W + x];
            v = v < 0 ? -v : v;
            lap[(int64_t)y * W + x] = v;
            if (v > mx) mx = v;
        }
    double denom = (double)mx + 1e-8;
    for (int y = 0; y < H; ++y) {
        double base = ((double)y / (double)H) * 0.8 + 0.2;
        if (y < H / 3) base = 1.0;
        if (y >= H / 2) base = base * 0.5;
        for (int x = 0; x < W; ++x) {
            double ts = (double)lap[(int64_t)y * W + x] / denom;
            double v = base + (-0.3 * ts);
            d[(int64_t)y * W + x] = v < 0.0 ? 0.0 : (v > 1.0 ? 1.0 : v);
        }
    }
    orc_gauss17(d, H, W, taps, depth);
    free(gray); free(lap); free(d);
}

ORC_API int orc_version(void) { return 1; }
