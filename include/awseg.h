/*
 * awseg.h — C ABI of libawseg_hip.so, the MI355X (gfx950) hot path of the
 * adverse-weather segmentation robustness benchmark.
 *
 * The reference (REF = A-SHOJAEI/adverse-weather-semantic-segmentation-robustness-benchmark,
 * PKG = REF/src/adverse_weather_semantic_segmentation_robustness_benchmark) is pure
 * Python and has no FFI of its own; each entry point below names the reference
 * function (file:line) whose arithmetic it replaces.  INTEGRATION.md shows the
 * ctypes stub a reference maintainer would add at each of those call sites.
 *
 * Conventions
 *   - every pointer is a DEVICE pointer unless the parameter is documented "host";
 *   - plain pointers and sizes only: no torch / HIP C++ types in any signature
 *     (`awseg_stream_t` is a `hipStream_t` passed as an opaque `void*`, NULL = the
 *     null stream);
 *   - every call is asynchronous and stream-ordered, never allocates, never
 *     synchronises, never throws; workspaces are caller-provided;
 *   - return value: 0 = success; a positive value is the `hipError_t` of the failed
 *     launch; a negative value is one of the AWSEG_E* codes below (argument check
 *     failed on the host, nothing was launched);
 *   - images are HWC uint8 (the loader's numpy layout, PKG/data/loader.py:206),
 *     logits are NCHW float32 (torch default, PKG/models/model.py:214),
 *     label / prediction maps are [B,H,W].
 */
#ifndef AWSEG_H
#define AWSEG_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

typedef void* awseg_stream_t;

/* argument-check failures (nothing launched) */
#define AWSEG_EINVAL   (-1) /* bad size / null pointer / unknown enum */
#define AWSEG_ERANGE   (-2) /* a size exceeds what the kernel's indexing supports */
#define AWSEG_EALIGN   (-3) /* pointer not aligned as documented */

/* element types of label / prediction maps */
#define AWSEG_U8   0
#define AWSEG_I64  1

/* ensemble strategies, PKG/models/model.py:443-458 */
#define AWSEG_COMBINE_WEIGHTED  0
#define AWSEG_COMBINE_MAXCONF   1
#define AWSEG_COMBINE_MEAN      2

/* base losses, PKG/models/model.py:548-553 */
#define AWSEG_LOSS_CE     0
#define AWSEG_LOSS_FOCAL  1

#define AWSEG_MAX_CLASSES 32   /* per-pixel channel vectors live in registers */
#define AWSEG_GAUSS_RADIUS 8   /* scipy gaussian_filter(sigma=2, truncate=4) -> 17 taps */

int         awseg_abi_version(void);          /* bumps when a signature changes */
/* 60 bits of sha256 of THIS file as it was when the library was built (csrc/build.py passes it to the compiler): a binding that
 * loads the library next to a header with another hash is looking at another ABI and must refuse to call it (the ctypes host side
 * does: adverse_weather_semantic_segmentation_robustness_benchmark_amd/_native.py).  0: built without the build script. */
unsigned long long awseg_header_hash(void);
const char* awseg_error_string(int code);     /* host string for any return code */
int         awseg_device_count(void);         /* hipGetDeviceCount, 0 when no GPU */

/* Per-image jobs of the batched weather kernels: HOST arrays (they travel in the kernel
 * arguments, 16 per launch).  `image` indexes the [B,H,W,3] batch for both input and output;
 * the job's own position j indexes the per-job device arrays (noise, depth_out).  Plain C
 * layout (numpy dtype with align=True reproduces it). */
typedef struct awseg_fog_job {
    int32_t  image; int32_t _pad;
    double   beta;            /* preprocessing.py:113 */
    double   atmos;           /* preprocessing.py:114, rounded to float32 inside the kernel */
    uint64_t seed;            /* Philox key, used only when noise == NULL */
} awseg_fog_job;

typedef struct awseg_night_job {
    int32_t  image; int32_t _pad;
    double   brightness;      /* 1 - I*U(.2,.6), preprocessing.py:212 */
    double   intensity;       /* preprocessing.py:207 */
    uint64_t seed;
} awseg_night_job;

typedef struct awseg_prim_job {
    int32_t  image;
    int32_t  prim_offset;     /* first primitive of this image in the shared primitive array */
    int32_t  prim_count;
    int32_t  blur_ksize;      /* snow: 3 or 7 (preprocessing.py:197); rain: ignored (3) */
    double   intensity;
} awseg_prim_job;

/* One frame of awseg_weather_batch: kind + the parameters of that kind.  a / b: fog beta / atmospheric light; night brightness /
 * intensity; rain, snow: intensity / unused.  seed: Philox key (fog, night).  prim_offset / prim_count: the frame's drops (rain:
 * int32[.,5] in rain_drops) or flakes (snow: int32[.,3] in snow_flakes). */
enum { AWSEG_WEATHER_CLEAN = 0, AWSEG_WEATHER_FOG = 1, AWSEG_WEATHER_RAIN = 2, AWSEG_WEATHER_SNOW = 3, AWSEG_WEATHER_NIGHT = 4 };
typedef struct awseg_weather_job {
    int32_t  kind; int32_t image;
    double   a; double b;
    uint64_t seed;
    int32_t  prim_offset; int32_t prim_count;
} awseg_weather_job;

/* ------------------------------------------------------------------------- *
 *  A13  IoUMetrics.compute_iou — confusion accumulation
 *       replaces PKG/evaluation/metrics.py:54-71
 * ------------------------------------------------------------------------- *
 * counts[(t*C [mod 256 when label_wrap_u8]) + p] += 1 for every pixel with
 * label t != ignore_index.  `counts` is int64[C*C], caller-zeroed, accumulated
 * in place (additive across calls, images and ranks).  label_wrap_u8 = 1
 * reproduces the reference's uint8 arithmetic of `targets * num_classes`
 * (metrics.py:68 on a uint8 tensor wraps mod 256 before promotion); with 0 the
 * product is exact (int64 labels).  A pixel whose flat index falls outside
 * [0, C*C) — or whose prediction is outside [0, C) — is not counted and
 * increments *oob (int64[1], caller-zeroed): the host raises IndexError as
 * torch's index_add_ does.  uint8 maps must be 16-byte aligned.
 * workspace: >= awseg_metrics_workspace(1, C, n) bytes (per-block partial
 * histograms; contents are scratch).
 */
int64_t awseg_metrics_workspace(int64_t batch, int num_classes, int64_t hw);
int awseg_confusion_accumulate(const void* pred, int pred_dtype,
                               const void* label, int label_dtype,
                               int64_t n, int num_classes, int ignore_index,
                               int label_wrap_u8,
                               int64_t* counts, int64_t* oob,
                               void* workspace, awseg_stream_t stream);

/* ------------------------------------------------------------------------- *
 *  A12  logits.argmax(dim=1)
 *       replaces REF/scripts/evaluate.py:179, PKG/training/trainer.py:447,
 *       PKG/evaluation/metrics.py:50-51
 * ------------------------------------------------------------------------- *
 * logits float32 [B,C,HW]; pred [B,HW] of pred_dtype.  First index wins on
 * ties, NaN compares as the maximum (torch semantics).
 */
int awseg_argmax(const float* logits, int64_t batch, int num_classes, int64_t hw,
                 void* pred, int pred_dtype, awseg_stream_t stream);

/* ------------------------------------------------------------------------- *
 *  A11+A12+A13  EnsembleModel combine -> /temperature -> argmax -> confusion
 *       replaces PKG/models/model.py:443-462, REF/scripts/evaluate.py:179,
 *       PKG/evaluation/metrics.py:54-71 in one pass over the member logits
 * ------------------------------------------------------------------------- *
 * seg1 (SegFormer) / seg2 (DeepLabV3+) float32 [B,C,HW].
 * weights: device float[2] = softmax(ensemble_weights) (WEIGHTED only, else may
 *          be NULL).  temperature: device float[1] or NULL (no scaling).
 * Arithmetic is four separately rounded float32 operations per element
 * ((w0*s1 + w1*s2) / T), no FMA contraction, as torch evaluates it.
 * Optional outputs (NULL = skip): out_logits float32 [B,C,HW]; pred [B,HW].
 * Optional fused confusion: label [B,HW] (NULL = skip), counts int64
 * [n_slots, C*C] where slot 0 receives every image and slot 1+cond[b] the
 * image's own weather condition (cond: device int32[B] or NULL -> slot 0 only;
 * cond[b] < 0 -> slot 0 only).  oob as in awseg_confusion_accumulate.
 * workspace: >= awseg_metrics_workspace(B, C, HW) bytes when label != NULL.
 */
int awseg_combine_argmax_confusion(const float* seg1, const float* seg2,
                                   int64_t batch, int num_classes, int64_t hw,
                                   int mode, const float* weights, const float* temperature,
                                   float* out_logits, void* pred, int pred_dtype,
                                   const void* label, int label_dtype, int ignore_index,
                                   int label_wrap_u8, const int32_t* cond,
                                   int64_t* counts, int n_slots, int64_t* oob,
                                   void* workspace, awseg_stream_t stream);

/* Single-model variant of the above: logits -> argmax -> confusion (77 B/px). */
int awseg_argmax_confusion(const float* logits, int64_t batch, int num_classes, int64_t hw,
                           void* pred, int pred_dtype,
                           const void* label, int label_dtype, int ignore_index,
                           int label_wrap_u8, const int32_t* cond,
                           int64_t* counts, int n_slots, int64_t* oob,
                           void* workspace, awseg_stream_t stream);

/* ------------------------------------------------------------------------- *
 *  A7  Normalize + ToTensorV2
 *       replaces PKG/data/loader.py:195-198, 275-278
 * ------------------------------------------------------------------------- *
 * imgs uint8 [B,H,W,3] -> out float32 [B,3,H,W], (x/255 - mean[c]) / std[c],
 * three separately rounded float32 operations.  sel: device int32[n_sel] image
 * indices to convert, or NULL for all B.  mean/std are host float[3].
 * H*W must be a multiple of 4.
 */
int awseg_normalize(const uint8_t* imgs, int64_t batch, int height, int width,
                    const int32_t* sel, int n_sel,
                    const float* mean_host, const float* std_host,
                    float* out, awseg_stream_t stream);

/* ------------------------------------------------------------------------- *
 *  next #4  WeatherAugmentationPipeline._apply_style_transfer
 *       replaces PKG/data/loader.py:360-387
 * ------------------------------------------------------------------------- *
 * Per-channel uint8 -> uint8 lookup: out[b,y,x,c] = luts[lut_of[b]][c][imgs[b,y,x,c]].
 * luts: device uint8 [n_luts,3,256]; lut_of: device int32 [batch], an index into luts or
 * < 0 to pass frame b through unchanged (a value >= n_luts is the caller's error, unchecked).
 * In place (out == imgs) is allowed.  H*W*3 must be a multiple of 4.
 */
int awseg_lut3_apply(const uint8_t* imgs, int batch, int height, int width,
                     const uint8_t* luts, int n_luts, const int32_t* lut_of,
                     uint8_t* out, awseg_stream_t stream);

/* ------------------------------------------------------------------------- *
 *  next #4  WeatherDegradationTransforms.get_fog_density_map — local contrast
 *       replaces PKG/data/preprocessing.py:270-278
 * ------------------------------------------------------------------------- *
 * imgs uint8 [batch,H,W,3] -> contrast float32 [batch,H,W] =
 * sqrt(box5((gray - box5(gray))^2)), gray = RGB2GRAY/255 in float32, box5 = the
 * 5x5 mean filter with BORDER_REFLECT_101.  The percentile / depth weighting of
 * :281-288 are three reductions the caller does on the result.
 */
int awseg_local_contrast(const uint8_t* imgs, int batch, int height, int width,
                         float* contrast, awseg_stream_t stream);

/* ------------------------------------------------------------------------- *
 *  A2  _generate_synthetic_depth
 *       replaces PKG/data/preprocessing.py:227-248
 * ------------------------------------------------------------------------- *
 * depth = max(gaussian_filter((y/H)*100 + noise, sigma=2), 1.0) in float64:
 * 17-tap separable filter, axis 0 first then axis 1, scipy 'reflect' border,
 * scipy's symmetric summation order.  noise: float64 [n_jobs,H,W] (the host's
 * np.random.normal(0,10) draws) or NULL -> N(0,10) from the in-kernel Philox
 * stream keyed by jobs[j].seed.  taps: host double[17] = the normalised kernel.
 * depth_out float64 [n_jobs,H,W].
 */
int awseg_synthetic_depth(int height, int width,
                          const awseg_fog_job* jobs, int n_jobs,
                          const double* noise, const double* taps_host,
                          double* depth_out, awseg_stream_t stream);

/* ------------------------------------------------------------------------- *
 *  next #2  DepthEstimationPreprocessor.estimate_depth
 *       replaces PKG/data/preprocessing.py:304-367 (called per sample by
 *       PKG/data/loader.py:270-272 when include_depth is set)
 * ------------------------------------------------------------------------- *
 * imgs uint8 [batch,H,W,3] (the weather-corrupted frames).  Per image:
 * gray = 8-bit RGB2GRAY, tex = 3x3 Laplacian (BORDER_REFLECT_101), depth =
 * clip(base(y) - 0.3*|tex|/(max|tex| + 1e-8), 0, 1) in float64 with base =
 * (y/H)*0.8+0.2, rows < H/3 -> 1, rows >= H/2 -> *0.5, then the sigma=2
 * Gaussian of A2 (taps_host: host double[17]).  Outputs float64 [batch,H,W]
 * (what the reference returns) and/or float32 (what the loader hands the model,
 * loader.py:290); either may be NULL, not both.  workspace: device memory of
 * awseg_depth_estimate_workspace(batch) bytes (per-image max |tex|, cleared
 * by the call).  Two launches, stream-ordered.
 */
size_t awseg_depth_estimate_workspace(int batch);
int awseg_depth_estimate(const uint8_t* imgs, int batch, int height, int width,
                         const double* taps_host, void* workspace,
                         double* depth_f64, float* depth_f32, awseg_stream_t stream);

/* ------------------------------------------------------------------------- *
 *  A3  _apply_fog            replaces PKG/data/preprocessing.py:113-123
 * ------------------------------------------------------------------------- *
 * t = exp(-beta*depth); out = trunc(clip(img/255 * t + (double)(float)A*(1-t), 0,1)*255)
 * in float64 (img/255 is float32, A is rounded to float32 first, as numpy does).
 * depth float64 [n_jobs,H,W].  out: uint8 [B,H,W,3] or NULL; norm_out: fused A7
 * output float32 [B,3,H,W] or NULL (at least one of the two).
 */
int awseg_fog_apply(const uint8_t* imgs, int height, int width,
                    const awseg_fog_job* jobs, int n_jobs, const double* depth,
                    uint8_t* out, float* norm_out,
                    const float* mean_host, const float* std_host,
                    awseg_stream_t stream);

/* A2+A3 in one kernel: depth never leaves the chip (noise tile + halo staged in LDS).
 * noise NULL -> Philox.  depth_out optional float64 [n_jobs,H,W] (NULL = not written). */
int awseg_fog_fused(const uint8_t* imgs, int height, int width,
                    const awseg_fog_job* jobs, int n_jobs,
                    const double* noise, const double* taps_host,
                    uint8_t* out, float* norm_out, double* depth_out,
                    const float* mean_host, const float* std_host,
                    awseg_stream_t stream);

/* ------------------------------------------------------------------------- *
 *  A6  _apply_night          replaces PKG/data/preprocessing.py:209-225
 * ------------------------------------------------------------------------- *
 * v = ((img/255)*f32(brightness)) * f32(gain[c])   (float32, two roundings)
 * out = trunc(clip((double)v + noise*intensity*0.5, 0,1)*255)  (float64)
 * noise: float64 [n_jobs,H,W,3] (host draws of np.random.normal(0, 5/255)) or
 * NULL -> Philox N(0, 5/255).  gains are the reference's 0.8 / 0.85 / 1.2
 * (host float[3]).  H*W*3 must be a multiple of 4.
 */
int awseg_night_apply(const uint8_t* imgs, int height, int width,
                      const awseg_night_job* jobs, int n_jobs,
                      const double* noise, const float* gains_host,
                      uint8_t* out, float* norm_out,
                      const float* mean_host, const float* std_host,
                      awseg_stream_t stream);

/* A1 + A3-A7 for a batch of frames of MIXED conditions in ONE launch (throughput mode: in-kernel Philox noise for fog and night):
 * apply_weather_effect's dispatch (PKG/data/preprocessing.py:61-92) over the frames of a batch as PKG/data/loader.py:265-278 draws them,
 * each frame one job (<= 16 per call), fused Normalize/ToTensor output norm_out float32 [B,3,H,W] (required) and optional uint8
 * out [B,H,W,3] (not written for clean frames; must not alias imgs).  Same kernels' bodies as awseg_normalize / awseg_fog_fused /
 * awseg_rain_apply / awseg_snow_apply (3x3 blur only) / awseg_night_apply: identical bytes.  width % 4 == 0, width >= 16, else
 * AWSEG_ERANGE (the caller uses the per-kind entry points, which take every size).  workspace: awseg_streak_workspace(rain + snow
 * frames, H, W) bytes for the coverage maps (NULL when the batch has neither).  taps_host: the 17 Gaussian taps of the synthetic
 * depth; gains_host: night's three channel gains. */
int awseg_weather_batch(const uint8_t* imgs, int height, int width, const awseg_weather_job* jobs, int n_jobs,
                        const int32_t* rain_drops, const int32_t* snow_flakes, const double* taps_host, const float* gains_host,
                        uint8_t* out, float* norm_out, const float* mean_host, const float* std_host, void* workspace,
                        awseg_stream_t stream);

/* ------------------------------------------------------------------------- *
 *  A4  _apply_rain           replaces PKG/data/preprocessing.py:131-168
 * ------------------------------------------------------------------------- *
 * haze: v = v*(1-0.3I) + 0.3I*0.7 (float32); streaks: line segments
 * (x0,y0,x1,y1,thickness) painted with colour (.8,.9,1.0); 3x3 Gaussian blur
 * sigma 0.5, BORDER_REFLECT_101, float32; quantise.  drops: device int32[n,5],
 * jobs[j] (host) names its slice.  OpenCV's rasteriser is not available offline: the
 * coverage rule is stated in oracle/awseg_oracle.c (parity unpinned).
 * out must not alias imgs.
 */
int awseg_rain_apply(const uint8_t* imgs, int height, int width,
                     const awseg_prim_job* jobs, int n_jobs, const int32_t* drops,
                     uint8_t* out, float* norm_out,
                     const float* mean_host, const float* std_host,
                     void* workspace, awseg_stream_t stream);

/* Device scratch for awseg_rain_apply / awseg_snow_apply: with a workspace of this many bytes the primitives of a frame are
 * rasterised ONCE into a 1-bit coverage map (one wave per primitive) and the blur kernel is a pure stencil; with
 * workspace == NULL every 64x32 tile rasterises the primitives that touch it.  Same bytes out either way. */
int64_t awseg_streak_workspace(int n_jobs, int height, int width);

/* ------------------------------------------------------------------------- *
 *  A5  _apply_snow           replaces PKG/data/preprocessing.py:176-202
 * ------------------------------------------------------------------------- *
 * v = clip(v + 0.2I, 0, 1); filled discs (x,y,r) of 1.0; Gaussian blur ksize 3
 * or 7, sigma 1.0, BORDER_REFLECT_101, float32; quantise.  flakes: device
 * int32[n,3].  out must not alias imgs.
 */
int awseg_snow_apply(const uint8_t* imgs, int height, int width,
                     const awseg_prim_job* jobs, int n_jobs, const int32_t* flakes,
                     uint8_t* out, float* norm_out,
                     const float* mean_host, const float* std_host,
                     void* workspace, awseg_stream_t stream);

/* ------------------------------------------------------------------------- *
 *  A16  AdverseWeatherTrainer._estimate_fog_density
 *       replaces PKG/training/trainer.py:494-511 (and its .to(device) at :321)
 * ------------------------------------------------------------------------- *
 * density[b] = u * scale[b] + offset[b] in float32 (two roundings, as torch evaluates
 * `torch.rand(h, w) * a + b`).  scale_offset: device float32 [B,2] ({.5,.5} fog,
 * {.3,.2} rain/snow, {.1,0} else).
 *   uniform == NULL  throughput mode: u from in-kernel Philox keyed by (seed, pixel) —
 *                    nothing per-pixel crosses PCIe; parity in distribution only;
 *   uniform != NULL  parity mode: device float32 [B,hw] holding the host's draws (the
 *                    reference calls torch.rand(h, w) once per sample, in sample order, on
 *                    torch's CPU generator): the result is bit-identical to the reference's.
 */
int awseg_fog_density_field(const float* scale_offset, int batch, int64_t hw, uint64_t seed,
                            const float* uniform, float* density, awseg_stream_t stream);

/* ------------------------------------------------------------------------- *
 *  A15  FogDensityAwareLoss  replaces PKG/models/model.py:577-587, 610, 638-642
 * ------------------------------------------------------------------------- *
 * Forward: per pixel ce = logsumexp(x) - x[label]; focal: (1-exp(-ce))^2 * ce;
 * times (1 + sensitivity*density) when density != NULL.  Writes per-block
 * partial sums (float64) into `partials` (>= awseg_loss_partials(B, HW)
 * doubles) and the float32 mean into loss_mean[0] via a second tiny launch.
 * pixel_loss: optional float32 [B,HW] output.  A label outside [0,C) raises
 * *oob (torch raises IndexError; ignore_index=-100 never occurs for uint8).
 * Backward: grad_logits[b,c,p] = g * w_p * (softmax_c - [c==label]) * focal' / N
 * with g = grad_scale[0] (device float, the upstream gradient of the mean).
 */
int64_t awseg_loss_partials(int64_t batch, int64_t hw);
int awseg_fog_ce_forward(const float* logits, const void* label, int label_dtype,
                         const float* density, int64_t batch, int num_classes, int64_t hw,
                         int base_loss, float sensitivity,
                         float* pixel_loss, double* partials, float* loss_mean,
                         int64_t* oob, awseg_stream_t stream);
int awseg_fog_ce_backward(const float* logits, const void* label, int label_dtype,
                          const float* density, int64_t batch, int num_classes, int64_t hw,
                          int base_loss, float sensitivity, const float* grad_scale,
                          float* grad_logits, awseg_stream_t stream);

/* _estimate_fog_density_from_depth, PKG/models/model.py:658-677.
 * depth float32 [B,H,W] -> density float32 [B,H,W]; statistics (min, max, mean
 * gradient magnitude) are global over the whole batch as in the reference.
 * workspace: >= awseg_density_workspace(B, H*W) bytes, 16-byte aligned. */
int64_t awseg_density_workspace(int64_t batch, int64_t hw);
int awseg_fog_density_from_depth(const float* depth, int64_t batch, int height, int width,
                                 float* density, void* workspace, awseg_stream_t stream);

/* ------------------------------------------------------------------------- *
 *  A8  SegFormer segmentation head, upsample fused away
 *       replaces PKG/models/model.py:209-214 (F.interpolate x32 -> Conv3x3 ->
 *       BatchNorm(eval) -> ReLU -> Conv1x1)
 * ------------------------------------------------------------------------- *
 * The full-resolution [B,Cin,H,W] tensor is never materialised: bilinear
 * upsampling and the 3x3 convolution are both linear, so
 * conv3x3(up(f)) = sum_tap shift_tap(up(W_tap f)).
 *   g9     float32 [B, h, w, 9, Cmid] = per-tap 1x1 products W_tap . f at the
 *          encoder's resolution (tap = ky*3+kx, channel-last), computed by the
 *          caller with one plain GEMM
 *   scale/shift float32 [Cmid]: conv bias + eval-mode BatchNorm folded to y*scale+shift;
 *          scale may be NULL when the caller has already multiplied it into g9 (cheaper epilogue)
 *   w2     float32 [Cout, Cmid], b2 float32 [Cout]: the 1x1 classifier (Cout <= 32)
 * out float32 [B,Cout,H,W].  Zero padding of the 3x3 at the image border and
 * align_corners=False source coordinates follow torch exactly.  Cmid % 32 == 0.
 */
int awseg_segformer_head_fused(const float* g9, int64_t batch, int cmid, int h, int w,
                               int height, int width,
                               const float* scale, const float* shift,
                               const float* w2, const float* b2, int cout,
                               float* out, awseg_stream_t stream);

/* The same head on v_mfma_f32_32x32x16_f16 with split operands (x = hi + lo in f16, three products per
 * float32-grade product, float32 accumulation): both contractions run at the f16 rate.  Rows in
 * which an operand reaches 2^15 are recomputed inside the kernel on the float32 instruction, so the result
 * is float32-grade for any finite input.  Same arguments; scale must be NULL (folded into g9), Cmid 128 or
 * 256; AWSEG_ERANGE for a geometry or width the kernel does not cover (the caller then uses the entry above).
 * Replaces the same reference lines (PKG/models/model.py:209-214). */
int awseg_segformer_head_fused_split(const float* g9, int64_t batch, int cmid, int h, int w,
                                     int height, int width,
                                     const float* scale, const float* shift,
                                     const float* w2, const float* b2, int cout,
                                     float* out, awseg_stream_t stream);

/* First stage only (no classifier): relu(bn(conv3x3(interpolate(f)))) written at full
 * resolution, out float32 [B,Cmid,H,W] (channels_last = 0) or [B,H,W,Cmid] (channels_last = 1).  Used for the first 3x3 of DepthEstimationHead on
 * the SegFormer branch (PKG/models/model.py:42-45 applied to the upsampled features, :219-221).
 * Needs the MFMA geometry (Cmid in {32,64,128,256}, upsample factor >= ~17); else AWSEG_ERANGE. */
int awseg_upconv3x3_bn_relu(const float* g9, int64_t batch, int cmid, int h, int w,
                            int height, int width,
                            const float* scale, const float* shift,
                            float* out, int channels_last, awseg_stream_t stream);

/* ------------------------------------------------------------------------- *
 *  A9  DeepLabV3+ ASPP: depthwise atrous 3x3 of all three rates in one pass
 *       replaces the three SeparableConv2d depthwise halves of smp's ASPP
 *       (call site PKG/models/model.py:259-265, 349)
 * ------------------------------------------------------------------------- *
 * x float32 NHWC [B,h,w,C]; wdw float32 [3 rates][9 taps][C]; out float32
 * [3][B,h,w,C] (NHWC per rate) ready for the pointwise GEMMs.  C % 4 == 0.
 */
int awseg_aspp_depthwise3(const float* x, int64_t batch, int h, int w, int channels,
                          const float* wdw, int rate0, int rate1, int rate2,
                          float* out, awseg_stream_t stream);
/* The same pass that also leaves ASPPPooling's global average in mean_out float32 [batch, channels] (nn.AdaptiveAvgPool2d(1) of
 * the smp ASPP's pooling branch): the rate-0 blocks of the LDS-staged walk meet every pixel of their channel slice once, so the mean
 * is a running sum beside the stencil instead of a pass over the 2048-channel map of its own.  AWSEG_ERANGE where that walk does not
 * apply (channels % 32 != 0, or width * 8 outside 64 .. 1024): the caller takes awseg_aspp_depthwise3 and a mean of its own. */
int awseg_aspp_depthwise3_mean(const float* x, int64_t batch, int h, int w, int channels, const float* wdw,
                               int rate0, int rate1, int rate2, float* out, float* mean_out, awseg_stream_t stream);

/* ------------------------------------------------------------------------- *
 *  channel-last helpers around the torch-ROCm backbones (callers of A8 / A9)
 * ------------------------------------------------------------------------- *
 * awseg_dwconv3x3_nhwc: depthwise 3x3, stride 1, dilation d, zero padding d, on float32
 * [B,H,W,C] (C % 4 == 0) with fused per-channel bias (nullable) and activation:
 * act 0 none, 1 ReLU, 2 exact (erf) GELU.  w9 float32 [9 taps][C].  Replaces
 * transformers' SegformerDepthWiseConv (nn.Conv2d(dim, dim, 3, 1, 1, groups=dim)) + GELU inside
 * the encoder PKG/models/model.py:193 calls, and the depthwise halves of smp's SeparableConv2d
 * in the DeepLabV3+ decoder (call site PKG/models/model.py:349).  out must not alias x.
 */
#define AWSEG_ACT_NONE 0
#define AWSEG_ACT_RELU 1
#define AWSEG_ACT_GELU 2
int awseg_dwconv3x3_nhwc(const float* x, int64_t batch, int height, int width, int channels, int dilation,
                         const float* w9, const float* bias, int act, float* out, awseg_stream_t stream);

/* awseg_conv3x3_winograd_nhwc: 3x3, stride 1, dilation d, zero padding d convolution of float32
 * [B,H,W,Cin] -> [B,H,W,Cout] as Winograd F(2x2,3x3) on the fp32 matrix cores, with the eval-mode
 * BatchNorm folded in.  u: device float32 [Cin/8][16][2][Cout][4] = (G g G^T) of the (scale-folded) 3x3
 * filters in the kernel's LDS image order: position p = 4i+j of input channel 8*chunk + 4*(s>>1) + 2*hk + (s&1) at
 * [chunk][p][hk][cout][s]; shift float32 [Cout].  Cin % 16 == 0, Cout % 64 == 0.
 *   w2 == NULL: out[b,y,x,n] = act(conv + shift[n] (+ residual[b,y,x,n])), act 0 none / 1 ReLU;
 *   w2 != NULL (Cout == 64, residual NULL): the whole tail of DepthEstimationHead
 *       (PKG/models/model.py:47-51: Conv3x3 -> BN -> ReLU -> Conv1x1 -> Sigmoid) in one launch:
 *       out[b,y,x] = sigmoid(b2[0] + sum_n w2[n] * relu(conv + shift[n])), float32 [B,H,W].
 * Replaces the depth-head 3x3s at PKG/models/model.py:42-52 (called from :219-221, :358-371) and
 * the 3x3 convolutions of the ResNet-50 bottlenecks behind :349 in the eval forward.
 * Result differs from a direct fp32 convolution by the Winograd transforms' rounding (~1e-6 rel). */
int awseg_conv3x3_winograd_nhwc(const float* x, int batch, int height, int width, int cin, int cout, int dilation,
                                const float* u, const float* shift, const float* residual, int act,
                                const float* w2, const float* b2, float* out, awseg_stream_t stream);

/* awseg_gemm_bias_act: out[M,N] = act(x[M,K] . w[N,K]^T + bias[N] (+ residual[M,N])), row-major float32,
 * act 0 none / 1 ReLU — a 1x1 stride-1 convolution of an NHWC tensor with the eval-mode BatchNorm folded
 * into w / bias and the bottleneck's identity as `residual` (the 1x1s of the ResNet-50 encoder and the
 * decoder projections behind PKG/models/model.py:349).  One hipBLASLt matmul (the library GEMM) whose
 * epilogue does the bias, the residual (beta*C) and the ReLU, so no separate pass touches the output.
 * residual may alias out.  workspace: device scratch for the library (32 MiB is plenty; pass the same size
 * for the same shape — the algorithm choice is cached per (M,N,K,residual,act,workspace_bytes)).  The
 * library handle and the cached choices are host state created on first use. */
int awseg_gemm_bias_act(const float* x, const float* w, const float* bias, const float* residual, int act,
                        float* out, int64_t m, int n, int k, void* workspace, size_t workspace_bytes,
                        awseg_stream_t stream);

/* awseg_gemm_tune: OPTIONAL and SYNCHRONISING (the only entry point that waits on the device) — times the
 * library's ranked algorithm candidates for one awseg_gemm_bias_act problem on the caller's buffers and
 * keeps the fastest for later calls with the same (M,N,K,residual,act,workspace_bytes).  scratch_out [M,N]
 * is overwritten and also stands in for the residual.  Returns the number of candidates timed or < 0.
 * Meant for a warm-up pass; awseg_gemm_bias_act never calls it. */
int awseg_gemm_tune(const float* x, const float* w, const float* bias, int has_residual, int act,
                    float* scratch_out, int64_t m, int n, int k, void* workspace, size_t workspace_bytes,
                    awseg_stream_t stream);

/* awseg_gemm_split_weights / awseg_gemm_split_bias_act: the operator of awseg_gemm_bias_act computed in float32 grade
 * on the f16 matrix cores with SPLIT operands (x = f16(x) + f16(x - f16(x)): 22 significant bits; x*w = xh*wh + xh*wl +
 * xl*wh as three f16 MFMA products, float32 accumulation) — for the compute-bound 1x1 convolutions (ResNet layer3 /
 * layer4, ASPP, decoder; PKG/models/model.py:349), where it runs at several times the float32-input MFMA rate.
 * awseg_gemm_split_weights writes, from w float32 [N][K], once per weight: uint16 [2][N][K] (f16 bit patterns of the high
 * parts, then of the low parts scaled by 2^11, of w * 2^-ew, where ew brings max|w| — found on the device — into
 * [2^13, 2^14)), 8 trailing uint16 holding {max|w| bits, ew, 0, 0} as uint32, and — when N >= 8 and K % 8 == 0 — a
 * second, k-blocked image uint16 [N'/T][ceil(K/32)][T][32 high | 32 low (unscaled)] (T = 256, 128 or 64 rows a tile: the
 * largest that divides N, else 64 with N' = N rounded up and zero rows behind N; zeros past K) that the LDS-DMA
 * kernel (csrc/gemm_split3.hip) streams into LDS without a staging pass.  awseg_gemm_split_weight_halfs(n, k) = the uint16
 * the buffer must hold (2NK + 8, plus 2 N' ceil32(K) with the second image): size the buffer with it, never by hand.  awseg_gemm_split_bias_act: x float32 [M][K], bias float32 [N] or NULL, residual float32 [M][N] or NULL (may
 * alias out), act 0 none / 1 ReLU, out float32 [M][N].  K % 8 == 0; x and w_split 16-byte aligned.  Operand range: any
 * finite float32.  Activations are split optimistically; a block that meets |x| >= 2^15 (2^11 in the single-accumulator
 * kernels) in its A tiles recomputes that output tile with x * 2^-e (exact) and multiplies 2^e back in the epilogue — twice
 * the time for that tile, same accuracy.  Inf / NaN propagate.  No workspace, no host state. */
int64_t awseg_gemm_split_weight_halfs(int n, int k);
int awseg_gemm_split_weights(const float* w, int n, int k, uint16_t* w_split, awseg_stream_t stream);
int awseg_gemm_split_bias_act(const float* x, const uint16_t* w_split, const float* bias, const float* residual, int act,
                              float* out, int64_t m, int n, int k, awseg_stream_t stream);

/* awseg_dwconv3x3_upcat_nhwc: the depthwise 3x3 (stride 1, zero padding 1, no bias) of the DeepLabV3+ decoder's
 * block2 applied to cat(UpsamplingBilinear2d(align_corners=True)(a), hi) without materialising the upsampled
 * map or the concatenation: a float32 [B,h,w,Ca] (the ASPP branch at stride 16), hi float32 [B,H,W,Ch] (the
 * 48-channel skip at stride 4), w9 float32 [9 taps][Ca+Ch], out float32 [B,H,W,Ca+Ch].  Replaces the
 * up -> cat -> depthwise part of smp's DeepLabV3PlusDecoder.forward behind PKG/models/model.py:349. */
int awseg_dwconv3x3_upcat_nhwc(const float* a, int a_height, int a_width, int a_channels, const float* hi, int hi_channels,
                               int64_t batch, int height, int width, const float* w9, float* out, awseg_stream_t stream);

/* awseg_conv3x3_winograd_split_nhwc: the operator of awseg_conv3x3_winograd_nhwc (same arguments, layouts, epilogues
 * and error codes) with its 16 position GEMMs on the f16 matrix cores and SPLIT float32 operands (V = f16(V) + f16(V -
 * f16(V)), three f16 products per float32-grade product, float32 accumulation): 16x the multiply-adds per cycle of the
 * float32-input MFMA that bounds the other kernel.  u_split: uint16 [Cin/16][16 positions][Cout/32][hi k 0-7 | hi k 8-15 |
 * lo k 0-7 | lo k 8-15][32 couts][8] f16 bit patterns of U * 2^-eu (U = G (w * bn_scale) G^T, max|U * 2^-eu| in
 * [2^13, 2^14)), followed by 2^eu as one float32 (a 16-byte trailer): awseg_winograd_split_weight_halfs(cin, cout) uint16
 * in front of the trailer.  Cin % 16 == 0, Cout % 64 == 0.  Operand range: any finite float32 activations — a block whose
 * patch maximum is >= 2^13 or < 2^-4 redoes its tile with power-of-two scaled activations. */
int64_t awseg_winograd_split_weight_halfs(int cin, int cout);

/* awseg_upconv_forms / awseg_depth_head_fused: DepthEstimationHead on the SegFormer branch (PKG/models/model.py:42-52 applied to
 * F.interpolate(features, (H, W), bilinear, align_corners=False), :211, :219-221) for H = 32 h, W = 32 w in ONE full-resolution
 * launch — the hidden map between its two 3x3 convolutions (Cmid channels at H x W: 8.6 GB per batch of 8 at 1024 x 2048) is never
 * written.  Inside one cell of the x32 upsampling the first convolution's pre-activation is an exact bilinear form of the pixel
 * coordinates; awseg_upconv_forms builds those forms per frame at the encoder's resolution from g9 float32 [B,h,w,9,Cmid] (the
 * per-tap products (W_tap * bn_scale) . f, the operand of awseg_upconv3x3_bn_relu) and the folded BatchNorm shift [Cmid]:
 *   forms float32 [B][ F4 [3h+3][w+1][4][Cmid] | F2 [3h+3][2w+2][2][Cmid] ]   (awseg_upconv_forms_floats(h, w, cmid) per frame;
 * row selector = cell row ky + 1, then h + 1 + special row id; special ids: 0 -> pixel 0, 1 + 2k -> 15 + 32k, 2 + 2k -> 16 + 32k,
 * 2n + 1 -> 32n - 1: csrc/depthfuse.hip).  awseg_depth_head_fused evaluates them per 16 x 16 tile inside the Winograd kernel of
 * awseg_conv3x3_winograd_split_nhwc / _bf16_nhwc (u_split / u_is_bf16: that operator's weight image of the SECOND 3x3, Cmid ->
 * 64, BatchNorm scale folded; shift2 its folded shift [64]) and finishes with ReLU -> Conv1x1 (w2 [64], b2 [1]) -> sigmoid:
 * out float32 [B,H,W].  Cmid % 16 == 0; forms, u_split, shift2, w2 16-byte aligned.  AWSEG_ERANGE: sizes beyond 32-bit offsets. */
int64_t awseg_upconv_forms_floats(int h, int w, int cmid);
int awseg_upconv_forms(const float* g9, int batch, int cmid, int h, int w, const float* shift, float* forms, awseg_stream_t stream);
int awseg_depth_head_fused(const float* forms, int batch, int h, int w, int cmid, const uint16_t* u_split, int u_is_bf16,
                           const float* shift2, const float* w2, const float* b2, float* out, awseg_stream_t stream);
int awseg_conv3x3_winograd_split_nhwc(const float* x, int batch, int height, int width, int cin, int cout, int dilation,
                                      const uint16_t* u_split, const float* shift, const float* residual, int act,
                                      const float* w2, const float* b2, float* out, awseg_stream_t stream);

/* awseg_gemm_split_pieces_bias_act: the split-operand GEMM over an A operand that lies in 2 .. 4 equally wide pieces along K —
 *     out[M,N] = act([x_0 | x_1 | ...] . w^T + bias (+ residual)),  x_i float32 [M,k_piece] (pieces: HOST array of device pointers),
 * w_split = awseg_gemm_split_weights of w [N, n_pieces*k_piece].  The 1x1 projection behind smp's ASPP (the model built at
 * PKG/models/model.py:262-268) is project(cat(branches)) = sum_i branch_i . P_i^T: with the four pixel branches as pieces the sum is
 * ONE product instead of four accumulating launches, and the concatenated map still never exists.  k_piece % 32 == 0; LDS-DMA kernel
 * only (AWSEG_ERANGE otherwise).  Epilogue, range guard and error codes as awseg_gemm_split_bias_act. */
int awseg_gemm_split_pieces_bias_act(const float* const* pieces, int n_pieces, int k_piece, const uint16_t* w_split, const float* bias,
                                     const float* residual, int act, float* out, int64_t m, int n, awseg_stream_t stream);

/* awseg_gemm_split_dual_bias_act: the split-operand GEMM with its A operand in TWO pieces along K —
 *     out[M,N] = act([x | x2] . w^T + bias (+ residual)),  x float32 [M,k1],  w_split = awseg_gemm_split_weights of w [N, k1+k2]
 * where x2 is either float32 rows [M,k2] (x2_stride == 0) or an NHWC image [batch, x2_height, x2_width, k2] whose pixels
 * (b, oy*x2_stride, ox*x2_stride) are the rows m = (b, oy, ox) — a 1x1 convolution of that stride folded into the product.  It is
 * the tail of the FIRST bottleneck of a ResNet stage (the smp encoder behind PKG/models/model.py:259-265, :349):
 *     relu(bn3(conv3(z)) + bn_d(downsample(block input)))  =  relu([z | input_s] . [W3' | Wd']^T + (b3' + bd'))
 * in one launch: the downsample branch's map is neither written nor read back as a residual.  k1 % 32 == k2 % 32 == 0, N >= 8;
 * LDS-DMA kernel of csrc/gemm_split3.hip only (AWSEG_ERANGE for shapes it does not take: the caller runs the two GEMMs).  Range
 * guard, epilogue and error codes as awseg_gemm_split_bias_act. */
int awseg_gemm_split_dual_bias_act(const float* x, int k1, const float* x2, int k2, int64_t batch, int x2_height, int x2_width,
                                   int x2_stride, const uint16_t* w_split, const float* bias, const float* residual, int act,
                                   float* out, int64_t m, int n, awseg_stream_t stream);

/* A convolution as ONE split-operand GEMM without the im2col matrix: x float32 NHWC [batch, height, width, channels]
 * (channels % 32 == 0), weights in im2col column order (ky, kx, c) split by awseg_gemm_split_weights
 * ([n, kernel_h * kernel_w * channels]); row (b, oy, ox) of the A operand is gathered from x while the K tiles are staged
 * (zero outside the image).  out float32 [batch * Ho * Wo, n] = NHWC; bias / residual / act as awseg_gemm_split_bias_act.
 * Same tiles, same summation order as awseg_im2col_nhwc + awseg_gemm_split_bias_act: bit-identical results.  Used for the
 * stride-2 3x3 / patch convolutions and the stride-2 1x1 downsample branches (reference: the smp ResNet encoder and the
 * HF SegFormer patch embeddings / sequence reductions built at PKG/models/model.py:262-268, :160-166). */
int awseg_conv_gemm_split_bias_act(const float* x, int64_t batch, int height, int width, int channels,
                                   int kernel_h, int kernel_w, int stride, int pad, int dilation,
                                   const uint16_t* w_split, const float* bias, const float* residual, int act,
                                   float* out, int n, awseg_stream_t stream);

/* The 7x7 stems on 3 input channels (the smp ResNet encoder's conv1: 7x7, stride 2, padding 3, behind PKG/models/model.py:262-268;
 * the first SegFormer patch embedding: 7x7, stride 4, 32 channels, behind :160-166) as the same GEMM with a ROW-gathered A operand.  x float32 [batch, height, width_padded, pixel_floats]: the image with
 * pixel_floats (4) floats a pixel — channels past the real ones zero — and zero columns on both sides: `pad` of them on the left,
 * enough on the right for the last output column's run.  Row (b, oy, ox) of the A operand is, per kernel row ky, the run of 32
 * consecutive floats (8 pixels) that starts at padded pixel ox * stride of image row oy * stride - pad_y + ky (zeros outside the
 * image rows), K = kernel_h * 32; the weights are [n, kernel_h * 32] with column ky * 32 + kx * pixel_floats + c (zero where
 * kx >= kernel_w or c is a padding channel), split by awseg_gemm_split_weights.  out float32 [batch * Ho * out_width, n] = NHWC.
 * AWSEG_ERANGE when the shape does not fit the LDS-DMA kernel (n neither a multiple of 64 nor below it, too few tiles): the
 * caller keeps its other path. */
int awseg_conv_rows_gemm_split_bias_act(const float* x, int64_t batch, int height, int width_padded, int pixel_floats,
                                        int kernel_h, int stride, int pad_y, int out_width, const uint16_t* w_split,
                                        const float* bias, const float* residual, int act, float* out, int n,
                                        awseg_stream_t stream);

/* ------------------------------------------------------------------------- *
 *  BASELINE config 5: the bf16 MFMA path (SegFormer-B5 + DeepLabV3+-R101)
 * ------------------------------------------------------------------------- *
 * The three dense contractions of the eval forward with ONE v_mfma_f32_32x32x16_bf16 per product tile instead of
 * three f16 products on split operands: operands are rounded to bfloat16 (round to nearest even) when they are staged,
 * accumulation, softmax, Winograd transforms and epilogues stay float32, tensors in memory stay float32.  Same
 * arguments, layouts, epilogues and error codes as the split-operand entry points they mirror; bfloat16 has float32's
 * exponent range, so there is no operand-range guard.  The reference has no counterpart (it cannot select these
 * backbones or a reduced precision: PKG/models/model.py:409-417); tolerance against the float32 path is stated in
 * tests/test_gpu_bf16.py.
 *   awseg_gemm_bf16_weights / awseg_gemm_bf16_bias_act  <->  awseg_gemm_split_weights / awseg_gemm_split_bias_act
 *       (w_bf16: uint16 [2][N][K] + 8 like w_split; plane 0 holds bf16(w), plane 1 is unused, trailer exponent 0)
 *   awseg_conv3x3_winograd_bf16_nhwc                    <->  awseg_conv3x3_winograd_split_nhwc
 *       (u_bf16: the same image with bf16(U * 2^-eu) in the "high" slots; the "low" slots are not read)
 *   awseg_attention_d32_bf16                            <->  awseg_attention_d32_split */
/* uint16 the awseg_gemm_bf16_weights buffer must hold: 2NK + 8, plus N ceil32(K) for the k-blocked bf16 image
 * [N/256][ceil(K/32)][256][32] that csrc/gemm_split3.hip streams into LDS (N % 256 == 0, K % 8 == 0). */
int64_t awseg_gemm_bf16_weight_halfs(int n, int k);
int awseg_gemm_bf16_weights(const float* w, int n, int k, uint16_t* w_bf16, awseg_stream_t stream);
int awseg_gemm_bf16_bias_act(const float* x, const uint16_t* w_bf16, const float* bias, const float* residual, int act,
                             float* out, int64_t m, int n, int k, awseg_stream_t stream);
int awseg_conv3x3_winograd_bf16_nhwc(const float* x, int batch, int height, int width, int cin, int cout, int dilation,
                                     const uint16_t* u_bf16, const float* shift, const float* residual, int act,
                                     const float* w2, const float* b2, float* out, awseg_stream_t stream);
int awseg_attention_d32_bf16(const float* q, const float* k, const float* v, float* out, int batch, int heads,
                             int n_queries, int n_keys, float scale, awseg_stream_t stream);
/* awseg_attention_d32_split with the keys and values PREPARED once per launch: a first small kernel writes every (image, head)'s
 * key / value tiles as the query blocks want them in LDS (f16 high | scaled low parts, V transposed into the PV operand order,
 * 9 216 bytes per 32 keys) and settles the key / value operand range there (exponents in a table); the query blocks (1 024 per image
 * in MiT stage 1) fetch tiles by LDS-DMA instead of each splitting the same 2 048 keys again.  k / v: float32 rows of kv_pitch floats
 * (0: heads*32; 2*heads*32 with v = k + heads*32 for the packed rows of awseg_attention_d32_packed_kv).  workspace:
 * awseg_attention_d32_split_workspace(batch, heads, n_keys) bytes, 256-byte aligned, scratch.  Same arithmetic per element as
 * awseg_attention_d32_split: same values. */
int64_t awseg_attention_d32_split_workspace(int batch, int heads, int n_keys);
int awseg_attention_d32_split_ws(const float* q, const float* k, const float* v, int kv_pitch, float* out, int batch, int heads,
                                 int n_queries, int n_keys, float scale, void* workspace, awseg_stream_t stream);

/* The same three attention kernels on PACKED keys and values: kv float32 [B, n_keys, 2*heads*32] holds a token's key in its first
 * heads*32 floats and its value in the last — what ONE GEMM with the key and value projections' weights stacked ([2C, C]) writes,
 * so that SegformerEfficientSelfAttention's two projections of the reduced tokens (transformers modeling_segformer.py, behind
 * PKG/models/model.py:186-200) are one launch.  mode 0: float32 MFMA, 1: split operands, 2: bf16.  Same results as the unpacked
 * entry points on the same numbers (the kernels only differ in the row pitch they read keys and values with). */
int awseg_attention_d32_packed_kv(const float* q, const float* kv, float* out, int batch, int heads, int n_queries,
                                  int n_keys, float scale, int mode, awseg_stream_t stream);

/* awseg_upconv3x3_linear / awseg_upconv3x3_adjoint: the TRAINING form of the upsample-free head stage (BASELINE config 4).
 * PKG/models/model.py:209-214 runs conv3x3(F.interpolate(f, (H,W), bilinear, align_corners=False)) at full resolution (2.47
 * TFLOP and a 2.15 GB tensor per frame; its backward through MIOpen is what makes the as-written train step take seconds).
 * Upsampling and convolution are linear, so with g9 = f . W (nine per-tap 1x1 products at the encoder's resolution, float32
 * [B,h,w,9,Cmid]) the forward is z = sum_tap shift_tap(upsample(g9_tap)) + bias:
 *   awseg_upconv3x3_linear  z float32 [B,Cmid,H,W] (channels_last = 0) or [B,H,W,Cmid] (1) — no BatchNorm, no activation:
 *                           BatchNorm in training mode needs the batch statistics of z itself; bias float32 [Cmid];
 *                           Cmid % 32 == 0, <= 256;
 *   awseg_upconv3x3_adjoint dg9 = (d z / d g9)^T dz — dz float32 [B,H,W,Cmid] read once, dg9 zeroed and accumulated with
 *                           float atomics; needs H >= 32 h and W >= 32 w (SegFormer: stride 32).
 * The two small GEMMs either side (g9 = f W; df = dg9 W^T, dW = f^T dg9) are the caller's (torch.matmul). */
int awseg_upconv3x3_linear(const float* g9, int64_t batch, int cmid, int h, int w, int height, int width,
                           const float* bias, float* out, int channels_last, awseg_stream_t stream);
int awseg_upconv3x3_adjoint(const float* dz, int64_t batch, int cmid, int h, int w, int height, int width,
                            float* dg9, awseg_stream_t stream);

/* awseg_im2col_nhwc: patch matrix of a strided / patch convolution on a channel-last tensor, so that the convolution
 * is ONE deterministic GEMM (awseg_gemm_split_bias_act / awseg_gemm_bias_act) with bias / folded BatchNorm / activation
 * in its epilogue: x float32 [B,H,W,C] -> cols float32 [B*Ho*Wo, k_padded], cols[m][(ky*kw + kx)*C + c] =
 * x[b][oy*stride - pad + ky*dilation][ox*stride - pad + kx*dilation][c] (zero outside the image and in the
 * k_padded - kh*kw*C tail columns), Ho = (H + 2 pad - dilation (kh-1) - 1) / stride + 1.  C % 4 == 0, k_padded % 4 == 0.
 * Used for the stride-2 3x3 convolutions of ResNet layer2/layer3 (smp encoder behind PKG/models/model.py:349), the MiT
 * patch embeddings and sequence-reduction convolutions (transformers SegformerOverlapPatchEmbeddings /
 * SegformerEfficientSelfAttention.sr inside PKG/models/model.py:193): MIOpen's default solver for those shapes is a
 * split-K implicit GEMM that accumulates with atomics — its result changes from run to run. */
int awseg_im2col_nhwc(const float* x, int64_t batch, int height, int width, int channels, int kernel_h, int kernel_w,
                      int stride, int pad, int dilation, int k_padded, float* cols, awseg_stream_t stream);

/* awseg_attention_d32: O = softmax(Q K^T * scale) V for head_dim 32 in exact float32 on the matrix cores — the
 * self-attention of the MiT encoder (transformers' SegformerEfficientSelfAttention inside the SegformerModel call at
 * PKG/models/model.py:193).  q float32 [batch, n_queries, heads*32], k / v float32 [batch, n_keys, heads*32]
 * (token-major, exactly what the q/k/v Linear layers write), out like q.  n_keys % 32 == 0.  One pass over the
 * keys (online softmax), no score matrix in memory. */
int awseg_attention_d32(const float* q, const float* k, const float* v, float* out, int batch, int heads,
                        int n_queries, int n_keys, float scale, awseg_stream_t stream);

/* awseg_attention_d32_split: the same operator with both GEMMs on the f16 matrix cores and SPLIT float32 operands
 * (x = f16(x) + f16((x - f16(x)) * 2048) / 2048: 22 significant bits, three f16 products per float32 product, float32
 * accumulation) — 16x the MFMA rate per product, so 5.3x per float32-grade product.  Same arguments, layouts and error
 * codes as awseg_attention_d32.  Operand range: any finite float32 — a block that meets |q*scale|, |k| or |v| >= 2^15
 * (outside the split range of f16) redoes its query tile with power-of-two scaled operands and scales the scores / the
 * output back.  Its error against a float64 reference is of the same order as the float32 kernel's
 * (tests/test_gpu_kernels.py). */
int awseg_attention_d32_split(const float* q, const float* k, const float* v, float* out, int batch, int heads,
                              int n_queries, int n_keys, float scale, awseg_stream_t stream);

/* BatchNorm2d (TRAINING: batch statistics) -> ReLU -> Dropout2d on a float32 NCHW map x [batch, channels, hw], fused: the layers
 * behind the first convolution of both SegFormer heads (PKG/models/model.py:152-158, :42-52) inside the training step
 * (PKG/training/trainer.py:299-353).  awseg_bn_train_stats: mean and BIASED variance per channel (float64 sums, fixed fold order).
 * awseg_bn_relu_dropout_forward: out = max((x - mean) invstd gamma + beta, 0) * noise[b, c] (noise float32 [batch, channels]: the
 * Dropout2d channel mask already divided by 1 - p, or NULL).  awseg_bn_relu_dropout_backward: dgamma, dbeta [channels] and dx from
 * x and grad_out — xhat and the ReLU mask are recomputed from x, so autograd keeps x, two per-channel vectors and the noise only.
 * dx_channels_last = 1 stores dx as [batch, hw, channels] memory (channels % 32 == 0): the layout awseg_upconv3x3_adjoint reads.
 * hw % 4 == 0; x, grad_out, out, dx 16-byte aligned; workspace of awseg_bn_train_workspace(batch, channels, hw) bytes. */
int64_t awseg_bn_train_workspace(int batch, int channels, int64_t hw);
int awseg_bn_train_stats(const float* x, int batch, int channels, int64_t hw, void* workspace, float* mean, float* var, awseg_stream_t stream);
int awseg_bn_relu_dropout_forward(const float* x, int batch, int channels, int64_t hw, const float* mean, const float* invstd,
                                  const float* gamma, const float* beta, const float* noise, float* out, awseg_stream_t stream);
int awseg_bn_relu_dropout_backward(const float* x, const float* grad_out, int batch, int channels, int64_t hw, const float* mean,
                                   const float* invstd, const float* gamma, const float* beta, const float* noise, void* workspace,
                                   float* dgamma, float* dbeta, float* dx, int dx_channels_last, awseg_stream_t stream);

/* awseg_dwconv3x3_wgrad_nhwc: weight and bias gradient of a depthwise 3x3 convolution (stride 1, zero padding = dilation) on float32
 * NHWC tensors x, dy [batch, height, width, channels] — the backward of the MiT Mix-FFN's depthwise convolution (transformers'
 * SegformerDepthWiseConv, PKG/models/model.py:120-130) and of the DeepLabV3+ separable convolutions (:259-265) inside
 * AdverseWeatherTrainer.train_epoch's loss.backward() (PKG/training/trainer.py:333).  dw9 float32 [9][channels] tap-major
 * (ky * 3 + kx), db float32 [channels] or NULL.  channels % 4 == 0; workspace of awseg_dwconv3x3_wgrad_workspace(...) bytes
 * (per-chunk partial sums, folded in a fixed order: deterministic).  The forward and the input gradient are awseg_dwconv3x3_nhwc
 * (the latter on dy with the taps flipped). */
int64_t awseg_dwconv3x3_wgrad_workspace(int64_t batch, int height, int width, int channels);
int awseg_dwconv3x3_wgrad_nhwc(const float* x, const float* dy, int64_t batch, int height, int width, int channels, int dilation,
                               void* workspace, float* dw9, float* db, awseg_stream_t stream);

/* awseg_mixffn_fused: one MiT Mix-FFN (transformers' SegformerLayer: hidden = hidden + MixFFN(layer_norm_2(hidden)), MixFFN = dense1
 * -> depthwise 3x3 (padding 1, bias) -> GELU (erf) -> dense2; the encoder PKG/models/model.py:120-130 configures and :193-197 calls)
 * as ONE tile kernel: out[b,y,x,:] = tok + w2 . gelu(dwconv(w1 . layernorm(tok) + b1) + dw_bias) + b2 on float32 NHWC tokens
 * [batch, height, width, channels]; the 4x-wide hidden map stays in LDS.  The two GEMMs run on the f16 matrix cores with SPLIT
 * float32 operands (three f16 products per float32-grade product, float32 accumulation): w1_split uint16 [2][4C][C], w2_split
 * uint16 [2][C][4C] = f16 bit patterns of f16(w) and of f16(w - f16(w)) (nn.Linear layouts [out][in]); w2 float32 [C][4C] as well
 * (a chunk whose GELU outputs reach 2^15 runs its fc2 products on the float32-input MFMA).  The caller guarantees |w| < 2^15 and
 * max|ln_gamma| sqrt(C) + max|ln_beta| < 2^15 (the LayerNorm outputs' bound); dw_taps [9][4C] tap-major (ky * 3 + kx); all
 * vectors 16-byte aligned.  channels 32 or 64 (AWSEG_ERANGE otherwise: the caller keeps its four launches); out must not alias tok. */
int awseg_mixffn_fused(const float* tok, int batch, int height, int width, int channels, const float* ln_gamma, const float* ln_beta,
                       float ln_eps, const uint16_t* w1_split, const float* b1, const float* dw_taps, const float* dw_bias,
                       const uint16_t* w2_split, const float* w2, const float* b2, float* out, awseg_stream_t stream);

/* nn.MaxPool2d(kernel 3, stride 2, padding 1) on a float32 NHWC tensor [batch, height, width, channels] (channels % 4 == 0)
 * -> [batch, (height - 1) / 2 + 1, (width - 1) / 2 + 1, channels]; padding does not take part in the maximum.  The ResNet stem
 * of the smp encoder the reference builds (PKG/models/model.py:262-268); no index tensor is produced. */
int awseg_maxpool3x3s2_nhwc(const float* x, int64_t batch, int height, int width, int channels, float* out,
                            awseg_stream_t stream);
/* The same pooling with the stem's epilogue in its store: out = relu(maxpool(x) + shift[c]) — BatchNorm's folded shift and the
 * ReLU of the smp ResNet stem (conv1 -> bn1 -> relu -> maxpool) commute with the maximum per channel, so they run once per POOLED
 * pixel here instead of as a pass of their own (same two float32 operations as awseg_bias_act_nhwc: bit-identical). */
int awseg_maxpool3x3s2_bias_relu_nhwc(const float* x, int64_t batch, int height, int width, int channels, const float* shift,
                                      float* out, awseg_stream_t stream);

/* Bilinear upsampling of [planes, low_height, low_width] float32 maps to [planes, height, width] with torch's
 * upsample_bilinear2d arithmetic (source index, weights, order of the four products), align_corners 0 or 1.  Replaces
 * the nn.UpsamplingBilinear2d(scale_factor=4) at the end of DeepLabV3+'s segmentation head (the smp model the reference
 * builds at PKG/models/model.py:262-268) and F.interpolate(..., mode="bilinear") calls on logit / depth planes. */
int awseg_upsample_bilinear(const float* low, int64_t planes, int low_height, int low_width,
                            int height, int width, int align_corners, float* out, awseg_stream_t stream);
/* The same upsampling of a low-resolution map that is NOT planar: element (b, c, y, x) at low[b*stride_b + c*stride_c +
 * y*stride_y + x*stride_x] (strides in floats, >= 0) — e.g. the NHWC rows [B*h*w, C] the 19-class head's GEMM wrote
 * (stride_b = h*w*C, stride_c = 1, stride_y = w*C, stride_x = C), so that no NCHW copy of them is made.  out is planar
 * [batch, channels, height, width].  Upsampling by more than 3 in both directions with width % 4 == 0 takes any strides; other
 * scales take planar maps only (AWSEG_ERANGE otherwise).  Same arithmetic, same values. */
int awseg_upsample_bilinear_strided(const float* low, int64_t batch, int channels, int low_height, int low_width,
                                    int64_t stride_b, int64_t stride_c, int64_t stride_y, int64_t stride_x,
                                    int height, int width, int align_corners, float* out, awseg_stream_t stream);

/* out[r] = sigmoid(x[r, :] . w + bias[0]) (sigmoid = 0: the plain sum) for x float32 [rows, k], k % 4 == 0: the 1x1 convolution to
 * one channel + nn.Sigmoid that ends DepthEstimationHead (PKG/models/model.py:49-51) on NHWC rows, for hidden widths the fused
 * Winograd epilogue (64) does not take — the DeepLab member's stride-16 depth map (model.py:368).  float32 FMAs, 1 / (1 + expf(-v)). */
int awseg_rowdot_sigmoid(const float* x, int64_t rows, int k, const float* w, const float* bias, int sigmoid, float* out,
                         awseg_stream_t stream);

/* smp's ASPPPooling branch behind its global mean, and that branch's slice of the ASPP projection (the model built at
 * PKG/models/model.py:262-268), for one row per image: out[b, :] = relu(mean[b, :] w1^T + b1) w2^T + b2 with mean [batch, cin],
 * w1 [cmid, cin] and b1 [cmid] (BatchNorm folded), w2 [cout, cmid], b2 [cout] or NULL.  Two small kernels (a block per four
 * channels each) with the hidden row in workspace = awseg_aspp_pool_branch_workspace(batch, cmid) bytes. */
int64_t awseg_aspp_pool_branch_workspace(int batch, int cmid);
int awseg_aspp_pool_branch(const float* mean, int batch, int cin, const float* w1, const float* b1, int cmid, const float* w2,
                           const float* b2, int cout, void* workspace, float* out, awseg_stream_t stream);

/* The frames as both 7x7 stems read them: planar float32 [batch, channels <= 4, height, width] (element (b, c, y, x) at
 * x[b*stride_b + c*stride_c + y*stride_y + x], strides in floats) into columns 3 .. 3 + width - 1 of a 4-channel NHWC image
 * [batch, height, padded_width, 4] whose other columns / channels the caller zeroed once (the operand of
 * awseg_conv_rows_gemm_split_bias_act: ResNet conv1 and MiT's first patch embedding behind PKG/models/model.py:186-200, :349).
 * One pass, 16-byte accesses on both sides. */
int awseg_stem_image(const float* x, int batch, int channels, int height, int width, int64_t stride_b, int64_t stride_c,
                     int64_t stride_y, float* image, int padded_width, awseg_stream_t stream);

/* awseg_depth_upsample_combine: the depth tail of the ensemble in one pass — d2_full = bilinear upsample
 * (align_corners=False) of the stride-16 DeepLab depth map d2_low [B,h,w] to [B,H,W] (PKG/models/model.py:368-371) and
 * d_out = weights[0]*d1 + weights[1]*d2_full, or (d1 + d2_full)/2 when weights is NULL (model.py:471-478).
 * weights: device float32 [2] (softmax of the ensemble weights).  d1, d2_full, d_out float32 [B,H,W]. */
int awseg_depth_upsample_combine(const float* d1, const float* d2_low, int batch, int low_height, int low_width,
                                 int height, int width, const float* weights, float* d2_full, float* d_out,
                                 awseg_stream_t stream);

/* awseg_bias_act_nhwc: x = act(x + bias[c] (+ residual)) in place on float32 [n_pixels, C]:
 * the epilogue of a convolution whose eval-mode BatchNorm scale was folded into its weights
 * (Conv -> BN -> [+identity] -> ReLU of the ResNet bottlenecks behind PKG/models/model.py:349). */
int awseg_bias_act_nhwc(float* x, int64_t n_pixels, int channels, const float* bias, const float* residual,
                        int act, awseg_stream_t stream);

/* awseg_layernorm_rows: LayerNorm over the last dimension of float32 [n_rows, C] (C % 4 == 0,
 * C <= 1024): out = (x - mean) * rsqrt(var + eps) * gamma + beta, biased variance, as
 * torch.nn.LayerNorm.  Replaces the nn.LayerNorm calls inside the SegFormer encoder that
 * PKG/models/model.py:193 runs (32..512 channels per token: far below torch's kernel sweet spot). */
int awseg_layernorm_rows(const float* x, int64_t n_rows, int channels, const float* gamma, const float* beta,
                         float eps, float* out, awseg_stream_t stream);

/* ------------------------------------------------------------------------- *
 *  next #1  ConfidenceCalibration.compute_ece accumulators
 *       replaces PKG/evaluation/metrics.py:161-194 (the per-pixel part)
 * ------------------------------------------------------------------------- *
 * For every pixel with label != 255: conf = max softmax prob, bin k with
 * edges[k] < conf <= edges[k+1] (edges: device float32[n_bins+1] =
 * torch.linspace(0,1,n_bins+1), n_bins <= 64); bins[slot][k] accumulates
 * {int64 count, int64 sum_conf in units of 2^-30 (fixed point: exact and independent of summation order), int64
 * sum_correct} (24 bytes per bin).
 * Slots as in awseg_combine_argmax_confusion.  workspace as awseg_metrics_workspace.
 */
int awseg_ece_accumulate(const float* logits, int64_t batch, int num_classes, int64_t hw,
                         const void* label, int label_dtype, const int32_t* cond,
                         const float* edges, int n_bins,
                         void* bins, int n_slots, void* workspace, awseg_stream_t stream);

/* Ensemble calibration + disagreement statistics in one pass over the two member logit maps
 * (replaces REF/scripts/evaluate.py:230-255 = ConfidenceCalibration.compute_ece on the ensemble logits,
 * PKG/evaluation/metrics.py:161-194, and the per-pixel part of
 * EnsembleDisagreementMetrics.compute_disagreement_auroc, :353-367, :414-426).  ECE bins as in
 * awseg_ece_accumulate, computed from r = combine(seg1, seg2)/T (mode WEIGHTED or MEAN).  The
 * disagreement score (mutual information) of every pixel with label != 255 is counted into
 * auroc_hist int64 [2][n_hist]: row 0 = correctly predicted (argmax of the mean probability == label),
 * row 1 = errors; bin = (score - hist_lo) * n_hist / (hist_hi - hist_lo), clamped; n_hist <= 8192 (the
 * histogram is aggregated per block in LDS before it touches global memory).  C must be 19,
 * H*W a multiple of 4.  workspace as awseg_metrics_workspace. */
int awseg_ensemble_eval_stats(const float* seg1, const float* seg2, int64_t batch, int num_classes, int64_t hw,
                              int mode, const float* weights, const float* temperature,
                              const void* label, int label_dtype, const int32_t* cond,
                              const float* edges, int n_bins, void* ece_bins, int n_slots,
                              int64_t* auroc_hist, int n_hist, float hist_lo, float hist_hi,
                              void* workspace, awseg_stream_t stream);

/* awseg_combine_argmax_confusion (no logits / prediction output) and awseg_ensemble_eval_stats in ONE pass over the two member
 * logit maps: the 19 x 19 confusion counts of argmax(combine(s1, s2) / T) against the labels (slots: overall + 1 + cond[b], the
 * combine kernel's argmax / ignore_index / uint8-wrap / out-of-range rules) together with the ECE bins and the disagreement
 * histogram.  Same results as the two calls; the second 2 x 19 x 4 B/px read is gone.  C = 19, hw % 4 == 0; workspace of
 * awseg_metrics_workspace(batch, 19, hw) bytes.  Replaces REF/scripts/evaluate.py:179-200 + :230-255 per batch. */
int awseg_combine_confusion_stats(const float* seg1, const float* seg2, int64_t batch, int num_classes, int64_t hw,
                                  int mode, const float* weights, const float* temperature, const void* label,
                                  int label_dtype, int ignore_index, int label_wrap_u8, const int32_t* cond,
                                  int64_t* counts, int count_slots, int64_t* oob, const float* edges, int n_bins,
                                  void* ece_bins, int ece_slots, int64_t* auroc_hist, int n_hist, float hist_lo,
                                  float hist_hi, void* workspace, awseg_stream_t stream);

#ifdef __cplusplus
}
#endif
#endif /* AWSEG_H */
