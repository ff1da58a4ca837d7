#!/usr/bin/env python3
"""bench.py — images/s of the north-star hot path on MI355X.

    python bench.py --gpus N --steps K --warmup W          # N > 1: starts its own N rank processes
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 \
           --master-port P bench.py --gpus N --steps K --warmup W      # or under a launcher (env RANK/WORLD_SIZE)

One "step" = one pass of the hot path over one batch of B synthetic 1024x2048 frames that are
already resident in HBM as uint8 HWC (+ uint8 labels):

    weather corruption (condition = global frame index mod 5, in-kernel Philox noise)
      -> fused Normalize/ToTensor -> SegFormer-B0 + DeepLabV3+-R50 ensemble forward (float32 results)
      -> combine / temperature / argmax / 19x19 confusion (overall + per condition) in one pass
      -> ECE bins + ensemble-disagreement histogram of the same logits (REF/scripts/evaluate.py:230-255)

The samples are a FIXED GLOBAL SET of 160 frames (C2/C3 N=160 = 8 GPUs x 5 conditions x 4):
frame g — pixels, labels, weather condition, every random draw — is a function of (seed, g) only.  Rank r
owns the contiguous block parallel.shard_range(160, r, N) and keeps it resident in HBM.
  * timed region (weak scaling): every rank runs K steps of B frames, cycling through its block; after the
    K steps the int64 counters are SUM-all-reduced over ranks (RCCL) and the mIoUs finished on the host,
    inside the timed region.  value = N*B*K / max-over-ranks time.
  * parity pass (untimed): every global frame is evaluated exactly ONCE by its owner, counters all-reduced:
    the reported `miou` dict is therefore bit-identical at any N (integer sums are order-independent).
  * per-kernel pass (untimed): a few more steps with HIP event pairs around every C-ABI launch on the
    launching stream -> `roofline` (dominant hand-written kernel) and `kernels`.
  * fp32 pass (N=1): a few steps with every split-operand f16-MFMA kernel replaced by its float32-input
    counterpart -> `fp32_mfma`.
`cpu_baseline` = the CPU oracle path timed on this box's host cores on a bounded sample (rank 0, N=1 only).
"""
from __future__ import annotations

import argparse
import json
import os
import socket
import subprocess
import sys
import time
from pathlib import Path

ROOT = Path(__file__).resolve().parent
if str(ROOT) not in sys.path:
    sys.path.insert(0, str(ROOT))

np = None       # numpy / torch are imported by main() AFTER the self-launch decision: the parent of a
torch = None    # self-launched multi-rank run never loads the GPU runtime

HBM_PEAK_GBS = 8000.0          # MI355X_MICROARCH.md: 8 TB/s spec (6.29 TB/s measured float4 copy)
MFMA_F32_PEAK_TFLOPS = 157.3   # MI355X_MICROARCH.md: fp32-input MFMA dense peak
MFMA_F16_PEAK_TFLOPS = 2516.6  # MI355X_MICROARCH.md: f16 / bf16 dense peak (256 CUs x 4 SIMDs x 1024 FLOP/cycle x 2.4 GHz)


class KernelClock:
    """HIP event pairs recorded immediately around each C-ABI launch, on the torch current stream
    (the stream the launchers enqueue on).  Read after the timed region has been synchronised."""

    def __init__(self):
        self.enabled = False
        self.pairs = {}

    def hook(self, name, thunk, args=()):
        if not self.enabled:
            return thunk()
        s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        s.record()
        rc = thunk()
        e.record()
        w = launch_work(name, args)
        # a launcher whose launches fall on both sides of the machine balance is listed as two rows (launch_work names the second)
        self.pairs.setdefault(w[2] if (w is not None and len(w) > 2) else name, []).append((s, e, None if w is None else w[:2]))
        return rc

    def summary(self):
        """name -> (launches, mean ms per launch, (bound, mean per-launch work) or None)."""
        out = {}
        for n, p in self.pairs.items():
            works = [w for _, _, w in p]
            own = None if works[0] is None else (works[0][0], sum(w[1] for w in works) / len(works))
            out[n] = (len(p), sum(s.elapsed_time(e) for s, e, _ in p) / len(p), own)
        return out


def launch_work(name, args):
    """(bound, work) of ONE launch when it depends on the launch's own shape arguments (kernels called at several
    shapes per step); None -> algorithmic_work() prices the launch from the bench shape."""
    if name == "awseg_conv3x3_winograd_nhwc":
        # (x, batch, H, W, Cin, Cout, dilation, ...): MFMA flops issued = 16 multiplies per 2x2 output tile,
        # 2.25x fewer than the direct convolution computes
        _, b, h, w, cin, cout = args[:6]
        return "mfma", 2.0 * 16 * cin * cout * b * ((h + 1) // 2) * ((w + 1) // 2)
    if name == "awseg_conv3x3_winograd_split_nhwc":
        # the same 16 multiplies per 2x2 output tile, each issued as THREE f16 MFMA products: priced as issued, against the f16 peak
        _, b, h, w, cin, cout = args[:6]
        return "mfma_f16", 3 * 2.0 * 16 * cin * cout * b * ((h + 1) // 2) * ((w + 1) // 2)
    if name == "awseg_depth_head_fused":
        # (forms, batch, h, w, cmid, u, u_is_bf16, ...): the Winograd products of the second 3x3 (cmid -> 64) at 32h x 32w, as issued
        # (three f16 products per multiply; one in bf16 mode); the generated hidden map costs vector instructions, no matrix flops
        _, b, h, w, cmid = args[:5]
        return "mfma_f16", (1 if args[6] else 3) * 2.0 * 16 * cmid * 64 * b * (16 * h) * (16 * w)
    if name == "awseg_mixffn_fused":
        # (tok, batch, H, W, C, ...): the two GEMMs (C -> 4C on the haloed 8 x 32 tile, 4C -> C) as issued, three f16 products per multiply;
        # algorithmic bytes would be tok in + out (8 C bytes a token) — the kernel is bound by the GELU's vector work, priced here against the matrix peak
        _, b, h, w, c = args[:5]
        return "mfma_f16", 3 * 2.0 * b * h * w * (4 * c * c) * (256.0 / 180.0 + 1.0)
    if name == "awseg_upconv_forms":
        # (g9, batch, cmid, h, w, shift, forms): g9 in, the two tables out
        _, b, cmid, h, w = args[:5]
        return "hbm", 4.0 * b * (h * w * 9 * cmid + (3 * h + 3) * (w + 1) * 4 * cmid + (3 * h + 3) * (2 * w + 2) * 2 * cmid)
    if name == "awseg_conv3x3_winograd_bf16_nhwc":
        _, b, h, w, cin, cout = args[:6]
        return "mfma_f16", 2.0 * 16 * cin * cout * b * ((h + 1) // 2) * ((w + 1) // 2)      # one bf16 product per multiply
    if name == "awseg_gemm_bf16_bias_act":
        m, n, k = args[6:9]
        flops = 2.0 * m * n * k
        nbytes = 4.0 * (m * k + m * n * (2 if args[3] is not None else 1)) + 2.0 * n * k     # float32 activations in memory, bf16 weights
        if nbytes / (HBM_PEAK_GBS * 1e9) > flops / (MFMA_F16_PEAK_TFLOPS * 1e12):             # as awseg_gemm_split_bias_act below
            return "hbm", nbytes, name + " [launches bound by HBM]"
        return "mfma_f16", flops
    if name == "awseg_attention_d32_bf16":
        b, heads, nq, nkv = args[4:8]
        return "mfma_f16", 4.0 * b * heads * nq * nkv * 32
    if name == "awseg_attention_d32":
        # (q, k, v, out, batch, heads, n_queries, n_keys, ...): QK^T and PV, head_dim 32
        b, heads, nq, nkv = args[4:8]
        return "mfma", 4.0 * b * heads * nq * nkv * 32
    if name == "awseg_attention_d32_split":
        # the same products, each issued as THREE f16 MFMA products (split operands): priced as issued, against the f16 peak
        b, heads, nq, nkv = args[4:8]
        return "mfma_f16", 3 * 4.0 * b * heads * nq * nkv * 32
    if name == "awseg_attention_d32_split_ws":
        # (q, k, v, kv_pitch, out, batch, heads, n_queries, n_keys, ...): the split-operand products (the preparing kernel's pass over the
        # keys / values is noise beside them)
        b, heads, nq, nkv = args[5:9]
        return "mfma_f16", 3 * 4.0 * b * heads * nq * nkv * 32
    if name == "awseg_attention_d32_packed_kv":
        # (q, kv, out, batch, heads, n_queries, n_keys, scale, mode, ...)
        b, heads, nq, nkv = args[3:7]
        return ("mfma", 4.0 * b * heads * nq * nkv * 32) if args[8] == 0 else ("mfma_f16", (3 if args[8] == 1 else 1) * 4.0 * b * heads * nq * nkv * 32)
    if name == "awseg_gemm_split_dual_bias_act":
        # (x, k1, x2, k2, batch, H2, W2, stride, w_split, bias, residual, act, out, m, n): one product over [x | x2]; bytes: both
        # operand pieces once (the strided piece: the rows it gathers), the output once
        k1, k2 = args[1], args[3]
        m, n = args[13], args[14]
        flops = 3 * 2.0 * m * n * (k1 + k2)
        nbytes = 4.0 * (m * (k1 + k2) + m * n + n * (k1 + k2))
        if nbytes / (HBM_PEAK_GBS * 1e9) > flops / (MFMA_F16_PEAK_TFLOPS * 1e12):
            return "hbm", nbytes, name + " [launches bound by HBM]"
        return "mfma_f16", flops
    if name == "awseg_gemm_split_pieces_bias_act":
        # (pieces, n_pieces, k_piece, w_split, bias, residual, act, out, m, n): one product over the pieces; each piece, the residual and the output once
        npc, kp = args[1], args[2]
        m, n = args[8], args[9]
        flops = 3 * 2.0 * m * n * npc * kp
        nbytes = 4.0 * (m * npc * kp + m * n * (2 if args[5] is not None else 1) + n * npc * kp)
        if nbytes / (HBM_PEAK_GBS * 1e9) > flops / (MFMA_F16_PEAK_TFLOPS * 1e12):
            return "hbm", nbytes, name + " [launches bound by HBM]"
        return "mfma_f16", flops
    if name == "awseg_stem_image":
        # (x, batch, channels, H, W, ...): the planar frames in, the 4-channel image's interior out
        _, b, c, h, w = args[:5]
        return "hbm", 4.0 * b * h * w * (c + 4)
    if name == "awseg_aspp_depthwise3_mean":
        # (x, batch, h, w, C, ...): x once, the three depthwise maps out
        _, b, h, w, c = args[:5]
        return "hbm", 16.0 * b * h * w * c
    if name == "awseg_rowdot_sigmoid":
        return "hbm", 4.0 * args[1] * (args[2] + 1)        # (x, rows, k, ...)
    if name == "awseg_upsample_bilinear_strided":
        # (low, batch, channels, h, w, strides x4, H, W, ...): the small map in, the full-resolution planes out
        _, b, c, h, w = args[:5]
        return "hbm", 4.0 * b * c * (h * w + args[9] * args[10])
    if name == "awseg_gemm_split_bias_act":
        # (x, w_split, bias, residual, act, out, m, n, k): three f16 MFMA products per float32-grade product, priced as issued —
        # where the matrix pipe is the roofline.  A launch whose algorithmic bytes (x once, the output once, the residual once) take
        # longer at the HBM peak than its issued products at the MFMA peak is priced in bytes, in a row of its own: the ResNet
        # layer1 / layer2 1x1s, the decoder and the MiT projections are such launches (K <= 512 on 10^5 - 10^6 rows)
        m, n, k = args[6:9]
        flops = 3 * 2.0 * m * n * k
        nbytes = 4.0 * (m * k + m * n * (2 if args[3] is not None else 1) + n * k)
        if nbytes / (HBM_PEAK_GBS * 1e9) > flops / (MFMA_F16_PEAK_TFLOPS * 1e12):
            return "hbm", nbytes, name + " [launches bound by HBM]"
        return "mfma_f16", flops
    if name == "awseg_conv_gemm_split_bias_act":
        # (x, batch, H, W, C, kh, kw, stride, pad, dil, w_split, bias, residual, act, out, n): the same GEMM, A operand gathered
        _, b, h, w, c, kh, kw, st, pd, dl = args[:10]
        ho, wo = (h + 2 * pd - dl * (kh - 1) - 1) // st + 1, (w + 2 * pd - dl * (kw - 1) - 1) // st + 1
        return "mfma_f16", 3 * 2.0 * b * ho * wo * args[15] * kh * kw * c
    if name == "awseg_conv_rows_gemm_split_bias_act":
        # (x, batch, H, Wp, pixel_floats, kh, stride, pad_y, out_w, w_split, bias, residual, act, out, n): K = kh * 32 as issued
        # (35 % of it multiplies zero weights: kx = 7, c = 3)
        _, b, h, _, _, kh, st, pdy, wo = args[:9]
        ho = (h + 2 * pdy - kh) // st + 1
        return "mfma_f16", 3 * 2.0 * b * ho * wo * args[14] * kh * 32
    if name == "awseg_dwconv3x3_wgrad_nhwc":
        # (x, dy, batch, H, W, C, ...): the two maps read once; the partial sums are noise beside them
        _, _, b, h, w, c = args[:6]
        return "hbm", 8.0 * b * h * w * c
    if name == "awseg_dwconv3x3_nhwc":
        # (x, batch, H, W, C, ...): read + write of the activation
        _, b, h, w, c = args[:5]
        return "hbm", 8.0 * b * h * w * c
    if name == "awseg_dwconv3x3_upcat_nhwc":
        # (a, h, w, Ca, hi, Ch, batch, H, W, ...): read the two inputs once, write the concatenated map
        _, h, w, ca, _, ch, b, H, W = args[:9]
        return "hbm", 4.0 * b * (h * w * ca + H * W * ch + H * W * (ca + ch))
    if name == "awseg_bias_act_nhwc":
        # (x, n_pixels, C, bias, residual, act): in-place pass (+ residual read)
        _, npx, c, _, res = args[:5]
        return "hbm", 4.0 * npx * c * (3 if res is not None else 2)
    if name == "awseg_layernorm_rows":
        return "hbm", 8.0 * args[1] * args[2]            # (x, n_rows, C, ...)
    # ---- training step (BASELINE configs[3]) ----
    if name == "awseg_upconv3x3_adjoint":
        # (dz, batch, cmid, h, w, height, width, dg9): ONE read of the full-resolution gradient map; the low-resolution output is noise beside it
        _, b, cmid, _, _, hh, ww = args[:7]
        return "hbm", 4.0 * b * cmid * hh * ww
    if name == "awseg_upconv3x3_linear":
        # (g9, batch, cmid, h, w, height, width, ...): the write of the full-resolution pre-activation map
        _, b, cmid, _, _, hh, ww = args[:7]
        return "hbm", 4.0 * b * cmid * hh * ww
    if name == "awseg_fog_ce_forward":
        # (logits, label, label_dtype, density, batch, C, hw, ...): logits + label + density in
        b, c, hw = args[4:7]
        return "hbm", float(b) * hw * (4 * c + 1 + 4)
    if name == "awseg_fog_ce_backward":
        b, c, hw = args[4:7]
        return "hbm", float(b) * hw * (8 * c + 1 + 4)
    return None


CLOCK = KernelClock()


def algorithmic_work(name, B, H, W, C, info):
    """Algorithmic bytes (or flops) of one launch: SURVEY §8(d) per-pixel figure x pixels per launch
    (DESIGN.md §4 lists them).  `info` carries how many frames of the batch each launch covered."""
    px = H * W
    n = info.get(name, B)
    if name == "awseg_fog_fused":
        return "hbm", (3 + 12) * px * n                # u8 in, fused f32 CHW normalised out; Philox noise: 0 B
    if name in ("awseg_night_apply", "awseg_rain_apply", "awseg_snow_apply", "awseg_normalize"):
        return "hbm", (3 + 12) * px * n
    if name == "awseg_weather_batch":                  # every frame of the batch in one launch (7x7-snow frames keep awseg_snow_apply): u8 in, f32 CHW out
        return "hbm", (3 + 12) * px * n
    if name == "awseg_combine_argmax_confusion":
        return "hbm", (2 * C * 4 + 1) * px * B         # two member logit maps + labels in; counters only out
    if name == "awseg_combine_confusion_stats":
        return "hbm", (2 * C * 4 + 1) * px * B         # ONE pass over the two member logit maps + labels: confusion, ECE bins, disagreement histogram
    if name == "awseg_ensemble_eval_stats":
        return "hbm", (2 * C * 4 + 1) * px * B         # the same two member logit maps + labels again; bins / histogram out
    if name == "awseg_depth_upsample_combine":
        return "hbm", (4 + 4 + 4) * px * B             # segformer depth in, upsampled deeplab depth + combined depth out
    if name == "awseg_aspp_depthwise3":
        h, w = H // 16, W // 16
        return "hbm", (2048 * 4 + 3 * 2048 * 4) * h * w * B
    if name == "awseg_segformer_head_fused":           # executed MFMA flops: GEMM1 K=12 + GEMM2 N padded to 32
        return "mfma", 2.0 * (12 * 256 + 256 * 32) * px * B
    if name == "awseg_segformer_head_fused_split":     # issued f16 MFMA flops: GEMM1 K padded to 16, 2 products; GEMM2 3 products
        return "mfma_f16", 2.0 * (2 * 16 * 256 + 3 * 256 * 32) * px * B
    if name == "awseg_upconv3x3_bn_relu":              # 24 MFMAs per 32 px, but 512 B/px of output: the write is the roofline
        return "hbm", 128 * 4 * px * B
    return "hbm", 0


# device-function names of the C-ABI launchers' dominant kernels (for the PMC traffic lookup)
DEVICE_KERNEL = {"awseg_conv3x3_winograd_nhwc": "conv3x3_wino_kernel<1>",
                 "awseg_conv_rows_gemm_split_bias_act": ("gemm_split3_kernel<true, 0, false, 2, 4", "gemm_split3_kernel<true, 0, false, 2, 8"),
                 "awseg_conv3x3_winograd_split_nhwc": ("wino8p_kernel<0, false>", "wino8p_kernel<1, false>", "wino8s_kernel<0, false>", "wino8s_kernel<1, false>", "wino8_kernel<0, false>", "wino8_kernel<1, false>", "wino_split_kernel<0, false>", "wino_split_kernel<1, false>"),
                 "awseg_conv3x3_winograd_bf16_nhwc": ("wino8p_kernel<0, true>", "wino8p_kernel<1, true>", "wino8s_kernel<0, true>", "wino8s_kernel<1, true>", "wino8_kernel<0, true>", "wino8_kernel<1, true>", "wino_split_kernel<0, true>", "wino_split_kernel<1, true>"),
                 "awseg_gemm_split_bias_act": ("gemm_split3_kernel<false, 0, false", "gemm_split3_kernel<true, 0, false", "gemm_split_kernel<4, 2, 2, 4, false, false", "gemm_split_kernel<4, 2, 2, 4, true, false", "gemm_split_kernel<2, 2, 2, 4, false, false", "gemm_split_kernel<1, 2, 4, 2, false, false", "gemm_split_kernel<2, 2, 2, 4, true, false", "gemm_split_kernel<1, 2, 4, 2, true, false", "gemm_split_kernel<2, 2, 4, 2, false, false, true", "gemm_split_kernel<2, 2, 4, 2, true, false, true"),
                 "awseg_gemm_bf16_bias_act": ("gemm_split3_kernel<false, 0, true", "gemm_split_kernel<2, 2, 2, 4, false, true", "gemm_split_kernel<1, 2, 4, 2, false, true"),
                 "awseg_attention_d32_split": "attention_d32_split_kernel", "awseg_attention_d32_split_ws": "attention_d32_split_img_kernel",
                 "awseg_attention_d32_packed_kv": "attention_d32_split_kernel", "awseg_gemm_split_dual_bias_act": ("gemm_split3_kernel<false, 0, false, 8, 8, true", "gemm_split3_kernel<false, 0, false, 4, 4, true", "gemm_split3_kernel<false, 0, false, 4, 8, true", "gemm_split3_kernel<false, 0, false, 2, 4, true", "gemm_split3_kernel<false, 0, false, 2, 8, true"),
                 "awseg_gemm_split_pieces_bias_act": ("gemm_split3_kernel<false, 0, false, 8, 8, true", "gemm_split3_kernel<false, 0, false, 4, 4, true"),
                 "awseg_aspp_depthwise3_mean": "aspp_dw3_lds_kernel", "awseg_stem_image": "stem_image_kernel",
                 "awseg_depth_head_fused": ("wino8p_kernel<2, false>", "wino8p_kernel<2, true>"),
                 "awseg_upconv_forms": "upconv_forms_kernel", "awseg_weather_batch": "weather_batch_kernel", "awseg_mixffn_fused": ("mixffn_kernel<32>", "mixffn_kernel<64>"),
                 "awseg_segformer_head_fused": "head_mfma_classify_kernel<8>", "awseg_segformer_head_fused_split": "head_split_classify_kernel<8>", "awseg_combine_argmax_confusion": "combine_argmax_confusion_kernel<0", "awseg_combine_confusion_stats": "ensemble_stats_kernel<",
                 "awseg_upconv3x3_adjoint": "upconv3x3_adjoint_kernel", "awseg_upconv3x3_linear": "head_mfma_kernel<", "awseg_dwconv3x3_wgrad_nhwc": "dwconv3x3_wgrad_partial_kernel",
                 "awseg_dwconv3x3_nhwc": ("dwconv3x3_nhwc_strip2_kernel", "dwconv3x3_nhwc_strip_kernel", "dwconv3x3_nhwc_kernel"),
                 "awseg_fog_ce_forward": "fog_ce_forward", "awseg_fog_ce_backward": "fog_ce_backward",
                 "awseg_upconv3x3_bn_relu": "head_mfma_kernel<4, false", "awseg_aspp_depthwise3": ("aspp_dw3_lds_kernel", "aspp_dw3_rows_kernel", "aspp_dw3_walk_kernel")}


TRAFFIC_TABLES = ["r04_bench_step_stats_and_traffic.csv", "r03_bench_step_stats_and_traffic.csv", "r02_bench_step_stats_and_traffic.csv", "r02_kernel_bench_stats_and_traffic.csv",
                  "r01_kernel_bench_v4_stats_and_traffic.csv"]   # first match wins


B5_TRAFFIC_TABLE = "r04_bench_b5_step_stats_and_traffic.csv"     # the same passes of `bench.py --model b5_r101`


TRAIN_TRAFFIC_TABLE = "r04_train_step_stats_and_traffic.csv"     # the same passes of `bench.py --mode train`


def pmc_traffic(name, b5=False, train=False):
    """(HBM bytes per launch of `name`, source) from the committed PMC passes (profiles/: separate rocprofv3
    --pmc FETCH_SIZE and --pmc WRITE_SIZE runs; FETCH_SIZE doubled as MI355X_MICROARCH.md §HBM prescribes for
    wide coalesced reads).  The first table is from THIS command (bench.py, mean over the launches of a
    step); the older ones are tools/kernel_bench.py at the same problem size.  A launcher whose work is split
    over several device-function instances is the call-weighted mean of their rows.  (None, None) if no
    committed profile has the kernel — bench.py itself does not collect counters."""
    import csv
    keys = DEVICE_KERNEL.get(name) or DEVICE_KERNEL.get(name.split(" [")[0])      # "<launcher> [launches bound by HBM]": the launcher's kernels
    if not keys:
        return None, None
    if isinstance(keys, str):
        keys = (keys,)
    for fname in ([TRAIN_TRAFFIC_TABLE] if train else ([B5_TRAFFIC_TABLE] if b5 else []) + TRAFFIC_TABLES):
        path = ROOT / "profiles" / fname
        if not path.exists():
            continue
        with open(path) as f:
            rows = list(csv.DictReader(l for l in f if not l.startswith("#")))
        hit = [r for r in rows if any(k in r["kernel"] for k in keys) and r["FETCH_SIZE_KB"] and r["WRITE_SIZE_KB"]]
        if hit:
            calls = sum(float(r["calls"]) for r in hit)
            kb = sum(float(r["calls"]) * (2.0 * float(r["FETCH_SIZE_KB"]) + float(r["WRITE_SIZE_KB"])) for r in hit) / calls
            what = ("this command under rocprofv3, mean over a step's launches" if ("bench_step" in fname or fname in (B5_TRAFFIC_TABLE, TRAIN_TRAFFIC_TABLE))
                    else "tools/kernel_bench.py, median per dispatch of its shapes")
            return int(kb * 1024), f"profiles/{fname} (separate --pmc FETCH_SIZE / WRITE_SIZE passes; {what})"
    return None, None


# the reference's OWN Python functions, timed once in the 8-core survey container (the reference
# cannot travel to the GPU box, so these are quoted, not re-measured there).  Seconds per 1024x2048 frame unless the key says otherwise.
REFERENCE_CODE_TIMINGS = {
    "_apply_fog_s_per_frame": 1.67, "_apply_night_s_per_frame": 0.33, "compute_iou_s_per_frame": 0.11, "argmax_s_per_frame": 0.34,
    "compute_ece_s_per_frame": 0.70, "segformer_b0_branch_256x512_s": 1.9, "_apply_rain_snow": None, "deeplabv3plus_branch": None,
    "cores": 8, "where": "survey container (8 CPU cores, torch 2.10 CPU, numpy 2.2.6, scipy 1.15.3), single measurements",
    "note": "Reference code itself, 8-core survey container (BASELINE.md §2, quoted): _apply_fog 1.67 s/frame, _apply_night 0.33 s/frame, "
            "compute_iou 0.11 s, logits.argmax 0.34 s, compute_ece 0.70 s per 1024x2048 frame; SegFormer-B0 branch ~1.9 s at 256x512; "
            "rain / snow (cv2) and the DeepLabV3+ branch (smp) not measurable there"}


def cpu_baseline(model, H, W, C, seed=0, fwd_div=1, max_threads=16):
    """The CPU oracle path ("port") on this box's host cores, on a bounded sample: one full-size
    frame per weather condition through the C oracle transforms + normalise, oracle argmax +
    confusion on one full-size logit map, and the as-written torch-CPU ensemble forward on ONE frame
    of (H/fwd_div)x(W/fwd_div), time scaled by fwd_div^2 when fwd_div > 1 (convolution cost is
    linear in pixels).  A reported baseline, not the optimisation target."""
    import copy
    from oracle import cpu_oracle as O
    O.build()
    # a one-GPU box owns a 16-CPU share of the host (more torch threads than that only thrash:
    # 256 threads ran the same forward 300x slower than 16)
    try:
        avail = len(os.sched_getaffinity(0))
    except AttributeError:
        avail = os.cpu_count() or 1
    cores = max(1, min(avail, max_threads))
    torch.set_num_threads(cores)
    rs = np.random.RandomState(seed)
    img = rs.randint(0, 255, (H, W, 3), dtype=np.uint8)
    lab = rs.randint(0, C, (H, W)).astype(np.uint8)
    np.random.seed(42)
    t_weather = []
    for cond in ("clean", "fog", "rain", "snow", "night"):
        t0 = time.perf_counter()
        O.normalize(O.apply_weather_effect(img, cond))
        t_weather.append(time.perf_counter() - t0)
    cpu_model = copy.deepcopy(model).cpu().eval()
    for m in cpu_model.modules():
        m.fused_eval = False
    hs, ws = H // fwd_div, W // fwd_div
    x = torch.randn(1, 3, hs, ws)
    with torch.no_grad():
        cpu_model(x[:, :, :64, :64])                       # warm-up (thread pool, allocator)
        t0 = time.perf_counter()
        cpu_model(x)
        t_fwd = (time.perf_counter() - t0) * fwd_div * fwd_div
    logits = rs.randn(1, C, H, W).astype(np.float32)
    t0 = time.perf_counter()
    O.confusion(O.argmax(logits), lab[None], C)
    t_metric = time.perf_counter() - t0
    per_image = float(np.mean(t_weather)) + t_fwd + t_metric
    return {"value": round(1.0 / per_image, 5), "unit": "images/s", "cores": cores, "kind": "port",
            "sample": f"C oracle transforms+normalise on 5 frames {H}x{W}, one per condition (mean {np.mean(t_weather):.2f} s/frame, 1 thread); "
                      f"as-written torch-CPU ensemble forward on 1 frame {hs}x{ws} x{fwd_div * fwd_div} = {t_fwd:.1f} s/frame "
                      f"({cores} torch threads); oracle argmax+confusion on 1 frame {H}x{W} ({t_metric:.2f} s, 1 thread).  "
                      + REFERENCE_CODE_TIMINGS["note"],
            "reference_code_timings": REFERENCE_CODE_TIMINGS}


def cpu_train_baseline(model, H, W, C, div=8, max_threads=16):
    """One optimisation step of the AS-WRITTEN graph on the host cores (torch-CPU autograd: F.interpolate -> Conv2d heads,
    module ASPP, cross_entropy * (1 + 2 density) + 0.1 MSE depth, clip, AdamW) on a batch of 2 frames of (H/div)x(W/div),
    time scaled by div^2 (convolution cost is linear in pixels).  A reported baseline, not the optimisation target."""
    import copy
    import torch.nn.functional as F
    try:
        avail = len(os.sched_getaffinity(0))
    except AttributeError:
        avail = os.cpu_count() or 1
    cores = max(1, min(avail, max_threads))
    torch.set_num_threads(cores)
    m = copy.deepcopy(model).cpu().train()
    for mod in m.modules():
        mod.fused_eval = False
        mod.fused_train = False
    opt = torch.optim.AdamW(m.parameters(), lr=1e-4, weight_decay=0.01)
    h, w, nb = H // div, W // div, 2
    g = torch.Generator().manual_seed(0)

    def one(hh, ww):
        x = torch.randn(nb, 3, hh, ww, generator=g)
        lab = torch.randint(0, C, (nb, hh, ww), generator=g)
        dens, dep = torch.rand(nb, hh, ww, generator=g), torch.rand(nb, hh, ww, generator=g)
        opt.zero_grad()
        out = m(x)
        loss = (F.cross_entropy(out["segmentation"], lab, reduction="none") * (1.0 + 2.0 * dens)).mean() + 0.1 * F.mse_loss(out["depth"].squeeze(1), dep)
        loss.backward()
        torch.nn.utils.clip_grad_norm_(m.parameters(), 1.0)
        opt.step()
    one(64, 64)                                              # warm-up (thread pool, allocator, optimizer state)
    t0 = time.perf_counter()
    one(h, w)
    dt = (time.perf_counter() - t0) * div * div / nb
    return {"value": round(1.0 / dt, 5), "unit": "images/s", "cores": cores, "kind": "port",
            "sample": f"one as-written torch-CPU optimisation step (forward + loss + backward + clip + AdamW) on {nb} frames of {h}x{w}, x{div * div} "
                      f"= {dt:.1f} s/frame ({cores} torch threads)"}


GLOBAL_FRAMES = 160          # C2 / C3 evaluate N = 160 frames (8 GPUs x 5 conditions x 4)
CONDITIONS = ["clean", "fog", "rain", "snow", "night"]


def parse_args(argv=None):
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=20)
    ap.add_argument("--warmup", type=int, default=5, help="untimed steps; the first ~4 steps after start-up run ~10 %% slower (allocator growth, clocks)")
    ap.add_argument("--batch", type=int, default=8, help="frames per GPU per step (README.md:125 batch size)")
    ap.add_argument("--height", type=int, default=1024)
    ap.add_argument("--width", type=int, default=2048)
    ap.add_argument("--frames", type=int, default=GLOBAL_FRAMES, help="size of the fixed global sample set")
    ap.add_argument("--no-depth", action="store_true", help="build the ensemble with include_depth=False")
    ap.add_argument("--no-stats", action="store_true", help="leave ECE + disagreement-AUROC accumulation (evaluate.py:230-255) out of the step")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--resident", action="store_true",
                    help="time the step on frames that are already resident in HBM (round-2 behaviour) instead of ingesting every batch from "
                         "pinned host memory on a side stream (REF/scripts/evaluate.py:172-173 starts each batch from host memory)")
    ap.add_argument("--resident-steps", type=int, default=5, help="steps of the resident-frames comparison pass (0: skip)")
    ap.add_argument("--no-parity-pass", action="store_true", help="skip the untimed pass over the whole global sample set (miou is then the timed region's)")
    ap.add_argument("--kernel-steps", type=int, default=2, help="steps of the untimed per-kernel HIP-event pass (0: no roofline / kernels)")
    ap.add_argument("--fp32-steps", type=int, default=3, help="steps of the float32-input-MFMA comparison pass at N=1 (0: skip)")
    ap.add_argument("--fp32-mfma", action="store_true",
                    help="run the MAIN measurement on the float32-input MFMA kernels (own fp32 attention, hipBLASLt fp32, fp32 Winograd) "
                         "instead of the split-operand f16-MFMA ones (DESIGN.md 5b); same as AWSEG_ATTN_SPLIT=0 AWSEG_GEMM_SPLIT=0 AWSEG_WINO_SPLIT=0")
    ap.add_argument("--model", choices=["b0_r50", "b5_r101"], default="b0_r50",
                    help="b0_r50: BASELINE configs[1] (float32 results); b5_r101: configs[4] (SegFormer-B5 + DeepLabV3+-R101, bf16 MFMA path)")
    ap.add_argument("--conv-search", type=int, default=int(os.environ.get("AWSEG_CONV_SEARCH", "0")),
                    help="1: let MIOpen time its solvers per convolution shape during warm-up (torch.backends.cudnn.benchmark)")
    ap.add_argument("--deterministic-convs", action="store_true", help="torch.backends.cudnn.deterministic = True for the two 7x7 stems left on MIOpen (measured 6x slower)")
    ap.add_argument("--mode", choices=["eval", "train"], default="eval",
                    help="eval: the north-star metric (BASELINE configs[1]); train: one AdverseWeatherTrainer optimisation step per "
                         "bench step (BASELINE configs[3]: ensemble + FogDensityAwareLoss + depth heads, bs 8 per GPU, DP gradient all-reduce)")
    ap.add_argument("--no-conv-search", action="store_true", help="train mode: keep MIOpen in immediate mode (no solver search)")
    ap.add_argument("--dry-run", action="store_true",
                    help="rendezvous only (gloo, CPU): every rank reports its block of the global sample set, rank 0 prints the pooled "
                         "frame count — exercises the self-launch / sharding path where there is no GPU (tests/)")
    ap.add_argument("--fail-rank", type=int, default=-1, help=argparse.SUPPRESS)      # tests: this rank exits 3 right after start-up
    return ap.parse_args(argv)


def dry_run(args) -> None:
    import torch
    import torch.distributed as dist
    from adverse_weather_semantic_segmentation_robustness_benchmark_amd import parallel
    os.environ["AWSEG_DIST_BACKEND"] = "gloo"
    rank, _, world = parallel.init_from_env(backend="gloo")
    if rank == args.fail_rank:
        raise SystemExit(3)
    mine = parallel.shard_range(args.frames, rank, world)
    t = torch.zeros(args.frames, dtype=torch.int64)
    t[list(mine)] = 1
    if parallel.is_dist():
        dist.all_reduce(t)
    if rank == 0:
        print(json.dumps({"dry_run": True, "n_gpus": world, "frames": args.frames, "frames_owned_once": bool((t == 1).all().item()),
                          "frames_rank0": len(mine), "backend": dist.get_backend() if parallel.is_dist() else "none"}), flush=True)
    if parallel.is_dist():
        dist.destroy_process_group()


def self_launch(args) -> int:
    """`python bench.py --gpus N` without a launcher: start N rank processes (children of a parent that never
    touches the GPU), rank 0 inherits stdout so its JSON line is this command's output.  Returns the exit code."""
    n = args.gpus
    with socket.socket() as sk:                       # a free rendezvous port
        sk.bind(("127.0.0.1", 0))
        port = sk.getsockname()[1]
    env = dict(os.environ, WORLD_SIZE=str(n), MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), HSA_ENABLE_IPC_MODE_LEGACY="0",
               AWSEG_SELF_LAUNCHED="1")
    procs = []
    for r in range(n):
        e = dict(env, RANK=str(r), LOCAL_RANK=str(r))
        procs.append(subprocess.Popen([sys.executable, str(Path(__file__).resolve()), *sys.argv[1:]], env=e,
                                      stdout=None if r == 0 else subprocess.DEVNULL))
    rc = 0
    try:
        pending = set(range(n))
        while pending:
            for r in sorted(pending):
                code = procs[r].poll()
                if code is None:
                    continue
                pending.discard(r)
                if code != 0 and rc == 0:
                    rc = code if code > 0 else 1
                    print(f"bench.py: rank {r} exited with code {code}; stopping the other ranks", file=sys.stderr)
                    for q in pending:
                        procs[q].terminate()
            time.sleep(0.05)
    finally:
        for pr in procs:
            if pr.poll() is None:
                pr.kill()
    return rc


def train_main(args):
    """BASELINE configs[3]: the AdverseWeatherTrainer step at its stated shape.  One bench step = one optimisation step on a
    batch of B frames per rank: weather corruption + Normalize (HIP) and the depth target (HIP) from resident uint8 frames,
    ensemble forward in TRAINING mode (the reference's op graph on torch-ROCm, autograd), per-batch fog-density field (HIP,
    Philox), FogDensityAwareLoss forward + backward (HIP), backward, bucketed gradient all-reduce overlapped with backward
    (RCCL), gradient clipping, AdamW.  value = N*B*K / max-over-ranks time."""
    from adverse_weather_semantic_segmentation_robustness_benchmark_amd import parallel
    from adverse_weather_semantic_segmentation_robustness_benchmark_amd.models.model import EnsembleModel
    from adverse_weather_semantic_segmentation_robustness_benchmark_amd.training.trainer import AdverseWeatherTrainer
    from adverse_weather_semantic_segmentation_robustness_benchmark_amd.data.preprocessing import WeatherDegradationTransforms, DepthEstimationPreprocessor
    import tempfile
    n_dev = torch.cuda.device_count()
    if int(os.environ.get("WORLD_SIZE", "1")) > max(n_dev, 1) and "AWSEG_DIST_BACKEND" not in os.environ:
        os.environ["AWSEG_DIST_BACKEND"] = "gloo"
    rank, local, world = parallel.init_from_env()
    if world != max(args.gpus, 1):
        raise SystemExit(f"bench.py: --gpus {args.gpus} but the launcher started WORLD_SIZE={world} ranks")
    dev = torch.device("cuda", local % max(n_dev, 1))
    torch.cuda.set_device(dev)
    backend = torch.distributed.get_backend() if parallel.is_dist() else "none"
    B, H, W, C = args.batch, args.height, args.width, 19
    torch.manual_seed(42)
    model = EnsembleModel(num_classes=C, include_depth=True, pretrained=False)
    tf = WeatherDegradationTransforms(seed=None, rng="philox", device=dev)
    tf._frame_seed = 1234
    depth_pre = DepthEstimationPreprocessor(dev)
    n_frames = min(args.frames, 4 * B * world)
    mine = list(parallel.shard_range(n_frames, rank, world))
    gen = torch.Generator(device=dev)
    raw = torch.empty(len(mine), H, W, 3, dtype=torch.uint8, device=dev)
    labels = torch.empty(len(mine), H, W, dtype=torch.uint8, device=dev)
    for k, g in enumerate(mine):
        gen.manual_seed(42 * 1000003 + g)
        raw[k] = torch.randint(0, 255, (H, W, 3), dtype=torch.uint8, device=dev, generator=gen)
        labels[k] = torch.randint(0, C, (H, W), dtype=torch.uint8, device=dev, generator=gen)

    class Batches:
        """`n` batches of B resident frames (cycling through the rank's block), born on the GPU."""
        def __init__(self, first, n):
            self.first, self.n = first, n

        def __len__(self):
            return self.n

        def __iter__(self):
            for i in range(self.first, self.first + self.n):
                if rank == 0:
                    print(f"[bench train] batch {i - self.first + 1}/{self.n} enqueued at {time.perf_counter() - t_start:.1f} s", file=sys.stderr, flush=True)
                idx = [(i * B + k) % len(mine) for k in range(B)]
                ids = [mine[k] for k in idx]
                conds = [CONDITIONS[g % len(CONDITIONS)] for g in ids]
                sel = torch.tensor(idx, dtype=torch.int64).pin_memory().to(dev, non_blocking=True)
                r, l = raw.index_select(0, sel), labels.index_select(0, sel)
                image = torch.empty(B, 3, H, W, dtype=torch.float32, device=dev)
                frames = torch.empty_like(r)
                tf.apply_batch(r, conds, out=frames, norm_out=image, frame_ids=ids)
                yield {"image": image, "label": l, "weather_condition": conds, "depth": depth_pre.estimate_depth_batch(frames),
                       "dataset": ["synthetic"] * B}

    t_start = time.perf_counter()
    # Solver choice: MIOpen in immediate mode with MIOPEN_FIND_MODE=FAST (set in main()): find-db, else its heuristic pick.
    # torch.backends.cudnn.benchmark (--conv-search 1) makes every new shape go through miopenFind*, which on a fresh box sat
    # for minutes inside ONE forward 1x1 convolution (gpurun_out/train_small.err, MIOPEN_ENABLE_LOGGING_CMD=1) — off by default.
    torch.backends.cudnn.benchmark = bool(args.conv_search) and not args.no_conv_search
    config = {"epochs": 1, "num_classes": C, "optimizer": {"type": "adamw", "learning_rate": 1e-4, "weight_decay": 0.01},
              "loss": {"type": "fog_density_aware"}, "grad_clip": 1.0, "seed": 42}
    tmp = tempfile.mkdtemp(prefix="awseg_bench_")
    trainer = AdverseWeatherTrainer(model, Batches(0, args.warmup), None, config, dev, checkpoint_dir=tmp + "/ck", log_dir=tmp + "/lg")
    trainer.train_epoch()                                    # warm-up steps (allocator, MIOpen solver choice)
    torch.cuda.synchronize()
    parallel.barrier()
    trainer.train_loader = Batches(args.warmup, args.steps)
    torch.cuda.reset_peak_memory_stats(dev)
    t0 = time.perf_counter()
    res = trainer.train_epoch()                              # ONE host synchronisation, at its end (device-resident running sums)
    torch.cuda.synchronize()
    parallel.barrier()
    dt = time.perf_counter() - t0
    tmax = torch.tensor([dt], dtype=torch.float64, device=dev)
    if parallel.is_dist():
        torch.distributed.all_reduce(tmax, op=torch.distributed.ReduceOp.MAX)
    dt = float(tmax.item())
    # ---- per-kernel pass (untimed): HIP event pairs around every C-ABI launch of ONE more step -> roofline of the dominant
    # hand-written kernel (the convolutions' forward / backward run on MIOpen: profiles/r03_train_step_kernels.csv lists them)
    from adverse_weather_semantic_segmentation_robustness_benchmark_amd import _native
    _native.launch_hook = CLOCK.hook
    kernels, roofline = [], None
    if args.kernel_steps > 0:
        trainer.train_loader = Batches(args.warmup + args.steps, 1)
        torch.cuda.synchronize()
        CLOCK.enabled = True
        trainer.train_epoch()
        torch.cuda.synchronize()
        CLOCK.enabled = False
        step_ms_t = dt / args.steps * 1e3
        for name, (count, avg_ms, own_work) in CLOCK.summary().items():
            if own_work is None:                             # (the per-condition weather launches cover 1-2 frames each here: priced in the eval line)
                continue
            bound, work = own_work
            if work <= 0 or avg_ms <= 0:
                continue
            if bound == "hbm":
                achieved, peak, unit = work / (avg_ms * 1e-3) / 1e9, HBM_PEAK_GBS, "GB/s"
            elif bound == "mfma_f16":
                achieved, peak, unit = work / (avg_ms * 1e-3) / 1e12, MFMA_F16_PEAK_TFLOPS, "TFLOP/s"
            else:
                achieved, peak, unit = work / (avg_ms * 1e-3) / 1e12, MFMA_F32_PEAK_TFLOPS, "TFLOP/s"
            kernels.append({"kernel": name, "launches_per_step": count, "avg_ms": round(avg_ms, 4), "bound": bound, "achieved": round(achieved, 2),
                            "peak": peak, "unit": unit, "frac": round(achieved / peak, 4), "time_share_of_step": round(count * avg_ms / step_ms_t, 4)})
        kernels.sort(key=lambda k: -k["launches_per_step"] * k["avg_ms"])
        if kernels:
            k0 = kernels[0]
            traffic, tsrc = pmc_traffic(k0["kernel"], train=True)
            roofline = {"kernel": k0["kernel"], "bound": ("mfma" if k0["bound"].startswith("mfma") else "hbm"), "achieved": k0["achieved"],
                        "peak": k0["peak"], "unit": k0["unit"], "frac": k0["frac"], "traffic": traffic, "traffic_source": tsrc,
                        "measured": "HIP events around each launch on the launching stream, one untimed step after the timed region; dominant HAND-WRITTEN "
                                    "kernel (the forward / backward convolutions are MIOpen's: profiles/r04_train_step_kernels.csv)"}
    cpu = None
    if rank == 0 and world == 1 and not args.no_cpu_baseline:
        try:
            cpu = cpu_train_baseline(model, H, W, C)
        except Exception as e:  # noqa: BLE001
            cpu = {"value": None, "unit": "images/s", "cores": os.cpu_count(), "kind": "port", "sample": f"failed: {e!r}"}
    # every replica must hold the same parameters after the timed steps (data-parallel averaging): the per-rank float64 sums of all
    # parameters, gathered — rank 0 reports them and whether they are identical (tests/test_gpu_configs.py, RCCL-gated)
    with torch.no_grad():
        psum = torch.stack([p.detach().double().sum() for p in model.parameters()]).sum().reshape(1)
        pabs = torch.stack([p.detach().double().abs().sum() for p in model.parameters()]).sum().reshape(1)
    mine2 = torch.cat([psum, pabs]).to(dev)
    if parallel.is_dist():
        gathered = [torch.zeros_like(mine2) for _ in range(world)]
        torch.distributed.all_gather(gathered, mine2)
    else:
        gathered = [mine2]
    replica_sums = [[float(v) for v in g.cpu()] for g in gathered]
    if rank == 0:
        line = {
            "metric": f"images/sec ({H}x{W}, AdverseWeatherTrainer train step: ensemble + FogDensityAwareLoss + depth heads)",
            "value": round(B * args.steps * world / dt, 3), "unit": "images/s", "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
            "ms_per_step": round(dt / args.steps * 1e3, 3), "higher_is_better": True, "scaling": "weak", "vs_baseline": None,
            "dtype": "f32", "data": "synthetic",
            "config": {"workload": f"train_step_{H}x{W}_bs{B} (BASELINE.json configs[3]: SegFormer-B0 + DeepLabV3+-R50 ensemble, FogDensityAwareLoss, "
                                   "depth heads, AdamW, grad clip 1.0)", "per_gpu_batch": B, "global_batch": B * world,
                       "forward_backward": "reference op graph on torch-ROCm (autograd); loss forward/backward, density field, weather, "
                                           "Normalize and depth target on HIP kernels", "dist_backend": backend,
                       "gradient_all_reduce": "flat float32 buckets of 32 MB, launched from post-accumulate-grad hooks during backward"},
            "rccl_ranks": world if backend == "nccl" else 0,
            "roofline": roofline, "cpu_baseline": cpu, "kernels": kernels,
            "peak_hbm_gb": round(torch.cuda.max_memory_allocated(dev) / 2 ** 30, 1),
            "losses": {k: round(float(v), 6) for k, v in res.items()},
            "replica_parameter_sums": replica_sums, "replicas_identical": all(r == replica_sums[0] for r in replica_sums),
        }
        print(json.dumps(line), flush=True)
    if parallel.is_dist():
        torch.distributed.destroy_process_group()


def main():
    global np, torch
    args = parse_args()
    if args.gpus > 1 and "WORLD_SIZE" not in os.environ:
        raise SystemExit(self_launch(args))
    if args.dry_run:
        return dry_run(args)
    if args.mode == "train" and "MIOPEN_FIND_MODE" not in os.environ:
        # training shapes miss MIOpen's find-db on a fresh box; its default (hybrid) mode then BENCHMARKS every candidate
        # solver of every backward convolution — naive ones included — which takes minutes per shape at these sizes
        # (gpurun_out/r02_train_bs2.err).  FAST = find-db, else the heuristic pick, no benchmarking.
        os.environ["MIOPEN_FIND_MODE"] = "FAST"
    import numpy as np                     # noqa: F811
    import torch                           # noqa: F811
    if args.mode == "train":
        return train_main(args)
    torch.backends.cudnn.benchmark = bool(args.conv_search)
    # (the strided / patch convolutions run as im2col + GEMM on this repo's kernels: MIOpen's default pick for them is a
    # split-K igemm that accumulates with atomics, run-to-run different — tools/check_op_determinism.py.  Forcing
    # torch.backends.cudnn.deterministic instead costs 6x: 381 ms/step, gpurun_out/r02_bench_b.json)
    torch.backends.cudnn.deterministic = bool(args.deterministic_convs)

    from adverse_weather_semantic_segmentation_robustness_benchmark_amd import _native, ops, parallel
    if args.fp32_mfma:
        ops.set_split(False)
    from adverse_weather_semantic_segmentation_robustness_benchmark_amd.models.model import EnsembleModel
    from adverse_weather_semantic_segmentation_robustness_benchmark_amd.evaluation.metrics import RobustnessMetrics
    from adverse_weather_semantic_segmentation_robustness_benchmark_amd.evaluation.harness import EvalState, eval_batch, finalize
    from adverse_weather_semantic_segmentation_robustness_benchmark_amd.data.preprocessing import WeatherDegradationTransforms

    n_dev = torch.cuda.device_count()      # does not initialise the GPU runtime
    world_env = int(os.environ.get("WORLD_SIZE", "1"))
    if world_env > max(n_dev, 1) and "AWSEG_DIST_BACKEND" not in os.environ:
        # more ranks than GPUs (rehearsal on a 1-GPU box): RCCL cannot put two ranks on one device
        os.environ["AWSEG_DIST_BACKEND"] = "gloo"
    rank, local, world = parallel.init_from_env()
    if world != max(args.gpus, 1):
        raise SystemExit(f"bench.py: --gpus {args.gpus} but the launcher started WORLD_SIZE={world} ranks")
    dev = torch.device("cuda", local % max(n_dev, 1))
    torch.cuda.set_device(dev)
    backend = torch.distributed.get_backend() if parallel.is_dist() else "none"
    B, H, W, C = args.batch, args.height, args.width, 19
    bf16 = args.model == "b5_r101"

    torch.manual_seed(42)                  # identical weights on every rank
    np.random.seed(42)
    kw = {}
    if bf16:
        kw = {"segformer_name": "nvidia/segformer-b5-finetuned-cityscapes-1024-1024", "deeplab_backbone": "resnet101", "compute_dtype": "bf16"}
    model = EnsembleModel(num_classes=C, include_depth=not args.no_depth, pretrained=False, **kw).to(dev).eval()
    metrics = RobustnessMetrics(num_classes=C, weather_conditions=CONDITIONS)
    tf = WeatherDegradationTransforms(seed=None, rng="philox", device=dev)
    tf._frame_seed = 1234

    # ---- the rank's block of the fixed global sample set, resident in HBM ---------------------------------------
    mine = list(parallel.shard_range(args.frames, rank, world))
    if not mine:
        raise SystemExit(f"bench.py: rank {rank} owns no frame of a {args.frames}-frame set")
    gen = torch.Generator(device=dev)
    raw = torch.empty(len(mine), H, W, 3, dtype=torch.uint8, device=dev)
    labels = torch.empty(len(mine), H, W, dtype=torch.uint8, device=dev)
    for k, g in enumerate(mine):
        gen.manual_seed(42 * 1000003 + g)
        raw[k] = torch.randint(0, 255, (H, W, 3), dtype=torch.uint8, device=dev, generator=gen)      # loader.py:206
        labels[k] = torch.randint(0, C, (H, W), dtype=torch.uint8, device=dev, generator=gen)        # loader.py:231
    image = torch.empty(B, 3, H, W, dtype=torch.float32, device=dev)
    _native.launch_hook = CLOCK.hook
    info = {}
    with_stats = not args.no_stats

    def new_state():
        return EvalState(metrics, CONDITIONS, dev, 15, ensemble=True)

    def run_frames(st, local_idx, buffers=None):
        """One step over the rank's frames `local_idx` (positions in its block): resident in HBM, or — `buffers` — just ingested."""
        ids = [mine[k] for k in local_idx]
        conds = [CONDITIONS[g % len(CONDITIONS)] for g in ids]
        info.update({"awseg_fog_fused": conds.count("fog"), "awseg_night_apply": conds.count("night"),
                     "awseg_rain_apply": conds.count("rain"), "awseg_snow_apply": conds.count("snow"),
                     "awseg_normalize": conds.count("clean"), "awseg_weather_batch": len(ids)})   # (upper bound for the batch launch: a frame that drew the 7x7 snow blur runs in awseg_snow_apply)
        if buffers is not None:
            r, l = buffers[0][:len(local_idx)], buffers[1][:len(local_idx)]
        elif local_idx == list(range(local_idx[0], local_idx[0] + len(local_idx))):
            r, l = raw[local_idx[0]:local_idx[0] + len(local_idx)], labels[local_idx[0]:local_idx[0] + len(local_idx)]
        else:
            # a batch that wraps around the rank's block: gather it (index built on the host and copied without a
            # stream synchronisation — torch.tensor(..., device=cuda) would wait for the GPU to drain, every step)
            sel = torch.tensor(local_idx, dtype=torch.int64).pin_memory().to(dev, non_blocking=True)
            r, l = raw.index_select(0, sel), labels.index_select(0, sel)
        img = image[:len(ids)]
        tf.apply_batch(r, conds, norm_out=img, frame_ids=ids)
        eval_batch(model, st, img, l, conds, metrics, with_stats=with_stats)

    # ---- ingestion (REF/scripts/evaluate.py:172-173: every batch starts in host memory).  The rank's frames + labels also live in
    # PINNED host memory; step i runs on a device double buffer that a side stream filled during step i-1 (uint8: 6.3 + 2.1 MB per
    # frame, ~70 MB per batch of 8 at ~50 GB/s = 1.4 ms of a 44 ms step, off the compute stream), and the copy of batch i+1 is
    # enqueued before step i's kernels.
    ingest = not args.resident
    if ingest:
        host_raw, host_lab = raw.cpu().pin_memory(), labels.cpu().pin_memory()
        dbuf = [(torch.empty(B, H, W, 3, dtype=torch.uint8, device=dev), torch.empty(B, H, W, dtype=torch.uint8, device=dev)) for _ in range(2)]
        copy_stream = torch.cuda.Stream(device=dev)
        copied = [torch.cuda.Event(), torch.cuda.Event()]
        consumed = [torch.cuda.Event(), torch.cuda.Event()]
        staged = [None, None]                         # step index whose batch each buffer holds (or is receiving)

        def enqueue_copy(i):
            """host -> device copy of step i's batch into buffer i % 2 on the side stream (after the step that last used it)"""
            k = i % 2
            idx = [(i * B + j) % len(mine) for j in range(B)]
            with torch.cuda.stream(copy_stream):
                copy_stream.wait_event(consumed[k])
                r, l = dbuf[k]
                if idx == list(range(idx[0], idx[0] + B)):
                    r.copy_(host_raw[idx[0]:idx[0] + B], non_blocking=True); l.copy_(host_lab[idx[0]:idx[0] + B], non_blocking=True)
                else:
                    for j, g in enumerate(idx):
                        r[j].copy_(host_raw[g], non_blocking=True); l[j].copy_(host_lab[g], non_blocking=True)
                copied[k].record(copy_stream)
            staged[k] = i
        for k in range(2):
            consumed[k].record()

    def step(st, i):
        idx = [(i * B + k) % len(mine) for k in range(B)]
        if not ingest or not staging_on[0]:
            return run_frames(st, idx)
        k = i % 2
        if staged[k] != i:
            enqueue_copy(i)
        torch.cuda.current_stream().wait_event(copied[k])
        if staged[1 - k] != i + 1:
            enqueue_copy(i + 1)                        # next batch travels while this one computes
        run_frames(st, idx, buffers=dbuf[k])
        consumed[k].record()

    staging_on = [True]

    def finish(st):
        """Counters -> all-reduce -> the result dict (host math of evaluate.py:214-271)."""
        return finalize(st, metrics)

    def timed(n_steps, first):
        st = new_state()
        torch.cuda.synchronize()
        parallel.barrier()
        t0 = time.perf_counter()
        for i in range(n_steps):
            step(st, first + i)
        res = finish(st)                           # inside the timed region: the job is not done until the metric exists
        torch.cuda.synchronize()
        parallel.barrier()
        dt = time.perf_counter() - t0
        tmax = torch.tensor([dt], dtype=torch.float64, device=dev)
        if parallel.is_dist():
            torch.distributed.all_reduce(tmax, op=torch.distributed.ReduceOp.MAX)
        return float(tmax.item()), res

    with torch.no_grad():
        st = new_state()
        for i in range(args.warmup):
            step(st, i)
        finish(st)                                 # warm the host-side finalisation too (first-use costs of the CPU ops)
        dt, timed_results = timed(args.steps, args.warmup)

        # ---- parity pass: every global frame exactly once, by its owner ------------------------------------------
        results = timed_results
        if not args.no_parity_pass:
            st = new_state()
            for lo in range(0, len(mine), B):
                run_frames(st, list(range(lo, min(lo + B, len(mine)))))
            results = finish(st)

        # ---- per-kernel pass: HIP event pairs around every C-ABI launch (untimed) ----------------------------------
        if args.kernel_steps > 0:
            st = new_state()
            torch.cuda.synchronize()
            # (one stream for this pass: with the two ensemble members on two streams a launch's event pair also times whatever the
            # other stream runs beside it — the per-kernel figures are those of each kernel with the chip to itself)
            two_streams, ops.TWO_STREAMS = ops.TWO_STREAMS, False
            CLOCK.enabled = True
            for i in range(args.kernel_steps):
                step(st, args.warmup + args.steps + i)
            torch.cuda.synchronize()
            CLOCK.enabled = False
            ops.TWO_STREAMS = two_streams

        # ---- float32-input MFMA comparison (N = 1) -----------------------------------------------------------------
        fp32 = None
        if world == 1 and args.fp32_steps > 0 and not args.fp32_mfma and not bf16:
            saved = ops.split_state()
            ops.set_split(False)
            st = new_state()
            for i in range(2):
                step(st, i)
            dt32, res32 = timed(args.fp32_steps, 2)
            ops.restore_split(saved)
            fp32 = {"value": round(B * args.fp32_steps / dt32, 3), "unit": "images/s", "ms_per_step": round(dt32 / args.fp32_steps * 1e3, 3),
                    "steps": args.fp32_steps, "what": "same step with attention / 1x1 GEMMs / Winograd on the float32-input MFMA kernels "
                    "(v_mfma_f32_32x32x2_f32, hipBLASLt f32) instead of split-operand f16 MFMA"}

        # ---- resident-frames comparison: the same step without ingestion (the round-1/2 figure) --------------------------------
        resident = None
        if ingest and args.resident_steps > 0:
            staging_on[0] = False
            st = new_state()
            step(st, 0)
            dtr, _ = timed(args.resident_steps, 1)
            staging_on[0] = True
            resident = {"value": round(B * args.resident_steps * world / dtr, 3), "unit": "images/s", "ms_per_step": round(dtr / args.resident_steps * 1e3, 3),
                        "steps": args.resident_steps, "what": "same step on frames already resident in HBM (no host -> device copy)"}

    total_images = B * args.steps * world
    step_ms = dt / args.steps * 1e3

    # ---- roofline of the dominant hand-written kernel (rank 0's launches, untimed pass) ------------
    kernels = []
    for name, (count, avg_ms, own_work) in CLOCK.summary().items():
        bound, work = own_work if own_work is not None else algorithmic_work(name, B, H, W, C, info)
        if work <= 0 or avg_ms <= 0:
            continue
        if bound == "hbm":
            achieved, peak, unit = work / (avg_ms * 1e-3) / 1e9, HBM_PEAK_GBS, "GB/s"
        elif bound == "mfma_f16":
            achieved, peak, unit = work / (avg_ms * 1e-3) / 1e12, MFMA_F16_PEAK_TFLOPS, "TFLOP/s"
        else:
            achieved, peak, unit = work / (avg_ms * 1e-3) / 1e12, MFMA_F32_PEAK_TFLOPS, "TFLOP/s"
        per_step = count / max(args.kernel_steps, 1)
        kernels.append({"kernel": name, "launches_per_step": round(per_step, 2), "avg_ms": round(avg_ms, 4), "bound": bound,
                        "achieved": round(achieved, 2), "peak": peak, "unit": unit, "frac": round(achieved / peak, 4),
                        "time_share_of_step": round(per_step * avg_ms / step_ms, 4)})
    kernels.sort(key=lambda k: -k["launches_per_step"] * k["avg_ms"])
    roofline = None
    if kernels:
        k0 = kernels[0]
        traffic, tsrc = pmc_traffic(k0["kernel"], b5=bf16)
        roofline = {"kernel": k0["kernel"], "bound": ("mfma" if k0["bound"].startswith("mfma") else "hbm"), "achieved": k0["achieved"],
                    "peak": k0["peak"], "unit": k0["unit"], "frac": k0["frac"], "traffic": traffic, "traffic_source": tsrc,
                    "measured": f"HIP events around each launch on the launching stream, {args.kernel_steps} untimed steps after the timed region "
                                "(the two ensemble members on ONE stream for this pass; the timed region runs them on two)"}

    if rank == 0:
        cpu = None
        if world == 1 and not args.no_cpu_baseline:
            try:
                # B5 + R101 on the host: the forward on a quarter of the pixels, time x 4 (keeps the sample inside ~30 s)
                cpu = cpu_baseline(model, H, W, C, fwd_div=2 if bf16 else 1)
            except Exception as e:  # noqa: BLE001
                cpu = {"value": None, "unit": "images/s", "cores": os.cpu_count(), "kind": "port", "sample": f"failed: {e!r}"}
        split = ops.split_state()
        if bf16:
            dtype = "bf16 (bf16 MFMA operands, f32 accumulate)"
        elif any(split.values()):
            dtype = "f32 (results float32; " + ", ".join(k for k, v in split.items() if v) + " on 3xf16 split operands: 22-bit operands, f32 accumulate)"
        else:
            dtype = "f32"
        workload = (f"ensemble_eval_{H}x{W}_5cond (BASELINE.json configs[1]: SegFormer-B0 + DeepLabV3+-R50, all 5 weather conditions)" if not bf16 else
                    f"ensemble_eval_{W}x{H}_b5_r101_bf16 (BASELINE.json configs[4]: SegFormer-B5 + DeepLabV3+-R101, bf16 MFMA path)")
        line = {
            "metric": "images/sec (1024x2048, 5 weather conds, ensemble eval)",
            "value": round(total_images / dt, 3), "unit": "images/s", "n_gpus": world, "steps": args.steps,
            "warmup": args.warmup, "ms_per_step": round(step_ms, 3), "higher_is_better": True,
            "scaling": "weak", "vs_baseline": None, "dtype": dtype, "data": "synthetic",
            "config": {"workload": workload, "per_gpu_batch": B, "global_batch": B * world,
                       "global_sample_set": args.frames, "frames_per_rank": len(mine),
                       "ingest": ("h2d overlapped (pinned host -> HBM double buffer on a side stream, uint8 frames + labels)" if ingest
                                  else "none (frames resident in HBM)"),
                       "include_depth": not args.no_depth, "eval_stats_in_step": with_stats,
                       "weather_rng": "philox (in-kernel), keyed by global frame index", "ensemble_logits_materialised": False,
                       "split_operand_kernels": (None if bf16 else split),
                       "bf16_mfma_kernels": (["awseg_gemm_bf16_bias_act", "awseg_conv3x3_winograd_bf16_nhwc", "awseg_attention_d32_bf16"] if bf16 else None),
                       "weights": "random init (no checkpoints offline)",
                       "parallelism": f"batch-sharded x{world}, one counter all-reduce (int64 confusion + ECE bins + AUROC histogram)",
                       "ensemble_members_on_two_streams": bool(ops.TWO_STREAMS),
                       "dist_backend": backend},
            "rccl_ranks": world if backend == "nccl" else 0,
            "roofline": roofline, "cpu_baseline": cpu, "fp32_mfma": fp32, "resident_frames": resident if ingest else None, "kernels": kernels,
            "miou": {k: round(float(v), 6) for k, v in results.items()},
            "miou_source": ("parity pass: every frame of the global set exactly once, sharded by parallel.shard_range, counters all-reduced"
                            if not args.no_parity_pass else "timed region"),
            "timed_region_overall_miou": round(float(timed_results["overall_miou"]), 6),
        }
        print(json.dumps(line), flush=True)
    if parallel.is_dist():
        torch.distributed.destroy_process_group()


if __name__ == "__main__":
    main()
