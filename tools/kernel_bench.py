#!/usr/bin/env python3
"""Per-kernel micro-benchmark: every hand-written kernel on a full batch (B frames of HxW of the
SAME kind per launch), timed with HIP events over repeated launches on the launching stream, against
its roofline.  Interleaved rounds in one process (cdna_hip_programming.md rule 24).

    python tools/kernel_bench.py [--batch 8] [--height 1024] [--width 2048] [--iters 20] [--only fog,night]
"""
import argparse
import json
import sys
from pathlib import Path

import numpy as np
import torch

ROOT = Path(__file__).resolve().parent.parent
sys.path.insert(0, str(ROOT))
from adverse_weather_semantic_segmentation_robustness_benchmark_amd import ops  # noqa: E402
from adverse_weather_semantic_segmentation_robustness_benchmark_amd.data import preprocessing as P  # noqa: E402

HBM, MFMA = 8000.0, 157.3


def timeit(fn, iters, burst=10):
    """Median / min over `iters` rounds of the mean time of `burst` back-to-back launches between one
    HIP event pair on the launching stream (amortises the event + launch overhead, ~15 us, that
    dominates a single 50 us kernel).  Work per launch is unchanged."""
    for _ in range(3):
        fn()
    torch.cuda.synchronize()
    ts = []
    for _ in range(iters):
        s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        s.record()
        for _ in range(burst):
            fn()
        e.record()
        torch.cuda.synchronize()
        ts.append(s.elapsed_time(e) / burst)
    ts.sort()
    return ts[len(ts) // 2], ts[0]


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--batch", type=int, default=8)
    ap.add_argument("--height", type=int, default=1024)
    ap.add_argument("--width", type=int, default=2048)
    ap.add_argument("--iters", type=int, default=20)
    ap.add_argument("--only", type=str, default="")
    a = ap.parse_args()
    B, H, W, C = a.batch, a.height, a.width, 19
    px = H * W
    dev = "cuda"
    torch.manual_seed(0); np.random.seed(0)
    imgs = torch.randint(0, 255, (B, H, W, 3), dtype=torch.uint8, device=dev)
    out = torch.empty_like(imgs)
    norm = torch.empty(B, 3, H, W, device=dev)
    labels = torch.randint(0, C, (B, H, W), dtype=torch.uint8, device=dev)
    idx = list(range(B))
    cases = {}

    fj = ops.fog_jobs(idx, [0.5] * B, list(range(1, B + 1)))
    cases["fog_fused philox ->norm"] = (lambda: ops.fog(imgs, fj, norm_out=norm), "hbm", (3 + 12) * px * B)
    cases["fog_fused philox ->u8"] = (lambda: ops.fog(imgs, fj, out=out), "hbm", (3 + 3) * px * B)
    nj = ops.night_jobs(idx, [0.8] * B, [0.6] * B, list(range(1, B + 1)))
    cases["night philox ->norm"] = (lambda: ops.night(imgs, nj, norm_out=norm), "hbm", (3 + 12) * px * B)
    cases["night philox ->u8"] = (lambda: ops.night(imgs, nj, out=out), "hbm", (3 + 3) * px * B)
    rd = [P.draw_rain(H, W, 0.5) for _ in range(B)]
    rj, rp = ops.prim_jobs(idx, [d[0] for d in rd], [d[1] for d in rd])
    cases["rain ->norm"] = (lambda: ops.rain(imgs, rj, rp, norm_out=norm), "hbm", (3 + 12) * px * B)
    cases["rain ->u8"] = (lambda: ops.rain(imgs, rj, rp, out=out), "hbm", (3 + 3) * px * B)
    sd = [P.draw_snow(H, W, 0.5) for _ in range(B)]
    for ks in (3, 7):
        sj, sp = ops.prim_jobs(idx, [d[0] for d in sd], [d[1] for d in sd], [ks] * B)
        cases[f"snow k{ks} ->norm"] = (lambda sj=sj, sp=sp: ops.snow(imgs, sj, sp, norm_out=norm), "hbm", (3 + 12) * px * B)
    cases["normalize"] = (lambda: ops.normalize(imgs, out=norm), "hbm", 15 * px * B)

    s1 = torch.randn(B, C, H, W, device=dev); s2 = torch.randn(B, C, H, W, device=dev)
    wts = torch.tensor([0.5, 0.5], device=dev); T = torch.ones(1, device=dev)
    counts = ops.new_counts(C, dev, 6); oob = torch.zeros(1, dtype=torch.int64, device=dev)
    cond = torch.tensor([i % 5 for i in range(B)], dtype=torch.int32, device=dev)
    cases["combine+argmax+confusion (no logits out)"] = (
        lambda: ops.combine_argmax_confusion(s1, s2, 0, wts, T, want_logits=False, label=labels, counts=counts, oob=oob, cond=cond),
        "hbm", (2 * C * 4 + 1) * px * B)
    cases["combine+argmax+confusion (+logits out)"] = (
        lambda: ops.combine_argmax_confusion(s1, s2, 0, wts, T, want_logits=True, label=labels, counts=counts, oob=oob, cond=cond),
        "hbm", (3 * C * 4 + 1) * px * B)
    cases["argmax+confusion single model"] = (
        lambda: ops.combine_argmax_confusion(s1, None, 3, want_logits=False, label=labels, counts=counts, oob=oob, cond=cond),
        "hbm", (C * 4 + 1) * px * B)
    pred = torch.randint(0, C, (B, H, W), dtype=torch.uint8, device=dev)
    c1 = ops.new_counts(C, dev)
    cases["confusion from u8 predictions"] = (lambda: ops.confusion_accumulate(pred, labels, C, c1, oob), "hbm", 2 * px * B)
    dens = torch.rand(B, H, W, device=dev)
    cases["fog_ce forward"] = (lambda: ops.fog_ce_forward(s1, labels, dens, False, 2.0, oob), "hbm", (C * 4 + 1 + 4) * px * B)
    g = torch.ones(1, device=dev)
    cases["fog_ce backward"] = (lambda: ops.fog_ce_backward(s1, labels, dens, False, 2.0, g), "hbm", (2 * C * 4 + 1 + 4) * px * B)
    edges = torch.linspace(0, 1, 16).to(dev); bins = ops.new_ece_bins(15, dev, 6)
    cases["ece accumulate"] = (lambda: ops.ece_accumulate(s1, labels, bins, edges, cond), "hbm", (C * 4 + 1) * px * B)

    h, w = H // 32, W // 32
    g9 = torch.randn(B, h, w, 9, 256, device=dev)
    sc, sf = torch.rand(256, device=dev) + 0.5, torch.randn(256, device=dev) * 0.1
    w2, b2 = torch.randn(C, 256, device=dev) * 0.05, torch.zeros(C, device=dev)
    cases["segformer_head_fused (MFMA)"] = (lambda: ops.segformer_head_fused(g9, None, sf, w2, b2, H, W, split=False), "mfma",
                                             2.0 * (12 * 256 + 256 * 32) * px * B)
    # split operands: the same products three times over (two for GEMM 1 when the bilinear weights are exact in f16, as here)
    cases["segformer_head_fused [split f16x3, issued flops]"] = (lambda: ops.segformer_head_fused(g9, None, sf, w2, b2, H, W, split=True), "mfma_f16",
                                                                 2.0 * (2 * 16 * 256 + 3 * 256 * 32) * px * B)
    g9d = torch.randn(B, h, w, 9, 128, device=dev)
    cases["upconv3x3_bn_relu 128ch NHWC (512 B/px written)"] = (lambda: ops.upconv3x3_bn_relu(g9d, None, sf[:128].contiguous(), H, W, True),
                                                                "hbm", 128 * 4 * px * B)
    # the depth head as ONE launch: forms at the encoder's resolution, then the Winograd kernel with the patch generator (no hidden map)
    if H == 32 * h and W == 32 * w:
        fm = ops.upconv_forms(g9d, sf[:128].contiguous())
        usd = ops.winograd_split_weights(torch.randn(64, 128, 3, 3, device=dev) * 0.05)
        sh64, w64, b1 = torch.randn(64, device=dev), torch.randn(64, device=dev), torch.zeros(1, device=dev)
        cases["upconv_forms 128ch (tables at 1/32)"] = (lambda: ops.upconv_forms(g9d, sf[:128].contiguous()), "hbm", 4.0 * (g9d.numel() + fm.numel()))
        cases["depth_head_fused 128->64 +1x1+sigmoid @full, patch generated [split f16x3, issued flops]"] = (
            lambda: ops.depth_head_fused(fm, h, w, 128, usd, sh64, w64, b1), "mfma_f16", 3 * 2.0 * 16 * 128 * 64 * B * (H // 2) * (W // 2))
    # MiT Mix-FFN as one tile kernel (stage 1: 32 channels at 1/4, stage 2: 64 at 1/8): algorithmic bytes = tokens in + out
    for cm, dv in ((32, 4), (64, 8)):
        tk = torch.randn(B, H // dv, W // dv, cm, device=dev)
        gm, bt = torch.rand(cm, device=dev) + 0.5, torch.randn(cm, device=dev) * 0.1
        f1, fb1 = torch.randn(4 * cm, cm, device=dev) / cm ** 0.5, torch.randn(4 * cm, device=dev) * 0.1
        tp, tb = torch.randn(9, 4 * cm, device=dev) * 0.3, torch.randn(4 * cm, device=dev) * 0.1
        f2, fb2 = torch.randn(cm, 4 * cm, device=dev) / (4 * cm) ** 0.5, torch.randn(cm, device=dev) * 0.1
        s1w, s2w = ops.mixffn_split_weights(f1), ops.mixffn_split_weights(f2)
        cases[f"mixffn_fused C={cm} @1/{dv} (LN+fc1+dw3x3+GELU+fc2+res, one launch)"] = (
            lambda tk=tk, gm=gm, bt=bt, f1=f1, fb1=fb1, tp=tp, tb=tb, f2=f2, fb2=fb2, s1w=s1w, s2w=s2w:
            ops.mixffn_fused(tk, gm, bt, 1e-6, f1, fb1, tp, tb, f2, fb2, w1_split=s1w, w2_split=s2w, checked=True), "hbm", 8.0 * tk.numel())
    xa = torch.randn(B, H // 16, W // 16, 2048, device=dev); wdw = torch.randn(3, 9, 2048, device=dev)
    cases["aspp_depthwise3"] = (lambda: ops.aspp_depthwise3(xa, wdw, (12, 24, 36)), "hbm", 4 * 2048 * 4 * (H // 16) * (W // 16) * B)
    xd = torch.randn(B, H // 4, W // 4, 128, device=dev); w9 = torch.randn(9, 128, device=dev); bb = torch.randn(128, device=dev)
    cases["dwconv3x3_nhwc+gelu 128ch @1/4"] = (lambda: ops.dwconv3x3_nhwc(xd, w9, bb, 2), "hbm", 2 * 128 * 4 * (H // 4) * (W // 4) * B)
    xe = torch.randn(B, H // 4, W // 4, 256, device=dev); re_ = torch.randn_like(xe); b256 = torch.randn(256, device=dev)
    cases["bias_act_nhwc +res 256ch @1/4"] = (lambda: ops.bias_act_nhwc_(xe, b256, re_, 1), "hbm", 3 * 256 * 4 * (H // 4) * (W // 4) * B)

    cases["depth_estimate ->f32 (f64 ladder)"] = (lambda: ops.depth_estimate(imgs, dtype=torch.float32), "hbm", (3 + 4) * px * B)
    luts = torch.randint(0, 256, (4, 3, 256), dtype=torch.uint8, device=dev); lut_of = torch.tensor([i % 4 for i in range(B)], dtype=torch.int32, device=dev)
    cases["style lut3 u8->u8"] = (lambda: ops.lut3_apply(imgs, luts, lut_of, out=out), "hbm", 6 * px * B)
    cases["local_contrast (fog density map)"] = (lambda: ops.local_contrast(imgs), "hbm", (3 + 4) * px * B)
    hist = torch.zeros(2, 8192, dtype=torch.int64, device=dev)
    cnt6 = torch.zeros(6, C * C, dtype=torch.int64, device=dev); oob1 = torch.zeros(1, dtype=torch.int64, device=dev)
    cases["combine + confusion + eval stats in one pass"] = (
        lambda: ops.combine_confusion_stats(s1, s2, 0, wts, T, labels, cond, cnt6, oob1, edges, bins, hist, 0.0, 3.0), "hbm", (2 * C * 4 + 1) * px * B)
    cases["ensemble eval stats (ECE + disagreement hist)"] = (
        lambda: ops.ensemble_eval_stats(s1, s2, 0, wts, T, labels, cond, edges, bins, hist, 0.0, 3.0), "hbm", (2 * C * 4 + 1) * px * B)
    xmp = torch.randn(B, H // 2, W // 2, 64, device=dev)
    cases["maxpool3x3s2 nhwc 64ch (resnet stem)"] = (lambda: ops.maxpool3x3s2_nhwc(xmp), "hbm", 64 * 4 * (H // 2) * (W // 2) * B * (1 + 1 / 4))
    lowl = torch.randn(B, C, H // 4, W // 4, device=dev)
    cases["upsample_bilinear x4 19 planes (deeplab logits)"] = (lambda: ops.upsample_bilinear(lowl, (H, W), True), "hbm", C * 4 * px * B * (1 + 1 / 16))
    xl = torch.randn(B * (H // 4) * (W // 4), 32, device=dev); lw = torch.randn(32, device=dev); lb = torch.randn(32, device=dev)
    cases["layernorm_rows C=32 @1/4"] = (lambda: ops.layernorm_rows(xl, lw, lb, 1e-6), "hbm", 2 * 32 * 4 * xl.shape[0])

    # stride-2 3x3 convolution (resnet layer2 conv2: 128 -> 128 at 1/4 -> 1/8): A operand gathered inside the split GEMM vs im2col + the same GEMM
    xcv = torch.randn(B, H // 4, W // 4, 128, device=dev); wcv = torch.randn(128, 9 * 128, device=dev) * 0.03; bcv = torch.randn(128, device=dev)
    wscv = ops.gemm_split_weights(wcv)
    flc = 3 * 2.0 * B * (H // 8) * (W // 8) * 128 * 9 * 128
    cases["conv3x3 s2 128->128 @1/4 gathered in the split GEMM [split f16x3, issued flops]"] = (lambda: ops.conv_gemm_split(xcv, wscv, bcv, 1, 3, 3, 2, 1), "mfma_f16", flc)

    def _im2col_then_gemm():
        cols, _, _ = ops.im2col_nhwc(xcv, 3, 3, 2, 1, 1, 9 * 128)
        return ops.gemm_split_bias_act(cols, wscv, bcv, 1)
    cases["conv3x3 s2 128->128 @1/4 im2col + split GEMM [split f16x3, issued flops]"] = (_im2col_then_gemm, "mfma_f16", flc)

    # Winograd F(2x2,3x3): MFMA flops actually issued = 16 multiplies per 2x2 outputs (2.25x fewer than direct)
    def wino_case(name, b, hh, ww, cin, cout, dil, head):
        xw = torch.randn(b, hh, ww, cin, device=dev)
        u = ops.winograd_weights(torch.randn(cout, cin, 3, 3, device=dev) * 0.05)
        shw = torch.randn(cout, device=dev)
        w2w = torch.randn(64, device=dev) if head else None
        b2w = torch.zeros(1, device=dev) if head else None
        fl = 2.0 * 16 * cin * cout * b * ((hh + 1) // 2) * ((ww + 1) // 2)
        cases[name] = (lambda: ops.conv3x3_winograd(xw, u, shw, act=1, dilation=dil, w2=w2w, b2=b2w), "mfma", fl)
        us = ops.winograd_split_weights(torch.randn(cout, cin, 3, 3, device=dev) * 0.05)
        cases[name + " [split f16x3, issued flops]"] = (lambda: ops.conv3x3_winograd_split(xw, us, cout, shw, act=1, dilation=dil, w2=w2w, b2=b2w),
                                                         "mfma_f16", 3 * fl)
    wino_case("winograd 128->64 +1x1+sigmoid @full (segformer depth)", B, H, W, 128, 64, 1, True)
    wino_case("winograd 64->64 @1/4 (resnet l1)", B, H // 4, W // 4, 64, 64, 1, False)
    wino_case("winograd 256->256 @1/16 (resnet l3)", B, H // 16, W // 16, 256, 256, 1, False)
    wino_case("winograd 512->512 d2 @1/16 (resnet l4)", B, H // 16, W // 16, 512, 512, 2, False)
    wino_case("winograd 2048->256 @1/16 (deeplab depth)", B, H // 16, W // 16, 2048, 256, 1, False)

    def gemm_case(name, m, n, k, res):
        xg = torch.randn(m, k, device=dev); wg = torch.randn(n, k, device=dev) * 0.05; bg = torch.randn(n, device=dev)
        rg = torch.randn(m, n, device=dev) if res else None
        og = torch.empty(m, n, device=dev)
        wsg = ops.gemm_split_weights(wg)
        fl = 2.0 * m * n * k
        by = 4.0 * (m * k + m * n * (2 if res else 1))
        cases[f"gemm {name} M={m} N={n} K={k} [hipBLASLt f32]"] = (lambda: ops.gemm_bias_act(xg, wg, bg, 1, residual=rg, out=og, split=False), "mfma", fl)   # split=False: the library kernel, not the dispatcher's choice
        cases[f"gemm {name} M={m} N={n} K={k} [split f16x3, issued flops]"] = (lambda: ops.gemm_split_bias_act(xg, wsg, bg, 1, residual=rg, out=og), "mfma_f16", 3 * fl)
        cases[f"gemm {name} M={m} N={n} K={k} [split, as HBM bytes]"] = (lambda: ops.gemm_split_bias_act(xg, wsg, bg, 1, residual=rg, out=og), "hbm", by)
    px4, px8, px16 = B * (H // 4) * (W // 4), B * (H // 8) * (W // 8), B * (H // 16) * (W // 16)
    gemm_case("l1 conv1", px4, 64, 256, False)
    gemm_case("mit s2 fc2", px8, 64, 256, True)
    gemm_case("l1 conv3", px4, 256, 64, True)
    gemm_case("l2 conv1", px8, 128, 512, False)
    gemm_case("l2 conv3", px8, 512, 128, True)
    gemm_case("l3 conv1", px16, 256, 1024, False)
    gemm_case("l3 conv3", px16, 1024, 256, True)
    gemm_case("l4 conv1", px16, 512, 2048, False)
    gemm_case("l4 conv3", px16, 2048, 512, True)
    gemm_case("aspp 1x1", px16, 256, 2048, False)
    gemm_case("aspp project", px16, 256, 1280, False)
    gemm_case("decoder pw1", px4, 256, 304, False)

    def dual_case(name, b, ho, wo, k1, k2, n, st):
        # first bottleneck of a ResNet stage: relu([z | input at the stride] . [W3 | Wd]^T + b) in one launch; bytes: z, the gathered input rows, the output
        m = b * ho * wo
        z = torch.randn(m, k1, device=dev)
        x2 = torch.randn(b, ho * st, wo * st, k2, device=dev) if st > 1 else torch.randn(m, k2, device=dev)
        wsd = ops.gemm_split_weights(torch.randn(n, k1 + k2, device=dev) * 0.05); bd = torch.randn(n, device=dev)
        cases[f"gemm dual {name} M={m} N={n} K={k1}+{k2} [split f16x3, issued flops]"] = (lambda: ops.gemm_split_dual(z, x2, wsd, bd, 1, stride=st if st > 1 else 0), "mfma_f16", 3 * 2.0 * m * n * (k1 + k2))
        cases[f"gemm dual {name} M={m} N={n} K={k1}+{k2} [as HBM bytes]"] = (lambda: ops.gemm_split_dual(z, x2, wsd, bd, 1, stride=st if st > 1 else 0), "hbm", 4.0 * (m * (k1 + k2) + m * n))
    dual_case("l1 block0 tail", B, H // 4, W // 4, 64, 64, 256, 1)
    dual_case("l2 block0 tail (stride-2 gather)", B, H // 8, W // 8, 128, 256, 512, 2)
    dual_case("l4 block0 tail", B, H // 16, W // 16, 512, 1024, 2048, 1)
    gemm_case("decoder pw2", px4, 256, 256, False)

    def attn_case(name, nh, nq, nkv):
        qa = torch.randn(B, nq, nh * 32, device=dev); ka = torch.randn(B, nkv, nh * 32, device=dev); va = torch.randn(B, nkv, nh * 32, device=dev)
        fl = 4.0 * B * nh * nq * nkv * 32
        cases[name] = (lambda: ops.attention_d32(qa, ka, va, nh, 32 ** -0.5, split=False), "mfma", fl)
        cases[name + " [split f16x3, issued flops]"] = (lambda: ops.attention_d32(qa, ka, va, nh, 32 ** -0.5, split=True), "mfma_f16", 3 * fl)
        qh, kh, vh = (t.view(B, -1, nh, 32).transpose(1, 2) for t in (qa, ka, va))
        cases[name + " [torch SDPA]"] = (lambda: torch.nn.functional.scaled_dot_product_attention(qh, kh, vh, scale=32 ** -0.5), "mfma", fl)
    attn_case("attention d32 stage1 (131072 q x 2048 k, 1 head)", 1, (H // 4) * (W // 4), (H // 32) * (W // 32))
    attn_case("attention d32 stage3 (8192 q x 2048 k, 5 heads)", 5, (H // 16) * (W // 16), (H // 32) * (W // 32))

    a_lo = torch.randn(B, H // 16, W // 16, 256, device=dev); hi48 = torch.randn(B, H // 4, W // 4, 48, device=dev); w304 = torch.randn(9, 304, device=dev)
    cases["dwconv3x3_upcat (decoder: up x4 + cat + depthwise)"] = (lambda: ops.dwconv3x3_upcat(a_lo, hi48, w304), "hbm",
                                                                   4.0 * B * ((H // 16) * (W // 16) * 256 + (H // 4) * (W // 4) * (48 + 304)))

    only = [s for s in a.only.split(",") if s]
    rows = []
    for name, (fn, bound, work) in cases.items():
        if only and not any(o in name or o.replace("_", " ") in name for o in only):
            continue
        med, best = timeit(fn, a.iters)
        if bound == "hbm":
            ach, peak, unit = work / (med * 1e-3) / 1e9, HBM, "GB/s"
        elif bound == "mfma_f16":
            ach, peak, unit = work / (med * 1e-3) / 1e12, 2516.6, "TFLOP/s"
        else:
            ach, peak, unit = work / (med * 1e-3) / 1e12, MFMA, "TFLOP/s"
        rows.append({"kernel": name, "median_ms": round(med, 4), "min_ms": round(best, 4), "bound": bound,
                     "achieved": round(ach, 1), "unit": unit, "frac_of_peak": round(ach / peak, 3)})
        print(f"{name:45s} {med:8.3f} ms (min {best:7.3f})  {ach:9.1f} {unit:8s} {100 * ach / peak:5.1f}% of {bound} peak", flush=True)
    print(json.dumps({"batch": B, "height": H, "width": W, "rows": rows}))
    bad = [r["kernel"] for r in rows if r["frac_of_peak"] > 1.0]
    if bad:
        raise SystemExit(f"rows above their peak (wrong kernel measured or wrong work formula): {bad}")


if __name__ == "__main__":
    main()
