#!/usr/bin/env python3
"""bench.py — images/s of the north-star hot path on MI355X.

    python bench.py --gpus 1 --steps 5 --warmup 2
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 \
           --master-port P bench.py --gpus N --steps K --warmup W

One "step" = one pass of the hot path over one batch of B synthetic 1024x2048 frames that are
already resident in HBM as uint8 HWC (+ uint8 labels):

    weather corruption (round-robin clean/fog/rain/snow/night, in-kernel Philox noise)
      -> fused Normalize/ToTensor -> SegFormer-B0 + DeepLabV3+-R50 ensemble forward (fp32)
      -> combine / temperature / argmax / 19x19 confusion (overall + per condition) in one pass

After the K timed steps the int64 counters are SUM-all-reduced over ranks (RCCL) and the mIoU /
degradation ratios are finished on the host — inside the timed region.  Weak scaling: every rank
runs its own B frames per step; value = N*B*K / max-over-ranks time.

The line also carries `roofline` for the dominant hand-written kernel (HIP events recorded around
every launch of it on the launching stream, inside the timed region) and `cpu_baseline` (the CPU
oracle path timed on this box's host cores on a bounded sample; rank 0, N=1 only).
"""
from __future__ import annotations

import argparse
import json
import os
import sys
import time
from pathlib import Path

ROOT = Path(__file__).resolve().parent
if str(ROOT) not in sys.path:
    sys.path.insert(0, str(ROOT))

import numpy as np
import torch

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
        self.pairs.setdefault(name, []).append((s, e, launch_work(name, args)))
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
    if name == "awseg_attention_d32":
        # (q, k, v, out, batch, heads, n_queries, n_keys, ...): QK^T and PV, head_dim 32
        b, heads, nq, nkv = args[4:8]
        return "mfma", 4.0 * b * heads * nq * nkv * 32
    if name == "awseg_attention_d32_split":
        # the same products, each issued as THREE f16 MFMA products (split operands): priced as issued, against the f16 peak
        b, heads, nq, nkv = args[4:8]
        return "mfma_f16", 3 * 4.0 * b * heads * nq * nkv * 32
    if name == "awseg_gemm_split_bias_act":
        # (x, w_split, bias, residual, act, out, m, n, k): three f16 MFMA products per float32-grade product, priced as issued
        m, n, k = args[6:9]
        return "mfma_f16", 3 * 2.0 * m * n * k
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
    if name == "awseg_combine_argmax_confusion":
        return "hbm", (2 * C * 4 + 1) * px * B         # two member logit maps + labels in; counters only out
    if name == "awseg_aspp_depthwise3":
        h, w = H // 16, W // 16
        return "hbm", (2048 * 4 + 3 * 2048 * 4) * h * w * B
    if name == "awseg_segformer_head_fused":           # executed MFMA flops: GEMM1 K=12 + GEMM2 N padded to 32
        return "mfma", 2.0 * (12 * 256 + 256 * 32) * px * B
    if name == "awseg_upconv3x3_bn_relu":              # 24 MFMAs per 32 px, but 512 B/px of output: the write is the roofline
        return "hbm", 128 * 4 * px * B
    return "hbm", 0


# device-function names of the C-ABI launchers' dominant kernels (for the PMC traffic lookup)
DEVICE_KERNEL = {"awseg_conv3x3_winograd_nhwc": "conv3x3_wino_kernel<1>", "awseg_segformer_head_fused": "head_mfma_classify_kernel<8>", "awseg_combine_argmax_confusion": "combine_argmax_confusion_kernel<0",
                 "awseg_upconv3x3_bn_relu": "head_mfma_kernel<4, false", "awseg_aspp_depthwise3": "aspp_dw3_kernel"}


def pmc_traffic(name):
    """HBM bytes per launch of `name` from the committed PMC passes (profiles/: separate rocprofv3
    --pmc FETCH_SIZE and --pmc WRITE_SIZE runs of tools/kernel_bench.py at this same problem size;
    FETCH_SIZE doubled as MI355X_MICROARCH.md §HBM prescribes for wide coalesced reads).  None if the
    profile is not there — bench.py itself does not collect counters."""
    import csv
    path = ROOT / "profiles" / "r01_kernel_bench_v4_stats_and_traffic.csv"
    key = DEVICE_KERNEL.get(name)
    if not key or not path.exists():
        return None
    with open(path) as f:
        rows = list(csv.DictReader(l for l in f if not l.startswith("#")))
    for r in rows:
        if key in r["kernel"] and r["FETCH_SIZE_KB"] and r["WRITE_SIZE_KB"]:
            return int((2.0 * float(r["FETCH_SIZE_KB"]) + float(r["WRITE_SIZE_KB"])) * 1024)
    return None


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
                      f"({cores} torch threads); oracle argmax+confusion on 1 frame {H}x{W} ({t_metric:.2f} s, 1 thread)"}


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=20)
    ap.add_argument("--warmup", type=int, default=5, help="untimed steps; the first ~4 steps after start-up run ~10 %% slower (allocator growth, clocks)")
    ap.add_argument("--batch", type=int, default=8, help="frames per GPU per step (README.md:125 batch size)")
    ap.add_argument("--height", type=int, default=1024)
    ap.add_argument("--width", type=int, default=2048)
    ap.add_argument("--no-depth", action="store_true", help="build the ensemble with include_depth=False")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--fp32-mfma", action="store_true",
                    help="attention and 1x1 convolutions on the float32-input MFMA kernels (own fp32 attention, hipBLASLt fp32) "
                         "instead of the split-operand f16-MFMA ones (DESIGN.md 5b); same as AWSEG_ATTN_SPLIT=0 AWSEG_GEMM_SPLIT=0")
    ap.add_argument("--conv-search", type=int, default=int(os.environ.get("AWSEG_CONV_SEARCH", "0")),
                    help="1: let MIOpen time its solvers per convolution shape during warm-up (torch.backends.cudnn.benchmark)")
    args = ap.parse_args()
    torch.backends.cudnn.benchmark = bool(args.conv_search)

    from adverse_weather_semantic_segmentation_robustness_benchmark_amd import ops, parallel
    if args.fp32_mfma:
        ops.ATTENTION_SPLIT = False
        ops.GEMM_SPLIT = False
    from adverse_weather_semantic_segmentation_robustness_benchmark_amd.models.model import EnsembleModel
    from adverse_weather_semantic_segmentation_robustness_benchmark_amd.evaluation.metrics import RobustnessMetrics
    from adverse_weather_semantic_segmentation_robustness_benchmark_amd.data.preprocessing import WeatherDegradationTransforms

    rank, local, world = parallel.init_from_env()
    assert world == max(args.gpus, 1), f"--gpus {args.gpus} but WORLD_SIZE={world}"
    dev = torch.device("cuda", local % max(torch.cuda.device_count(), 1))     # ranks beyond the device count share GPUs (rehearsal only)
    torch.cuda.set_device(dev)
    B, H, W, C = args.batch, args.height, args.width, 19
    conds_all = ["clean", "fog", "rain", "snow", "night"]

    torch.manual_seed(42)
    np.random.seed(42 + rank)
    model = EnsembleModel(num_classes=C, include_depth=not args.no_depth, pretrained=False).to(dev).eval()
    metrics = RobustnessMetrics(num_classes=C, weather_conditions=conds_all)
    acc = metrics.new_accumulator(dev)
    tf = WeatherDegradationTransforms(seed=1234 + rank, rng="philox", device=dev)
    gen = torch.Generator(device=dev)
    gen.manual_seed(42 + rank)
    raw = torch.randint(0, 255, (B, H, W, 3), dtype=torch.uint8, device=dev, generator=gen)      # loader.py:206
    labels = torch.randint(0, C, (B, H, W), dtype=torch.uint8, device=dev, generator=gen)        # loader.py:231
    image = torch.empty(B, 3, H, W, dtype=torch.float32, device=dev)
    from adverse_weather_semantic_segmentation_robustness_benchmark_amd import _native
    _native.launch_hook = CLOCK.hook
    info = {}

    def step(i):
        start = (rank * B + i * B * world) % len(conds_all)
        conds = [conds_all[(start + k) % len(conds_all)] for k in range(B)]
        info.update({"awseg_fog_fused": conds.count("fog"), "awseg_night_apply": conds.count("night"),
                     "awseg_rain_apply": conds.count("rain"), "awseg_snow_apply": conds.count("snow"),
                     "awseg_normalize": conds.count("clean")})
        tf.apply_batch(raw, conds, norm_out=image)
        model.forward_eval(image, labels, acc.counts, acc.oob, acc.cond_ids(conds), want_logits=False, want_pred=False)

    with torch.no_grad():
        def finish():
            """Counters -> mIoUs (inside the timed region: the job is not done until the metric exists)."""
            acc.all_reduce()
            acc.check()
            res = {"overall_miou": acc.miou(0)}
            for k, name in enumerate(conds_all):
                if acc.present(1 + k):
                    res[f"miou_{name}"] = acc.miou(1 + k)
            return res

        for i in range(args.warmup):
            step(i)
        finish()                                  # warm the host-side finalisation too (first-use costs of the CPU ops)
        acc.counts.zero_()
        torch.cuda.synchronize()
        parallel.barrier()
        CLOCK.enabled = True
        t0 = time.perf_counter()
        for i in range(args.steps):
            step(args.warmup + i)
        results = finish()
        torch.cuda.synchronize()
        parallel.barrier()
        dt = time.perf_counter() - t0
        CLOCK.enabled = False

    tmax = torch.tensor([dt], dtype=torch.float64, device=dev)
    if parallel.is_dist():
        torch.distributed.all_reduce(tmax, op=torch.distributed.ReduceOp.MAX)
    dt = float(tmax.item())
    total_images = B * args.steps * world

    # ---- roofline of the dominant hand-written kernel (rank 0's launches) ----------------------
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
        kernels.append({"kernel": name, "launches": count, "avg_ms": round(avg_ms, 4), "bound": bound,
                        "achieved": round(achieved, 2), "peak": peak, "unit": unit, "frac": round(achieved / peak, 4),
                        "time_share_of_step": round(count * avg_ms / (dt * 1e3), 4)})
    kernels.sort(key=lambda k: -k["launches"] * k["avg_ms"])
    roofline = None
    if kernels:
        k0 = kernels[0]
        roofline = {"kernel": k0["kernel"], "bound": k0["bound"], "achieved": k0["achieved"], "peak": k0["peak"],
                    "unit": k0["unit"], "frac": k0["frac"], "traffic": pmc_traffic(k0["kernel"]),
                    "traffic_source": "profiles/r01_kernel_bench_v4_stats_and_traffic.csv (separate --pmc FETCH_SIZE / WRITE_SIZE passes of tools/kernel_bench.py; for the Winograd kernel: the full-resolution 128->64 depth-head launch, the largest of its 17 launches per step)"}

    if rank == 0:
        cpu = None
        if world == 1 and not args.no_cpu_baseline:
            try:
                cpu = cpu_baseline(model, H, W, C)
            except Exception as e:  # noqa: BLE001
                cpu = {"value": None, "unit": "images/s", "cores": os.cpu_count(), "kind": "port", "sample": f"failed: {e!r}"}
        line = {
            "metric": "images/sec (1024x2048, 5 weather conds, ensemble eval)",
            "value": round(total_images / dt, 3), "unit": "images/s", "n_gpus": world, "steps": args.steps,
            "warmup": args.warmup, "ms_per_step": round(dt / args.steps * 1e3, 3), "higher_is_better": True,
            "scaling": "weak", "vs_baseline": None, "dtype": "f32", "data": "synthetic",
            "config": {"workload": f"ensemble_eval_{H}x{W}_5cond (BASELINE.json configs[1]: SegFormer-B0 + DeepLabV3+-R50, "
                                   "all 5 weather conditions round-robin)", "per_gpu_batch": B, "global_batch": B * world,
                       "include_depth": not args.no_depth, "weather_rng": "philox (in-kernel)", "ensemble_logits_materialised": False,
                       "attention": ("split-operand f16 MFMA (22-bit operands, f32 accumulate)" if ops.ATTENTION_SPLIT else "f32 MFMA"),
                       "gemm_1x1": ("split-operand f16 MFMA (22-bit operands, f32 accumulate)" if ops.GEMM_SPLIT else "hipBLASLt f32"),
                       "weights": "random init (no checkpoints offline)", "parallelism": f"batch-sharded x{world}, one int64 counter all-reduce"},
            "roofline": roofline, "cpu_baseline": cpu, "kernels": kernels,
            "miou": {k: round(v, 6) for k, v in results.items()},
        }
        print(json.dumps(line))
    if parallel.is_dist():
        torch.distributed.destroy_process_group()


if __name__ == "__main__":
    main()
