#!/usr/bin/env python3
"""Does the row stride of x / w (K * 4 bytes) matter to the split-operand GEMM?  Times M=65536, N=512 at K = 2048 (rows
8 KB apart: a power of two) and K = 2056 / 2080 (not), per FLOP."""
import sys
from pathlib import Path
sys.path.insert(0, str(Path(__file__).resolve().parent.parent))
import torch
from adverse_weather_semantic_segmentation_robustness_benchmark_amd import ops

dev = "cuda"
for (m, n, k) in [(65536, 512, 2048), (65536, 512, 2056), (65536, 512, 2080), (65536, 256, 1024), (65536, 256, 1032), (65536, 256, 1056)]:
    x = torch.randn(m, k, device=dev); w = torch.randn(n, k, device=dev) * 0.05; b = torch.zeros(n, device=dev)
    ws = ops.gemm_split_weights(w); o = torch.empty(m, n, device=dev)
    for _ in range(3):
        ops.gemm_split_bias_act(x, ws, b, 1, out=o)
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(20):
        ops.gemm_split_bias_act(x, ws, b, 1, out=o)
    e1.record(); torch.cuda.synchronize()
    ms = e0.elapsed_time(e1) / 20
    print(f"M={m} N={n} K={k}: {ms:.3f} ms  {3 * 2.0 * m * n * k / ms / 1e9:.0f} TFLOP/s issued  ({ms / k * 2048:.3f} ms per 2048 of K)", flush=True)
