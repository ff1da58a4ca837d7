#!/usr/bin/env python3
"""Run every hand-written operator (and the library calls the eval forward makes) twice on the same inputs and compare
the outputs bit for bit.  A run-to-run difference means atomics / a race / an uninitialised read inside that operator."""
import sys
from pathlib import Path

import torch

ROOT = Path(__file__).resolve().parent.parent
sys.path.insert(0, str(ROOT))
from adverse_weather_semantic_segmentation_robustness_benchmark_amd import ops  # noqa: E402

dev = "cuda"
bad = 0


def check(name, fn, reps=4):
    global bad
    ref = fn()
    ref = [r.clone() for r in (ref if isinstance(ref, (tuple, list)) else [ref])]
    for i in range(reps):
        # disturb the allocator / caches between runs
        junk = torch.randn(1 << 22, device=dev)
        out = fn()
        out = out if isinstance(out, (tuple, list)) else [out]
        for a, b in zip(ref, out):
            if not torch.equal(a, b):
                d = (a.float() - b.float()).abs().max().item()
                print(f"NONDETERMINISTIC {name}: rep {i}: max |d| {d:.3e} ({int((a != b).sum())} elements, magnitude {a.abs().max().item():.3g})")
                bad += 1
                return
        del junk
    print(f"ok  {name}")


def main():
    torch.manual_seed(0)
    for scale in (1.0, 3e4):
        for (m, n, k) in [(4096, 512, 2048), (4096, 2048, 512), (65536, 2048, 512), (4096, 256, 1024), (1000, 19 * 8, 304), (65536, 256, 64)]:
            x = torch.randn(m, k, device=dev) * scale
            w = torch.randn(n, k, device=dev) * 0.05
            b = torch.randn(n, device=dev)
            r = torch.randn(m, n, device=dev)
            ws = ops.gemm_split_weights(w)
            check(f"gemm_split M={m} N={n} K={k} x*{scale:g}", lambda: ops.gemm_split_bias_act(x, ws, b, 1, residual=r))
            if scale == 1.0:
                check(f"hipBLASLt gemm_bias_act M={m} N={n} K={k}", lambda: ops.gemm_bias_act(x, w, b, 1, residual=r, split=False))
    for (bb, h, w_, cin, cout, dil, head) in [(2, 64, 128, 128, 64, 1, True), (8, 16, 32, 512, 512, 2, False), (8, 16, 32, 2048, 256, 1, False),
                                              (2, 64, 128, 64, 64, 1, False), (8, 16, 32, 256, 256, 1, False), (8, 17, 33, 512, 512, 2, False)]:
        x = torch.randn(bb, h, w_, cin, device=dev)
        u = ops.winograd_weights(torch.randn(cout, cin, 3, 3, device=dev) * 0.05)
        sh = torch.randn(cout, device=dev)
        w2 = torch.randn(64, device=dev) if head else None
        b2 = torch.zeros(1, device=dev) if head else None
        check(f"winograd {cin}->{cout} d{dil} {h}x{w_} head={head}", lambda: ops.conv3x3_winograd(x, u, sh, act=1, dilation=dil, w2=w2, b2=b2))
    g9 = torch.randn(2, 8, 16, 9, 256, device=dev)
    sf = torch.randn(256, device=dev) * 0.1
    w2 = torch.randn(19, 256, device=dev) * 0.05
    b2 = torch.zeros(19, device=dev)
    check("segformer_head_fused", lambda: ops.segformer_head_fused(g9, None, sf, w2, b2, 256, 512))
    g9d = torch.randn(2, 8, 16, 9, 128, device=dev)
    check("upconv3x3_bn_relu", lambda: ops.upconv3x3_bn_relu(g9d, None, sf[:128].contiguous(), 256, 512, True))
    tok = torch.randn(8 * 8 * 16, 256, device=dev)
    w1r = torch.randn(256, 9 * 256, device=dev)
    check("torch matmul g9 [1024,256]x[256,2304]", lambda: tok @ w1r)
    tokb = torch.randn(8 * 32 * 64, 256, device=dev)
    check("torch matmul g9 [16384,256]x[256,2304]", lambda: tokb @ w1r)
    q = torch.randn(2, 4096, 64, device=dev); k = torch.randn(2, 256, 64, device=dev); v = torch.randn(2, 256, 64, device=dev)
    check("attention split", lambda: ops.attention_d32(q, k, v, 2, 32 ** -0.5, split=True))
    check("attention f32", lambda: ops.attention_d32(q, k, v, 2, 32 ** -0.5, split=False))
    xa = torch.randn(2, 16, 32, 2048, device=dev); wdw = torch.randn(3, 9, 2048, device=dev)
    check("aspp_depthwise3", lambda: ops.aspp_depthwise3(xa, wdw, (12, 24, 36)))
    xd = torch.randn(2, 64, 128, 128, device=dev); w9 = torch.randn(9, 128, device=dev); bb_ = torch.randn(128, device=dev)
    check("dwconv3x3_nhwc", lambda: ops.dwconv3x3_nhwc(xd, w9, bb_, 2))
    a_lo = torch.randn(2, 16, 32, 256, device=dev); hi48 = torch.randn(2, 64, 128, 48, device=dev); w304 = torch.randn(9, 304, device=dev)
    check("dwconv3x3_upcat", lambda: ops.dwconv3x3_upcat(a_lo, hi48, w304))
    lin = torch.nn.Linear(32, 32).cuda()
    t = torch.randn(8, 64, 128, 32, device=dev)
    check("F.linear 32->32 on 65536 rows", lambda: torch.nn.functional.linear(t, lin.weight, lin.bias))
    conv = torch.nn.Conv2d(3, 64, 7, 2, 3, bias=False).cuda()
    xi = torch.randn(8, 3, 256, 512, device=dev).contiguous(memory_format=torch.channels_last)
    check("MIOpen stem conv 7x7 s2", lambda: conv(xi))
    conv2 = torch.nn.Conv2d(128, 128, 3, 2, 1, bias=False).cuda().to(memory_format=torch.channels_last)
    xj = torch.randn(8, 128, 64, 128, device=dev).contiguous(memory_format=torch.channels_last)
    check("MIOpen conv 3x3 s2 128ch", lambda: conv2(xj))
    print("all deterministic" if bad == 0 else f"{bad} nondeterministic operators")
    raise SystemExit(1 if bad else 0)


if __name__ == "__main__":
    main()
