#!/usr/bin/env python3
"""Per-layer timing of the ResNet-50 (output stride 16) convolutions at bench shapes: MIOpen conv on
channels_last tensors vs the same 1x1 as a row-major GEMM (torch.mm / addmm / _addmm_activation)."""
import sys, time
from pathlib import Path
import torch
import torch.nn.functional as F
sys.path.insert(0, str(Path(__file__).resolve().parent.parent))

torch.backends.cudnn.benchmark = True
B, H, W = 8, 1024, 2048
dev = "cuda"


def t(fn, n=5):
    for _ in range(2):
        fn()
    torch.cuda.synchronize()
    s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    s.record()
    for _ in range(n):
        fn()
    e.record(); torch.cuda.synchronize()
    return s.elapsed_time(e) / n


# (name, cin, cout, k, stride, dilation, in_h, in_w, count)
L = []
def add(name, cin, cout, k, s, d, h, w, cnt=1): L.append((name, cin, cout, k, s, d, h, w, cnt))
add("stem 7x7/2", 3, 64, 7, 2, 1, H, W)
h, w = H // 4, W // 4
add("l1 conv1 first", 64, 64, 1, 1, 1, h, w); add("l1 conv1", 256, 64, 1, 1, 1, h, w, 2); add("l1 conv2", 64, 64, 3, 1, 1, h, w, 3)
add("l1 conv3", 64, 256, 1, 1, 1, h, w, 3); add("l1 down", 64, 256, 1, 1, 1, h, w)
add("l2 conv1 first", 256, 128, 1, 1, 1, h, w); add("l2 conv2 first /2", 128, 128, 3, 2, 1, h, w); add("l2 down /2", 256, 512, 1, 2, 1, h, w)
h, w = H // 8, W // 8
add("l2 conv1", 512, 128, 1, 1, 1, h, w, 3); add("l2 conv2", 128, 128, 3, 1, 1, h, w, 3); add("l2 conv3", 128, 512, 1, 1, 1, h, w, 4)
add("l3 conv1 first", 512, 256, 1, 1, 1, h, w); add("l3 conv2 first /2", 256, 256, 3, 2, 1, h, w); add("l3 down /2", 512, 1024, 1, 2, 1, h, w)
h, w = H // 16, W // 16
add("l3 conv1", 1024, 256, 1, 1, 1, h, w, 5); add("l3 conv2", 256, 256, 3, 1, 1, h, w, 5); add("l3 conv3", 256, 1024, 1, 1, 1, h, w, 6)
add("l4 conv1 first", 1024, 512, 1, 1, 1, h, w); add("l4 conv2 d2", 512, 512, 3, 1, 2, h, w, 3); add("l4 conv3", 512, 2048, 1, 1, 1, h, w, 3)
add("l4 down", 1024, 2048, 1, 1, 1, h, w); add("l4 conv1", 2048, 512, 1, 1, 1, h, w, 2)

tot_conv = tot_best = 0.0
for (name, cin, cout, k, s, d, ih, iw, cnt) in L:
    x = torch.randn(B, cin, ih, iw, device=dev).contiguous(memory_format=torch.channels_last)
    wt = torch.randn(cout, cin, k, k, device=dev).contiguous(memory_format=torch.channels_last)
    pad = d * (k // 2)
    oh, ow = (ih + 2 * pad - d * (k - 1) - 1) // s + 1, (iw + 2 * pad - d * (k - 1) - 1) // s + 1
    fl = 2.0 * B * oh * ow * cout * cin * k * k
    tc = t(lambda: F.conv2d(x, wt, None, s, pad, d))
    line = f"{name:20s} x{cnt} {cin:5d}->{cout:5d} k{k} s{s} d{d} {oh}x{ow}: conv {tc:7.3f} ms {fl / tc / 1e9:6.1f} TF"
    best = tc
    if k == 1 and s == 1:
        x2 = x.permute(0, 2, 3, 1).reshape(-1, cin); w2 = wt.view(cout, cin).t().contiguous(); w2t = wt.view(cout, cin)
        bias = torch.randn(cout, device=dev)
        tm = t(lambda: torch.mm(x2, w2t.t()))
        ta = t(lambda: torch._addmm_activation(bias, x2, w2t.t(), use_gelu=False))
        line += f" | mm {tm:7.3f} ms {fl / tm / 1e9:6.1f} TF | addmm+relu {ta:7.3f} ms"
        best = min(tc, tm)
    print(line, flush=True)
    tot_conv += tc * cnt; tot_best += best * cnt
    del x, wt
print(f"total conv {tot_conv:.2f} ms; best-of {tot_best:.2f} ms")
