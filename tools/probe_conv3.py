#!/usr/bin/env python3
"""3x3 convolutions of the step in NCHW vs channels_last memory, MIOpen solver search on/off."""
import sys, os
import torch
import torch.nn.functional as F
bench = int(sys.argv[1]) if len(sys.argv) > 1 else 1
torch.backends.cudnn.benchmark = bool(bench)
dev = "cuda"
def t(fn, n=3):
    for _ in range(2): fn()
    torch.cuda.synchronize()
    s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    s.record()
    for _ in range(n): fn()
    e.record(); torch.cuda.synchronize()
    return s.elapsed_time(e) / n
shapes = [("segformer depth conv2", 8, 128, 64, 1024, 2048, 1), ("l1 conv2", 8, 64, 64, 256, 512, 1), ("l2 conv2", 8, 128, 128, 128, 256, 1),
          ("l3 conv2", 8, 256, 256, 64, 128, 1), ("l4 conv2 d2", 8, 512, 512, 64, 128, 2), ("deeplab depth conv1", 8, 2048, 256, 64, 128, 1)]
for (name, B, cin, cout, h, w, d) in shapes:
    fl = 2.0 * B * h * w * cout * cin * 9
    for fmt in ("nchw", "nhwc"):
        x = torch.randn(B, cin, h, w, device=dev); wt = torch.randn(cout, cin, 3, 3, device=dev)
        if fmt == "nhwc":
            x = x.contiguous(memory_format=torch.channels_last); wt = wt.contiguous(memory_format=torch.channels_last)
        ms = t(lambda: F.conv2d(x, wt, None, 1, d, d))
        print(f"search={bench} {name:24s} {fmt}: {ms:8.3f} ms  {fl / ms / 1e9:7.1f} TF(direct-equivalent)", flush=True)
        del x, wt
