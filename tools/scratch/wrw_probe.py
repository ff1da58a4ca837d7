"""Probe: which MIOpen solver computes the 3x3 weight gradients of the training step, and what do the alternatives cost?
Runs each shape's backward (weight gradient only) under the environment given on the command line, HIP-event timed."""
import os, sys, time
import torch
import torch.nn.functional as F

shapes = [("l1 conv2 64->64 @256x512", 8, 64, 64, 256, 512, 1), ("l2 conv2 128->128 @128x256", 8, 128, 128, 128, 256, 1),
          ("l3 conv2 256->256 @64x128", 8, 256, 256, 64, 128, 1), ("l4 conv2 512->512 d2 @64x128", 8, 512, 512, 64, 128, 2),
          ("depth 128->64 @1024x2048", 8, 128, 64, 1024, 2048, 1), ("dl depth 2048->256 @64x128", 8, 2048, 256, 64, 128, 1)]
torch.backends.cudnn.benchmark = False
for name, b, ci, co, h, w, d in shapes:
    x = torch.randn(b, ci, h, w, device="cuda")
    wt = torch.randn(co, ci, 3, 3, device="cuda", requires_grad=True)
    y = F.conv2d(x, wt, None, 1, d, d)
    g = torch.randn_like(y)
    for it in range(3):
        torch.cuda.synchronize(); t0 = time.perf_counter()
        (dw,) = torch.autograd.grad(y, wt, g, retain_graph=True)
        torch.cuda.synchronize(); dt = time.perf_counter() - t0
    fl = 2.0 * b * h * w * ci * co * 9
    print(f"{name}: weight gradient {dt * 1e3:.2f} ms = {fl / dt / 1e12:.1f} TFLOP/s", flush=True)
    del x, y, g, wt
