import sys, torch
sys.path.insert(0, "/root/repo")
from adverse_weather_semantic_segmentation_robustness_benchmark_amd import ops
def t(fn, n=20):
    for _ in range(3): fn()
    torch.cuda.synchronize()
    s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    s.record()
    for _ in range(n): fn()
    e.record(); torch.cuda.synchronize()
    return s.elapsed_time(e) / n
x = torch.randn(8, 256, 512, 128, device="cuda"); w9 = torch.randn(9, 128, device="cuda"); b = torch.randn(128, device="cuda")
for act in (0, 1, 2):
    print("dwconv 128ch @1/4 act", act, round(t(lambda: ops.dwconv3x3_nhwc(x, w9, b, act)), 4), "ms")
y = torch.empty_like(x)
print("copy", round(t(lambda: y.copy_(x)), 4), "ms")
