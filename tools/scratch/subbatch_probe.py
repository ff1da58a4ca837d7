"""Does the 256 MB memory-side cache pay for sub-batching?  Times the ensemble forward (and the two members) at batch 8 against
4 x batch 2 and 2 x batch 4 and 8 x batch 1 (same frames), HIP events, one stream."""
import sys, torch
sys.path.insert(0, ".")
import adverse_weather_semantic_segmentation_robustness_benchmark_amd as P
from adverse_weather_semantic_segmentation_robustness_benchmark_amd import ops
torch.manual_seed(0)
ops.TWO_STREAMS = False
m = P.EnsembleModel(num_classes=19, include_depth=True, pretrained=False).cuda().eval()
x = torch.randn(8, 3, 1024, 2048, device="cuda")
def timed(f, n=5):
    f(); f(); s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    s.record()
    for _ in range(n): f()
    e.record(); torch.cuda.synchronize()
    return s.elapsed_time(e) / n
with torch.no_grad():
    for name, mod in (("ensemble", m), ("segformer", m.segformer), ("deeplabv3plus", m.deeplabv3plus)):
        for sb in (8, 4, 2, 1):
            t = timed(lambda: [mod(x[i:i + sb]) for i in range(0, 8, sb)])
            print(f"{name:14s} sub-batch {sb}: {t:7.2f} ms per 8 frames", flush=True)
