"""Per-step wall time of the bench step from a cold start (what settles during the first steps?)."""
import sys, time, torch
from pathlib import Path
sys.path.insert(0, str(Path(__file__).resolve().parent.parent))
import bench  # noqa
from adverse_weather_semantic_segmentation_robustness_benchmark_amd.models.model import EnsembleModel
from adverse_weather_semantic_segmentation_robustness_benchmark_amd.data.preprocessing import WeatherDegradationTransforms
from adverse_weather_semantic_segmentation_robustness_benchmark_amd.evaluation.metrics import RobustnessMetrics
B, H, W = 8, 1024, 2048
dev = torch.device("cuda")
model = EnsembleModel(include_depth=True).to(dev).eval()
tf = WeatherDegradationTransforms(rng="philox", device=dev)
acc = RobustnessMetrics().new_accumulator(dev)
raw = torch.randint(0, 255, (B, H, W, 3), dtype=torch.uint8, device=dev)
labels = torch.randint(0, 19, (B, H, W), dtype=torch.uint8, device=dev)
image = torch.empty(B, 3, H, W, device=dev)
conds_all = ["clean", "fog", "rain", "snow", "night"]
def step(i):
    conds = [conds_all[(i * B + k) % 5] for k in range(B)]
    with torch.no_grad():
        tf.apply_batch(raw, conds, norm_out=image)
        model.forward_eval(image, labels, acc.counts, acc.oob, acc.cond_ids(conds), want_logits=False, want_pred=False)
step(0); torch.cuda.synchronize()
ev = [torch.cuda.Event(enable_timing=True) for _ in range(17)]
host = []
ev[0].record()
for i in range(16):
    h0 = time.perf_counter(); step(1 + i); host.append(time.perf_counter() - h0)
    ev[i + 1].record()
torch.cuda.synchronize()
for i in range(16):
    print(f"step {i + 1}: gpu {ev[i].elapsed_time(ev[i + 1]):8.2f} ms   host enqueue {1e3 * host[i]:7.2f} ms", flush=True)
