import sys, time, torch
sys.path.insert(0, '.')
import adverse_weather_semantic_segmentation_robustness_benchmark_amd as P
from adverse_weather_semantic_segmentation_robustness_benchmark_amd.data.loader import CityscapesKITTIDataset, create_dataloader
from adverse_weather_semantic_segmentation_robustness_benchmark_amd.evaluation.harness import evaluate_model
torch.manual_seed(0)
cfg = P.Config({"data": {"weather_conditions": ["clean", "fog", "rain", "snow", "night"]}, "evaluation": {"num_bins": 15}})
model = P.EnsembleModel(pretrained=False).cuda().eval()
for n in (16, 32):
    ds = CityscapesKITTIDataset(split="test", image_size=(1024, 2048), weather_schedule="round_robin", num_samples=n)
    loader = create_dataloader(ds, batch_size=8, shuffle=False)
    torch.cuda.synchronize(); t0 = time.time()
    res = evaluate_model(model, loader, P.RobustnessMetrics(19), torch.device("cuda"), cfg)
    torch.cuda.synchronize(); dt = time.time() - t0
    print(n, "frames: %.2f s -> %.1f img/s" % (dt, n / dt), {k: round(float(v), 4) for k, v in list(res.items())[:3]})
# breakdown of the extra pieces on one batch
x = torch.randn(8, 19, 1024, 2048, device="cuda"); y = torch.randn(8, 19, 1024, 2048, device="cuda")
lab = torch.randint(0, 19, (8, 1024, 2048), device="cuda", dtype=torch.uint8)
from adverse_weather_semantic_segmentation_robustness_benchmark_amd.evaluation.harness import EvalState
st = EvalState(P.RobustnessMetrics(19), ["clean", "fog", "rain", "snow", "night"], "cuda", 15, True)
def t(fn, n=3):
    fn(); torch.cuda.synchronize(); t0 = time.time()
    for _ in range(n): fn()
    torch.cuda.synchronize(); return (time.time() - t0) / n * 1e3
print("auroc update ms", t(lambda: st.update_auroc(x, y, lab)))
from adverse_weather_semantic_segmentation_robustness_benchmark_amd import ops
print("ece ms", t(lambda: ops.ece_accumulate(x, lab, st.ece, st.edges, None)))
