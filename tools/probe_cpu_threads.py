import sys, time, copy, os, torch
sys.path.insert(0, '.')
import adverse_weather_semantic_segmentation_robustness_benchmark_amd as P
torch.manual_seed(0)
m = P.EnsembleModel(pretrained=False).eval()
for mod in m.modules(): mod.fused_eval = False
x = torch.randn(1, 3, 256, 512)
print('cpus', os.cpu_count(), flush=True)
for th in (16, 32, 64):
    torch.set_num_threads(th)
    with torch.no_grad():
        m(x[:, :, :64, :64])
        t0 = time.perf_counter(); m(x); dt = time.perf_counter() - t0
    print('threads', th, 'fwd 256x512: %.2f s' % dt, flush=True)
