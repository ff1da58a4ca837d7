"""Every GEMM of one eval forward at the bench shape (8 x 1024 x 2048): (M, N, K), residual, which path takes it (split-operand
kernel or hipBLASLt) and its time (HIP events, second call).  python tools/scratch/gemm_shapes.py"""
import sys, collections, torch
sys.path.insert(0, ".")
import adverse_weather_semantic_segmentation_robustness_benchmark_amd as P
from adverse_weather_semantic_segmentation_robustness_benchmark_amd import ops
torch.manual_seed(0)
m = P.EnsembleModel(num_classes=19, include_depth=True, pretrained=False).cuda().eval()
x = torch.randn(8, 3, 1024, 2048, device="cuda")
log = collections.OrderedDict()
orig = ops.gemm_bias_act
def wrapped(xx, w, bias, act=0, residual=None, out=None, w_split=None, split=None):
    mm, k = xx.shape; n = w.shape[0]
    s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    s.record(); r = orig(xx, w, bias, act, residual=residual, out=out, w_split=w_split, split=split); e.record()
    log.setdefault((mm, n, k, residual is not None, bool(ops.gemm_wants_split(mm, n, k))), []).append((s, e))
    return r
ops.gemm_bias_act = wrapped
import adverse_weather_semantic_segmentation_robustness_benchmark_amd.models.fused as F_
with torch.no_grad():
    for _ in range(2):
        log.clear(); m(x)
torch.cuda.synchronize()
tot = {True: 0.0, False: 0.0}
for (mm, n, k, res, split), evs in log.items():
    t = sum(s.elapsed_time(e) for s, e in evs)
    tot[split] += t
    gb = (mm * k + mm * n * (2 if res else 1)) * 4 / 1e9
    print(f"M={mm:8d} N={n:5d} K={k:5d} res={int(res)} {'split' if split else 'hipBLASLt'}: {len(evs)} calls, {t / len(evs) * 1e3:7.1f} us each, {gb / (t / len(evs)) * 1e3:7.0f} GB/s")
print("split total", round(tot[True], 3), "ms; hipBLASLt total", round(tot[False], 3), "ms")
