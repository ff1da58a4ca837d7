"""What is left to the libraries in one eval step at the bench shape: (1) the GEMMs hipBLASLt still takes — each timed on the
library and forced onto the split-operand kernel, with both results' error against float64; (2) the ATen kernels of the step
(copies, reductions, elementwise) with their operand shapes.  python tools/scratch/leftovers_probe.py"""
import sys, collections, torch
sys.path.insert(0, ".")
import adverse_weather_semantic_segmentation_robustness_benchmark_amd as P
from adverse_weather_semantic_segmentation_robustness_benchmark_amd import ops
torch.manual_seed(0)
ops.TWO_STREAMS = False
m = P.EnsembleModel(num_classes=19, include_depth=True, pretrained=False).cuda().eval()
x = torch.randn(8, 3, 1024, 2048, device="cuda")
shapes = collections.OrderedDict()
orig = ops.gemm_bias_act
def wrapped(xx, w, bias, act=0, residual=None, out=None, w_split=None, split=None):
    mm, k = xx.shape; n = w.shape[0]
    if split is None and not ops.gemm_wants_bf16(mm, n, k) and not ops.gemm_wants_split(mm, n, k):
        shapes.setdefault((mm, n, k, residual is not None, act), 0)
        shapes[(mm, n, k, residual is not None, act)] += 1
    return orig(xx, w, bias, act, residual=residual, out=out, w_split=w_split, split=split)
ops.gemm_bias_act = wrapped
with torch.no_grad():
    m(x); shapes.clear(); m(x)
torch.cuda.synchronize()
ops.gemm_bias_act = orig
def timed(f, n=20):
    f(); s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    s.record()
    for _ in range(n): f()
    e.record(); torch.cuda.synchronize()
    return s.elapsed_time(e) / n * 1e3
for (mm, n, k, res, act), cnt in shapes.items():
    a = torch.randn(mm, k, device="cuda"); w = torch.randn(n, k, device="cuda") / k ** 0.5; b = torch.randn(n, device="cuda")
    r = torch.randn(mm, n, device="cuda") if res else None
    ref = a.double() @ w.double().t() + b.double() + (r.double() if res else 0)
    if act: ref = ref.clamp_min(0)
    line = f"M={mm:7d} N={n:5d} K={k:5d} res={int(res)} act={act} x{cnt}:"
    for name, sp in (("hipBLASLt", False), ("split", True)):
        if sp and k % 8:
            line += "  split n/a"; continue
        ws = ops.gemm_split_weights(w) if sp else None
        try:
            y = orig(a, w, b, act, residual=r, w_split=ws, split=sp)
            err = (y.double() - ref).abs().max().item()
            t = timed(lambda: orig(a, w, b, act, residual=r, w_split=ws, split=sp))
            line += f"  {name} {t:7.1f} us err {err:.2e}"
        except Exception as ex:
            line += f"  {name} FAILED {type(ex).__name__}: {ex}"
    print(line, flush=True)
from torch.profiler import profile, ProfilerActivity
with torch.no_grad(), profile(activities=[ProfilerActivity.CPU, ProfilerActivity.CUDA], record_shapes=True, with_stack=True) as prof:
    m(x); torch.cuda.synchronize()
for ev in prof.key_averages(group_by_input_shape=True, group_by_stack_n=6):
    if ev.key.startswith("aten::") and ev.device_time_total > 0 and ev.key not in ("aten::empty",):
        st = [s for s in ev.stack if "adverse_weather" in s][:3]
        print(f"{ev.key:28s} x{ev.count:3d} {ev.device_time_total:9.1f} us  {ev.input_shapes}  {st}", flush=True)
