"""Which torch copy / layout kernels are left in one eval forward (8 frames of 1024x2048): shapes and the python frames that
issue them.    python tools/find_copies.py"""
import sys
from pathlib import Path

import torch
from torch.profiler import ProfilerActivity, profile

sys.path.insert(0, str(Path(__file__).resolve().parent.parent))
import adverse_weather_semantic_segmentation_robustness_benchmark_amd as P   # noqa: E402


def main():
    torch.manual_seed(0)
    model = P.EnsembleModel(num_classes=19, include_depth=True, pretrained=False).cuda().eval()
    x = torch.randn(8, 3, 1024, 2048, device="cuda")
    lab = torch.randint(0, 19, (8, 1024, 2048), device="cuda", dtype=torch.uint8)
    counts = torch.zeros(6, 19, 19, dtype=torch.int64, device="cuda"); oob = torch.zeros(1, dtype=torch.int64, device="cuda")
    cond = torch.zeros(8, dtype=torch.int32, device="cuda")
    with torch.no_grad():
        for _ in range(2):
            model.forward_eval(x, lab, counts, oob, cond, want_logits=False, want_pred=False)
        torch.cuda.synchronize()
        with profile(activities=[ProfilerActivity.CPU, ProfilerActivity.CUDA], record_shapes=True, with_stack=True) as prof:
            model.forward_eval(x, lab, counts, oob, cond, want_logits=False, want_pred=False)
            torch.cuda.synchronize()
    rows = []
    for e in prof.events():
        if e.name in ("aten::copy_", "aten::contiguous", "aten::clone", "aten::cat", "aten::max_pool2d_with_indices", "aten::upsample_bilinear2d",
                      "aten::mean", "aten::sub", "aten::add", "aten::mul") and e.device_time_total > 20:
            stack = [s for s in (e.stack or []) if "adverse_weather" in s][:3]
            rows.append((e.device_time_total, e.name, str(e.input_shapes)[:80], " <- ".join(s.split("/")[-1] for s in stack)))
    rows.sort(reverse=True)
    for t, n, sh, st in rows[:25]:
        print(f"{t:8.1f} us  {n:28s} {sh:80s} {st}")


if __name__ == "__main__":
    main()
