#!/usr/bin/env python3
"""Record the output of every conv_bn_act / gemm / winograd call of the ResNet + decoder eval forward over three runs on
the same input and report the first call whose output differs between runs."""
import sys
from pathlib import Path

import torch

ROOT = Path(__file__).resolve().parent.parent
sys.path.insert(0, str(ROOT))
from adverse_weather_semantic_segmentation_robustness_benchmark_amd import ops  # noqa: E402
from adverse_weather_semantic_segmentation_robustness_benchmark_amd.models import fused  # noqa: E402
from adverse_weather_semantic_segmentation_robustness_benchmark_amd.models.model import EnsembleModel  # noqa: E402

torch.manual_seed(0)
torch.backends.cudnn.deterministic = "--deterministic-convs" in sys.argv
model = EnsembleModel(pretrained=False).cuda().eval()
x = torch.randn(8, 3, 256, 512, device="cuda")
log = []


def wrap(mod, name):
    orig = getattr(mod, name)

    def f(*a, **k):
        out = orig(*a, **k)
        desc = name
        for t in a:
            if isinstance(t, torch.nn.Conv2d):
                desc += f" conv {t.in_channels}->{t.out_channels} k{t.kernel_size[0]} s{t.stride[0]} d{t.dilation[0]}"
        shapes = [tuple(t.shape) for t in a if isinstance(t, torch.Tensor)][:2]
        amax = max([t.abs().max().item() for t in a if isinstance(t, torch.Tensor) and t.dtype == torch.float32][:1] or [0])
        log.append((desc + f" {shapes} in|max|={amax:.3g}", out.clone() if isinstance(out, torch.Tensor) else None))
        return out
    setattr(mod, name, f)


wrap(fused, "conv_bn_act")
for n in ("gemm_split_bias_act", "gemm_bias_act", "conv3x3_winograd", "aspp_depthwise3", "dwconv3x3_upcat", "dwconv3x3_nhwc", "bias_act_nhwc_"):
    wrap(ops, n)
runs = []
for r in range(3):
    log.clear()
    with torch.no_grad():
        feats = fused.resnet_features(model.deeplabv3plus.model.encoder, x.contiguous(memory_format=torch.channels_last))
        model.deeplabv3plus.model.decoder.forward_fused(*feats)
    runs.append(list(log))
for a, b, tag in ((0, 1, "run0 vs run1"), (1, 2, "run1 vs run2")):
    nd = 0
    for i, ((da, ta), (db, tb)) in enumerate(zip(runs[a], runs[b])):
        if ta is not None and not torch.equal(ta, tb):
            nd += 1
            if nd <= 6:
                print(f"{tag}: call {i} differs: {da}: max|d| {(ta - tb).abs().max().item():.3e} of |max| {ta.abs().max().item():.3g}, {int((ta != tb).sum())} elements")
    print(f"{tag}: {nd} of {len(runs[a])} calls differ")
