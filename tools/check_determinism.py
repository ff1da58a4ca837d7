#!/usr/bin/env python3
"""Is the eval forward a pure function of the frame?  (pooled mIoU must be bit-identical at any GPU
count, which needs every frame's logits to be independent of the run, the batch it sits in and its position.)

    python tools/check_determinism.py [--height 256 --width 512]

Runs the ensemble forward on the same frames (a) twice in the same batch, (b) alone vs inside a batch of 8, (c) at a
different batch position, and compares every output bit for bit, stage by stage (MiT tokens, ResNet features, decoder,
member logits, depth).  Exit code 1 on any difference."""
import argparse
import sys
from pathlib import Path

import torch

ROOT = Path(__file__).resolve().parent.parent
sys.path.insert(0, str(ROOT))
from adverse_weather_semantic_segmentation_robustness_benchmark_amd.models.model import EnsembleModel  # noqa: E402
from adverse_weather_semantic_segmentation_robustness_benchmark_amd.models import fused  # noqa: E402


def stages(model, x):
    out = {}
    with torch.no_grad():
        tok = fused.mit_features_nhwc(model.segformer.segformer, x)
        out["mit_tokens"] = tok
        feats = fused.resnet_features(model.deeplabv3plus.model.encoder, x.contiguous(memory_format=torch.channels_last))
        for i, f in enumerate(feats):
            if f is not None and i >= 2:
                out[f"resnet_f{i}"] = f
        out["decoder"] = model.deeplabv3plus.model.decoder.forward_fused(*feats)
        res = model.forward_eval(x, want_logits=True, want_pred=True)
        for k, v in res.items():
            out[k] = v
    return {k: v.clone() for k, v in out.items()}


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--height", type=int, default=256)
    ap.add_argument("--width", type=int, default=512)
    ap.add_argument("--deterministic-convs", action="store_true")
    a = ap.parse_args()
    torch.manual_seed(0)
    torch.backends.cudnn.deterministic = "--deterministic-convs" in sys.argv
    model = EnsembleModel(pretrained=False).cuda().eval()
    x = torch.randn(8, 3, a.height, a.width, device="cuda")
    bad = 0

    def cmp(tag, A, B, sel_a=slice(None), sel_b=slice(None)):
        nonlocal bad
        for k in A:
            ta, tb = A[k][sel_a], B[k][sel_b]
            if not torch.equal(ta, tb):
                d = (ta.float() - tb.float()).abs().max().item()
                print(f"DIFF {tag}: {k}: max |d| = {d:.3e}, {int((ta != tb).sum())} elements")
                bad += 1
    r1, r2 = stages(model, x), stages(model, x)
    cmp("same batch twice", r1, r2)
    alone = stages(model, x[3:4].contiguous())
    cmp("frame 3 alone vs in the batch", alone, r1, slice(0, 1), slice(3, 4))
    perm = torch.tensor([5, 3, 0, 1, 2, 4, 6, 7], device="cuda")
    rp = stages(model, x[perm].contiguous())
    cmp("frame 3 at position 1", rp, r1, slice(1, 2), slice(3, 4))
    print("deterministic and batch-independent" if bad == 0 else f"{bad} differing outputs")
    raise SystemExit(1 if bad else 0)


if __name__ == "__main__":
    main()
