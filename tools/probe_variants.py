import sys, torch
sys.path.insert(0, "/root/repo")
import adverse_weather_semantic_segmentation_robustness_benchmark_amd as P
torch.manual_seed(0)
x = torch.randn(1, 3, 256, 256, device="cuda")
for name in ("nvidia/segformer-b5-finetuned-ade-640-640", "nvidia/segformer-b2-finetuned-ade-512-512"):
    m = P.SegFormerModel(model_name=name, num_classes=19, include_depth=True, pretrained=False).cuda().eval()
    with torch.no_grad():
        o = m(x); m.fused_eval = False; r = m(x)
    print(name, tuple(o["segmentation"].shape), "max|diff| seg", (o["segmentation"] - r["segmentation"]).abs().max().item(), "depth", (o["depth"] - r["depth"]).abs().max().item())
for bb in ("resnet101", "resnet34" ):
    try:
        m = P.DeepLabV3PlusModel(backbone=bb, num_classes=19, include_depth=True, pretrained=False).cuda().eval()
        with torch.no_grad():
            o = m(x); m.fused_eval = False; r = m(x)
        print(bb, tuple(o["segmentation"].shape), "seg magnitude", r["segmentation"].abs().max().item(), "max|diff| seg", (o["segmentation"] - r["segmentation"]).abs().max().item(), "depth", (o["depth"] - r["depth"]).abs().max().item())
    except Exception as e:
        print(bb, "ERR", repr(e)[:200])
