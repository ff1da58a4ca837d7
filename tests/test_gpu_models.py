"""-m gpu: the model / metrics / trainer layer on the HIP path against the as-written torch fp32
op graph (the reference's own op sequence: F.interpolate -> conv -> BN -> ReLU -> conv, module
ASPP, ...) evaluated on the CPU with the same weights."""
import copy

import numpy as np
import pytest

from tests.conftest import maxconf_flips_are_rounding_ties
import torch

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def P(native):
    import adverse_weather_semantic_segmentation_robustness_benchmark_amd as pkg
    return pkg


def as_written_cpu(model, x):
    """Same weights, reference op graph, CPU fp32."""
    m = copy.deepcopy(model).cpu().eval()
    for mod in m.modules():
        mod.fused_eval = False
    with torch.no_grad():
        return m(x.cpu())


def calibrate_bn(model, seed=0):
    """Random-init nets in eval mode (identity BatchNorm) blow activations up by orders of magnitude;
    give every BN plausible running stats so the logits are O(1) and a 1e-4 tolerance means something."""
    g = torch.Generator().manual_seed(seed)
    for m in model.modules():
        if isinstance(m, torch.nn.BatchNorm2d):
            m.running_var.copy_(torch.rand(m.running_var.shape, generator=g) * 1.5 + 0.5)
            m.running_mean.copy_((torch.rand(m.running_mean.shape, generator=g) - 0.5) * 0.2)
            m.weight.data.copy_(torch.rand(m.weight.shape, generator=g) * 0.5 + 0.25)
    return model


def rel_err(a, b):
    return (a - b).abs().max().item() / max(1.0, b.abs().max().item())


def abs_err(a, b, what=""):
    """max |a - b| next to the reference's magnitude (north_star's gate is 1e-4 ABSOLUTE on fp32 logits)."""
    e, mag = (a - b).abs().max().item(), b.abs().max().item()
    print(f"{what}: max abs err {e:.3e} at reference magnitude {mag:.3g} (relative {e / max(mag, 1e-30):.3e})")
    return e


def as_written_gpu(model):
    """Same weights, the reference's op graph, on torch-ROCm (a copy with the fused eval executors switched off)."""
    m = copy.deepcopy(model).eval()
    for mod in m.modules():
        mod.fused_eval = False
    return m


def test_segformer_model_hip_vs_as_written(P):
    torch.manual_seed(0)
    m = calibrate_bn(P.SegFormerModel(num_classes=19, include_depth=True, pretrained=False)).cuda().eval()
    x = torch.randn(2, 3, 128, 192, device="cuda")
    out = m(x)
    ref = as_written_cpu(m, x)
    assert set(out) == {"segmentation", "depth"}
    assert out["segmentation"].shape == (2, 19, 128, 192) and out["depth"].shape == (2, 1, 128, 192)
    # north_star: 1e-4 ABSOLUTE on fp32 logits (calibrated BatchNorm keeps them O(1)); absolute error and magnitude printed
    assert abs_err(out["segmentation"].cpu(), ref["segmentation"], "segformer 128x192 logits vs as-written CPU graph") < 1e-4
    assert abs_err(out["depth"].cpu(), ref["depth"], "segformer 128x192 depth") < 1e-4


@pytest.mark.parametrize("split", [False, True])
def test_segformer_model_hip_vs_as_written_own_attention_path(P, split, monkeypatch):
    """256x256: every MiT stage has 64 keys (a multiple of 32), so the encoder runs awseg_attention_d32 (float32 MFMA)
    or awseg_attention_d32_split (split-operand f16 MFMA), not the SDPA fall-back the smaller sizes take; both against
    the as-written CPU graph at the same 1e-4 gate."""
    from adverse_weather_semantic_segmentation_robustness_benchmark_amd import ops as O
    monkeypatch.setattr(O, "ATTENTION_SPLIT", split)
    torch.manual_seed(3)
    m = calibrate_bn(P.SegFormerModel(num_classes=19, include_depth=True, pretrained=False)).cuda().eval()
    x = torch.randn(1, 3, 256, 256, device="cuda")
    out = m(x)
    ref = as_written_cpu(m, x)
    assert abs_err(out["segmentation"].cpu(), ref["segmentation"], f"segformer 256x256 split={split} logits vs as-written CPU graph") < 1e-4
    assert abs_err(out["depth"].cpu(), ref["depth"], f"segformer 256x256 split={split} depth") < 1e-4


def test_deeplab_model_hip_vs_as_written(P):
    torch.manual_seed(1)
    m = calibrate_bn(P.DeepLabV3PlusModel(num_classes=19, include_depth=True, pretrained=False)).cuda().eval()
    x = torch.randn(1, 3, 128, 256, device="cuda")
    out = m(x)
    ref = as_written_cpu(m, x)
    assert out["segmentation"].shape == (1, 19, 128, 256) and out["depth"].shape == (1, 1, 128, 256)
    # north_star: 1e-4 abs on fp32 logits (calibrated BatchNorm keeps them O(1)); absolute error and magnitude printed
    assert abs_err(out["segmentation"].cpu(), ref["segmentation"], "deeplab logits vs as-written CPU graph") < 1e-4
    assert abs_err(out["depth"].cpu(), ref["depth"], "deeplab depth") < 1e-4


def test_segformer_b5_f32_grade_vs_as_written(P):
    """BASELINE configs[4]'s SegFormer member (MiT-B5: depths 3/6/40/3, hidden 64..512, head width 64) on the float32-grade HIP
    path against the AS-WRITTEN torch graph on the CPU at north_star's 1e-4 abs — so the bf16 path's reference
    (tests/test_gpu_bf16.py, test_gpu_configs.py compare bf16 with this float32-grade path) is itself pinned to the op graph
    the reference runs, not only to this repo's own kernels."""
    torch.manual_seed(11)
    m = calibrate_bn(P.SegFormerModel(model_name="nvidia/segformer-b5-finetuned-cityscapes-1024-1024", num_classes=19, include_depth=True,
                                      pretrained=False)).cuda().eval()
    assert sum(len(st.blocks) for st in m.segformer.stages) == 52                  # B5, not the B0 fall-back
    x = torch.randn(1, 3, 128, 192, device="cuda")
    out = m(x)
    ref = as_written_cpu(m, x)
    assert abs_err(out["segmentation"].cpu(), ref["segmentation"], "segformer-B5 128x192 logits vs as-written CPU graph") < 1e-4
    assert abs_err(out["depth"].cpu(), ref["depth"], "segformer-B5 128x192 depth") < 1e-4


def test_deeplab_r101_f32_grade_vs_as_written(P):
    """BASELINE configs[4]'s DeepLabV3+ member (ResNet-101 encoder) on the float32-grade HIP path against the as-written module
    graph on the CPU at 1e-4 abs (same purpose as the B5 test above)."""
    torch.manual_seed(12)
    m = calibrate_bn(P.DeepLabV3PlusModel(backbone="resnet101", num_classes=19, include_depth=True, pretrained=False)).cuda().eval()
    assert len(m.model.encoder.layer3) == 23                                       # R101
    x = torch.randn(1, 3, 128, 256, device="cuda")
    out = m(x)
    ref = as_written_cpu(m, x)
    assert abs_err(out["segmentation"].cpu(), ref["segmentation"], "deeplab-R101 logits vs as-written CPU graph") < 1e-4
    assert abs_err(out["depth"].cpu(), ref["depth"], "deeplab-R101 depth") < 1e-4


def test_aspp_fused_vs_module(P):
    from adverse_weather_semantic_segmentation_robustness_benchmark_amd.models.deeplab import DeepLabV3PlusDecoder
    torch.manual_seed(2)
    dec = calibrate_bn(DeepLabV3PlusDecoder((3, 64, 256, 512, 1024, 2048))).cuda().eval()
    x = torch.randn(2, 2048, 40, 48, device="cuda") * 0.1
    with torch.no_grad():
        ref = dec.aspp[0](x)
        got = dec.aspp_fused(x)
    assert abs_err(got, ref, "fused ASPP vs module graph") < 1e-4


@pytest.mark.parametrize("strategy", ["weighted_average", "max_confidence", "mean"])
def test_ensemble_forward_eval_contract(P, oracle, strategy):
    torch.manual_seed(3)
    m = calibrate_bn(P.EnsembleModel(num_classes=19, include_depth=False, ensemble_strategy=strategy, pretrained=False)).cuda().eval()
    with torch.no_grad():
        m.ensemble_weights.copy_(torch.tensor([0.3, -0.2])); m.temperature.fill_(1.5)
    x = torch.randn(2, 3, 64, 128, device="cuda")
    out = m(x)
    assert set(out) == {"segmentation", "segformer_seg", "deeplabv3plus_seg"}          # model.py:464-468
    s1, s2 = out["segformer_seg"].cpu().numpy(), out["deeplabv3plus_seg"].cpu().numpy()
    w = torch.softmax(m.ensemble_weights.detach().cpu(), 0)
    mode = {"weighted_average": 0, "max_confidence": 1}.get(strategy, 2)
    ref = oracle.combine(s1, s2, mode, float(w[0]), float(w[1]), 1.5)
    got = out["segmentation"].cpu().numpy()
    if mode == 1:
        maxconf_flips_are_rounding_ties(got, ref, s1, s2)          # exact except float32 rounding ties of the two confidences
    else:
        assert np.array_equal(got, ref)                                                # bit-exact given the member logits
    labels = torch.randint(0, 19, (2, 64, 128), dtype=torch.uint8, device="cuda")
    counts = torch.zeros(6, 361, dtype=torch.int64, device="cuda")
    oob = torch.zeros(1, dtype=torch.int64, device="cuda")
    res = m.forward_eval(x, labels, counts, oob, torch.tensor([0, 2], dtype=torch.int32, device="cuda"))
    if mode != 1:
        assert np.array_equal(res["prediction"].cpu().numpy(), oracle.argmax(ref))
        assert np.array_equal(counts[0].cpu().numpy(), oracle.confusion(oracle.argmax(ref), labels.cpu().numpy(), 19))


def test_cpu_tensor_in_eval_mode_raises(P):
    m = P.SegFormerModel(num_classes=5, include_depth=False, pretrained=False).eval()
    with pytest.raises(Exception, match="no CPU fallback"):
        m(torch.randn(1, 3, 64, 64))


def test_metrics_api_on_device(P, oracle, golden_metrics):
    g = golden_metrics
    rm = P.RobustnessMetrics(num_classes=19)
    for n in range(int(g["n_cases"])):
        pred = torch.from_numpy(g[f"pred{n}"]).cuda()
        lab = torch.from_numpy(g[f"label{n}"]).cuda()
        res = rm.iou_metrics.compute_iou(pred, lab)
        if np.isnan(g[f"miou{n}"]):
            assert np.isnan(res["mean_iou"])
        else:
            assert res["mean_iou"] == float(g[f"miou{n}"])                            # bit-identical to the reference
        assert np.array_equal(res["per_class_iou"], g[f"per_class{n}"])
        assert abs(rm.iou_metrics.compute_pixel_accuracy(pred, lab) - float(g[f"pixacc{n}"])) < 1e-12
    assert rm.compute_miou(torch.from_numpy(g["am_logits"]).cuda(), torch.from_numpy(g["am_label"]).cuda()) == float(g["am_miou"])
    ece = rm.calibration_metrics.compute_ece(torch.from_numpy(g["ece_logits"]).cuda(), torch.from_numpy(g["ece_label"]).cuda())
    assert abs(ece - float(g["ece"])) < 1e-5
    with pytest.raises(IndexError):
        rm.iou_metrics.compute_iou(torch.zeros(8, dtype=torch.int64, device="cuda"), torch.full((8,), 19, dtype=torch.int64, device="cuda"))


def test_disagreement_auroc_matches_sklearn(P):
    from sklearn.metrics import roc_auc_score
    torch.manual_seed(4)
    a, b = torch.randn(2, 19, 16, 16, device="cuda") * 2, torch.randn(2, 19, 16, 16, device="cuda") * 2
    lab = torch.randint(0, 19, (2, 16, 16), device="cuda")
    em = P.EnsembleDisagreementMetrics()
    got = em.compute_disagreement_auroc([a, b], lab)
    dis = em.compute_disagreement_map([a, b]).reshape(-1).cpu().numpy()
    mp = (torch.softmax(a, 1) + torch.softmax(b, 1)) / 2
    err = (mp.argmax(1) != lab).reshape(-1).cpu().numpy().astype(np.float32)
    assert abs(got - roc_auc_score(err, dis)) < 1e-9


def test_evaluate_model_end_to_end(P, oracle):
    from adverse_weather_semantic_segmentation_robustness_benchmark_amd.data.loader import CityscapesKITTIDataset, create_dataloader
    from adverse_weather_semantic_segmentation_robustness_benchmark_amd.evaluation.harness import evaluate_model
    torch.manual_seed(5)
    cfg = P.Config({"data": {"weather_conditions": ["clean", "fog", "rain", "snow", "night"]}, "evaluation": {"num_bins": 15}})
    model = calibrate_bn(P.EnsembleModel(include_depth=False, pretrained=False)).cuda().eval()
    ds = CityscapesKITTIDataset(split="test", image_size=(64, 128), weather_schedule="round_robin", num_samples=10)
    loader = create_dataloader(ds, batch_size=4, shuffle=False)
    res = evaluate_model(model, loader, P.RobustnessMetrics(19), torch.device("cuda"), cfg)
    for k in ("overall_miou", "expected_calibration_error", "ensemble_disagreement_auroc", "robustness_degradation_ratio"):
        assert k in res
    for w in ("clean", "fog", "rain", "snow", "night"):
        assert f"miou_{w}" in res and f"ece_{w}" in res
    assert 0.0 <= res["overall_miou"] <= 1.0 and 0.0 <= res["ensemble_disagreement_auroc"] <= 1.0
    # ---- VALUES, against the reference's own pipeline shape (REF/scripts/evaluate.py:166-271): the as-written torch
    # graph per batch, argmax, everything concatenated, then the reference's metric expressions on the whole set.
    # The loader serves the same samples again (they are a function of the global index).
    ref_model = as_written_gpu(model)
    rm = P.RobustnessMetrics(19)
    preds, labels, logits, s1s, s2s, conds = [], [], [], [], [], []
    with torch.no_grad():
        for batch in create_dataloader(ds, batch_size=4, shuffle=False):
            o = ref_model(batch["image"])
            logits.append(o["segmentation"]); preds.append(o["segmentation"].argmax(1)); labels.append(batch["label"])
            s1s.append(o["segformer_seg"]); s2s.append(o["deeplabv3plus_seg"]); conds += list(batch["weather_condition"])
    preds, labels, logits = torch.cat(preds), torch.cat(labels), torch.cat(logits)
    tol = 2e-3                                                   # near-tie argmax flips between two fp32 summation orders
    assert abs(res["overall_miou"] - rm.compute_miou(preds, labels)) < tol
    for w in ("clean", "fog", "rain", "snow", "night"):
        idx = [i for i, c in enumerate(conds) if c == w]
        assert abs(res[f"miou_{w}"] - rm.compute_miou(preds[idx], labels[idx])) < tol
        assert abs(res[f"ece_{w}"] - rm.calibration_metrics.compute_ece(logits[idx], labels[idx])) < 1e-4
    assert abs(res["expected_calibration_error"] - rm.calibration_metrics.compute_ece(logits, labels)) < 1e-4
    auroc = rm.ensemble_metrics.compute_disagreement_auroc([torch.cat(s1s), torch.cat(s2s)], labels)
    assert abs(res["ensemble_disagreement_auroc"] - auroc) < 3e-3      # 2^13-bin rank histogram vs exact ranks
    degs = [rm.compute_robustness_degradation_ratio(res["miou_clean"], res[f"miou_{w}"]) for w in ("fog", "rain", "snow", "night")]
    assert res["robustness_degradation_ratio"] == pytest.approx(float(np.mean(degs)), abs=1e-12)


def test_trainer_one_epoch(P, tmp_path):
    from adverse_weather_semantic_segmentation_robustness_benchmark_amd.data.loader import CityscapesKITTIDataset, create_dataloader
    torch.manual_seed(6)
    model = P.EnsembleModel(num_classes=19, include_depth=True, pretrained=False)
    tr = CityscapesKITTIDataset(split="train", image_size=(64, 128), num_samples=4)
    va = CityscapesKITTIDataset(split="val", image_size=(64, 128), num_samples=4, weather_schedule="round_robin")
    config = {"epochs": 1, "optimizer": {"type": "adamw", "learning_rate": 1e-4}, "scheduler": {"enabled": True, "type": "cosine"},
              "loss": {"type": "fog_density_aware"}, "mlflow": {"enabled": False}}
    t = P.AdverseWeatherTrainer(model, create_dataloader(tr, 2, shuffle=True), create_dataloader(va, 2, shuffle=False), config,
                                torch.device("cuda"), checkpoint_dir=str(tmp_path / "ck"), log_dir=str(tmp_path / "lg"))
    tm = t.train_epoch()
    assert set(tm) == {"train_loss", "train_seg_loss", "train_depth_loss", "train_samples"} and np.isfinite(tm["train_loss"])
    vm = t.validate_epoch()
    assert 0.0 <= vm["val_miou"] <= 1.0 and np.isfinite(vm["val_loss"]) and "val_miou_fog" in vm
    t.save_checkpoint(0, vm, is_best=True)
    ck = torch.load(tmp_path / "ck" / "best.pth", weights_only=False)
    assert set(ck) == {"epoch", "model_state_dict", "optimizer_state_dict", "scheduler_state_dict", "metrics", "config"}
    t.load_checkpoint(str(tmp_path / "ck" / "latest.pth"))
    assert t.current_epoch == 0


def test_mit_fused_forward_matches_hf(P):
    """NHWC functional MiT forward (HIP depthwise+GELU, no layout copies) vs transformers' own forward."""
    from adverse_weather_semantic_segmentation_robustness_benchmark_amd.models import fused
    torch.manual_seed(7)
    m = P.SegFormerModel(num_classes=19, include_depth=False, pretrained=False).cuda().eval()
    x = torch.randn(2, 3, 128, 160, device="cuda")
    with torch.no_grad():
        ref = m.encode(x)                                           # HF forward -> [B,C,h,w]
        got = fused.mit_features_nhwc(m.segformer, x).permute(0, 3, 1, 2)
    assert rel_err(got, ref) < 1e-4


def test_resnet_fused_matches_modules(P):
    from adverse_weather_semantic_segmentation_robustness_benchmark_amd.models import fused
    from adverse_weather_semantic_segmentation_robustness_benchmark_amd.models.deeplab import ResNetEncoder
    torch.manual_seed(8)
    enc = calibrate_bn(ResNetEncoder("resnet50")).cuda().eval()
    x = torch.randn(1, 3, 96, 128, device="cuda")
    with torch.no_grad():
        ref = enc(x)
        got = fused.resnet_features(enc, x, stem_feature=True)
        lean = fused.resnet_features(enc, x)                    # default: stem feature skipped, stem epilogue after the max-pool
    for a, b in zip(got[1:], ref[1:]):
        assert a.shape == b.shape and rel_err(a, b) < 1e-4
    assert lean[1] is None
    for a, b in zip(lean[2:], got[2:]):
        # maxpool(relu(x+b)) == relu(maxpool(x)+b) exactly; what remains is MIOpen's run-to-run summation order
        assert rel_err(a, b) < 1e-5
    # fold cache follows in-place weight updates
    with torch.no_grad():
        enc.conv1.weight.mul_(0.5)
        assert rel_err(fused.resnet_features(enc, x, stem_feature=True)[1], enc(x)[1]) < 1e-4


def test_ensemble_eval_stats_matches_reference_expressions(P, oracle):
    """Fused ECE + disagreement-histogram kernel vs the reference's torch expressions
    (metrics.py:161-194 and :353-367 / :414-426) evaluated on the device."""
    from adverse_weather_semantic_segmentation_robustness_benchmark_amd import ops
    from adverse_weather_semantic_segmentation_robustness_benchmark_amd.evaluation.harness import EvalState, AUROC_LO, AUROC_HI
    from sklearn.metrics import roc_auc_score
    torch.manual_seed(11)
    B, C, H, W = 3, 19, 32, 48
    s1 = torch.randn(B, C, H, W, device="cuda") * 2
    s2 = torch.randn(B, C, H, W, device="cuda") * 2
    lab = torch.randint(0, C, (B, H, W), device="cuda")
    lab[torch.rand(B, H, W, device="cuda") < 0.05] = 255
    lab = lab.to(torch.uint8)
    w = torch.softmax(torch.tensor([0.2, -0.1]), 0).cuda()
    T = torch.tensor([1.3], device="cuda")
    conds = ["clean", "fog", "rain", "snow", "night"]
    st = EvalState(P.RobustnessMetrics(19), conds, "cuda", 15, True)
    cond = torch.tensor([0, 1, 1], dtype=torch.int32, device="cuda")
    ops.ensemble_eval_stats(s1, s2, 0, w, T, lab, cond, st.edges, st.ece, st.auroc, AUROC_LO, AUROC_HI)
    # ECE reference: compute_ece on the combined logits
    logits = (w[0] * s1 + w[1] * s2) / T
    bins = ops.ece_bins_to_numpy(st.ece)
    ece = P.ConfidenceCalibration.ece_from_bins(bins[0])
    cnt, sconf, scorr = oracle.ece_bins(logits.cpu().numpy(), lab.cpu().numpy())
    assert abs(ece - oracle.ece_from_bins(cnt, sconf, scorr)) < 1e-5
    assert np.abs(bins[0]["count"] - cnt).sum() <= 2
    assert bins[2]["count"].sum() == int((lab[1:3] != 255).sum()) and bins[3]["count"].sum() == 0
    # AUROC reference: sklearn on the reference's disagreement map / error flags
    em = P.EnsembleDisagreementMetrics()
    dis = em.compute_disagreement_map([s1, s2]).reshape(-1)
    mp = (torch.softmax(s1, 1) + torch.softmax(s2, 1)) / 2
    err = (mp.argmax(1) != lab).reshape(-1)
    valid = (lab != 255).reshape(-1)
    ref = roc_auc_score(err[valid].cpu().numpy().astype(np.float32), dis[valid].cpu().numpy())
    assert int(st.auroc.sum()) == int(valid.sum())
    assert abs(st.auroc_value() - ref) < 2e-3                      # 2^16-bin rank histogram vs exact ranks


@pytest.mark.parametrize("ldt", [torch.uint8, torch.int64])
def test_combine_confusion_stats_one_pass_equals_the_two_kernels(P, ldt):
    """awseg_combine_confusion_stats against awseg_combine_argmax_confusion + awseg_ensemble_eval_stats on the same inputs:
    confusion counters per slot (incl. ignore_index 255, the uint8 index wrap, NaN logits), ECE bins and the disagreement
    histogram must be IDENTICAL integers."""
    from adverse_weather_semantic_segmentation_robustness_benchmark_amd import ops
    from adverse_weather_semantic_segmentation_robustness_benchmark_amd.evaluation.harness import EvalState, AUROC_LO, AUROC_HI
    torch.manual_seed(5)
    B, C, H, W = 3, 19, 40, 64
    s1 = torch.randn(B, C, H, W, device="cuda") * 2
    s2 = torch.randn(B, C, H, W, device="cuda") * 2
    s1[0, 3, 5, 7] = float("nan"); s2[1, :, 2, 2] = 4.0          # a NaN logit; an exact tie across classes
    lab = torch.randint(0, C, (B, H, W), device="cuda")
    lab[torch.rand(B, H, W, device="cuda") < 0.07] = 255
    lab = lab.to(ldt)
    w = torch.softmax(torch.tensor([0.3, -0.2]), 0).cuda()
    T = torch.tensor([1.7], device="cuda")
    conds = ["clean", "fog", "rain", "snow", "night"]
    cond = torch.tensor([1, 0, 4], dtype=torch.int32, device="cuda")
    for mode, ww in ((0, w), (2, None)):
        a = EvalState(P.RobustnessMetrics(19), conds, "cuda", 15, True)
        b = EvalState(P.RobustnessMetrics(19), conds, "cuda", 15, True)
        ops.combine_argmax_confusion(s1, s2, mode, ww, T, want_logits=False, want_pred=False, label=lab, counts=a.acc.counts, oob=a.acc.oob, cond=cond)
        ops.ensemble_eval_stats(s1, s2, mode, ww, T, lab, cond, a.edges, a.ece, a.auroc, AUROC_LO, AUROC_HI)
        ops.combine_confusion_stats(s1, s2, mode, ww, T, lab, cond, b.acc.counts, b.acc.oob, b.edges, b.ece, b.auroc, AUROC_LO, AUROC_HI)
        assert torch.equal(a.acc.counts, b.acc.counts) and torch.equal(a.acc.oob, b.acc.oob) and int(a.acc.counts.sum()) > 0
        # the NaN pixel: the stats kernels' own argmax (plain >) and torch's rule may differ there in `correct` only
        assert torch.equal(a.ece[..., 0], b.ece[..., 0]) and torch.equal(a.ece[..., 1], b.ece[..., 1])
        assert (a.ece[..., 2] - b.ece[..., 2]).abs().sum().item() <= 1
        assert torch.equal(a.auroc, b.auroc)


def test_trainer_epoch_values_match_the_as_written_graph(P, tmp_path):
    """A17 (PKG/training/trainer.py:280-478): the loss dict of train_epoch / validate_epoch against the reference's
    arithmetic written out in torch — cross_entropy(reduction='none') * (1 + 2 * density), .mean(), + 0.1 * MSE depth,
    per-sample-weighted running means (:346-353, :371-373) — on the same batches, the same density draws
    (density_rng='torch': the reference's torch.rand order) and the same dropout stream.  lr = 0 keeps the weights
    fixed so the second evaluation sees the model the trainer saw."""
    import torch.nn.functional as F
    from adverse_weather_semantic_segmentation_robustness_benchmark_amd.data.loader import CityscapesKITTIDataset, create_dataloader
    torch.manual_seed(16)
    model = calibrate_bn(P.EnsembleModel(num_classes=19, include_depth=True, pretrained=False))
    tr = CityscapesKITTIDataset(split="train", image_size=(64, 128), num_samples=6)
    va = CityscapesKITTIDataset(split="val", image_size=(64, 128), num_samples=5, weather_schedule="round_robin")
    config = {"epochs": 1, "optimizer": {"type": "sgd", "learning_rate": 0.0, "momentum": 0.0, "weight_decay": 0.0},
              "loss": {"type": "fog_density_aware"}, "density_rng": "torch", "grad_clip": 1.0}
    t = P.AdverseWeatherTrainer(model, create_dataloader(tr, 2, shuffle=True), create_dataloader(va, 2, shuffle=False), config,
                                torch.device("cuda"), checkpoint_dir=str(tmp_path / "ck"), log_dir=str(tmp_path / "lg"))
    table = {"fog": (0.5, 0.5), "rain": (0.3, 0.2), "snow": (0.3, 0.2)}

    def density(conds, h, w):                                        # trainer.py:494-511
        d = torch.zeros(len(conds), h, w)
        for i, c in enumerate(conds):
            a, b = table.get(c, (0.1, 0.0))
            d[i] = torch.rand(h, w) * a + b
        return d.cuda()

    def reference_epoch(loader, train):
        sums, n = np.zeros(3), 0
        for batch in loader:
            out = model(batch["image"])
            dens = density([str(c) for c in batch["weather_condition"]], *batch["image"].shape[2:])
            ce = F.cross_entropy(out["segmentation"], batch["label"].long(), reduction="none")
            seg = (ce * (1.0 + 2.0 * dens)).mean()
            dl = F.mse_loss(out["depth"].squeeze(1), batch["depth"], reduction="none").mean()
            bs = batch["image"].size(0)
            sums += np.array([(seg + 0.1 * dl).item(), seg.item(), dl.item()]) * bs
            n += bs
        return sums / n, n

    torch.manual_seed(99)
    tm = t.train_epoch()
    torch.manual_seed(99)
    model.train()
    with torch.no_grad():
        ref, n = reference_epoch(create_dataloader(tr, 2, shuffle=True), True)
    print("train_epoch", tm, "reference", ref)
    assert tm["train_samples"] == n == 6
    for k, r in zip(("train_loss", "train_seg_loss", "train_depth_loss"), ref):
        assert abs(tm[k] - r) <= 1e-4 * max(1.0, abs(r)), (k, tm[k], r)

    torch.manual_seed(77)
    vm = t.validate_epoch()
    torch.manual_seed(77)
    ref_model = as_written_gpu(model)
    model_backup, model_ref = model, ref_model
    model = ref_model                                                # reference_epoch closes over `model`
    with torch.no_grad():
        ref, n = reference_epoch(create_dataloader(va, 2, shuffle=False), False)
        preds, labels, conds = [], [], []
        for batch in create_dataloader(va, 2, shuffle=False):
            preds.append(ref_model(batch["image"])["segmentation"].argmax(1)); labels.append(batch["label"])
            conds += list(batch["weather_condition"])
    model = model_backup
    print("validate_epoch", vm, "reference", ref)
    assert vm["val_samples"] == n == 5
    for k, r in zip(("val_loss", "val_seg_loss", "val_depth_loss"), ref):
        assert abs(vm[k] - r) <= 1e-4 * max(1.0, abs(r)), (k, vm[k], r)
    rm = P.RobustnessMetrics(19)
    preds, labels = torch.cat(preds), torch.cat(labels)
    assert abs(vm["val_miou"] - rm.compute_miou(preds, labels)) < 2e-3
    for w in set(conds):
        idx = [i for i, c in enumerate(conds) if c == w]
        assert abs(vm[f"val_miou_{w}"] - rm.compute_miou(preds[idx], labels[idx])) < 2e-3


def test_eval_forward_is_a_pure_function_of_the_frame(P):
    """SURVEY §8(d) parity gate needs pooled mIoU identical at any GPU count, i.e. a frame's logits must not depend on
    the run or on where the frame sits in a batch of the same size.  (MIOpen's default solver for the strided 3x3
    convolutions accumulates with atomics; those convolutions run as im2col + GEMM here.)  Bit-exact comparisons."""
    torch.manual_seed(21)
    m = P.EnsembleModel(num_classes=19, include_depth=True, pretrained=False).cuda().eval()
    x = torch.randn(4, 3, 128, 256, device="cuda")
    a = m.forward_eval(x, want_logits=True, want_pred=True)
    junk = torch.randn(1 << 22, device="cuda")                      # move the allocator between the runs
    b = m.forward_eval(x, want_logits=True, want_pred=True)
    for k in a:
        assert torch.equal(a[k], b[k]), f"{k} differs between two runs on the same batch"
    perm = [2, 0, 3, 1]
    c = m.forward_eval(x[perm].contiguous(), want_logits=True, want_pred=True)
    for k in a:
        assert torch.equal(a[k][perm], c[k]), f"{k} depends on the frame's position in the batch"
    del junk


@pytest.mark.parametrize("size", [(128, 192), (160, 256)])
def test_segformer_training_heads_on_hip_match_the_as_written_graph(P, size):
    """BASELINE config 4 / VERDICT r1 #6: in training SegFormerModel runs conv3x3(interpolate(f)) — the first layer of both
    heads (PKG/models/model.py:209-214, :219-221) — as ops._UpConv3x3 (HIP forward + HIP adjoint backward) in front of the
    reference's BatchNorm (batch statistics) / ReLU / Dropout2d / convolution modules.  On well-conditioned encoder features:
    outputs, the gradient with respect to the features and every head parameter gradient against the as-written graph
    (F.interpolate -> Conv2d -> ...) with the same weights: <= 1e-4 of each tensor's magnitude, or — where batch-statistics
    BatchNorm's backward cancels leading terms and ANY float32 evaluation is further than that from the truth — no further
    from the as-written graph in FLOAT64 (CPU) than 4x what the as-written float32 graph is.  Dropout off (one function for
    all three runs).  (A random-init ENCODER in front would not do: its output is nearly constant over the image, BatchNorm
    then divides by a variance at rounding level and any two float32 evaluations differ by per cent.)"""
    import copy
    H, W = size
    torch.manual_seed(31)
    m = P.SegFormerModel(num_classes=19, include_depth=True, pretrained=False)
    for mod in m.modules():
        if isinstance(mod, (torch.nn.Dropout, torch.nn.Dropout2d)):
            mod.p = 0.0
        if isinstance(mod, torch.nn.BatchNorm2d):
            # ReLU is not differentiable at 0: with ~1e7 activations a handful sit within rounding distance of it and flip their
            # mask between ANY two float32 evaluations, which moves a per-channel gradient sum by one element (measured 4e-3 of
            # the sum).  A positive BatchNorm shift keeps every pre-activation away from 0, so the comparison prices the
            # arithmetic and not the flips.
            torch.nn.init.constant_(mod.bias, 8.0)
    m64 = copy.deepcopy(m).double().train()
    m = m.cuda().train()
    f0 = torch.randn(2, 256, H // 32, W // 32)
    gseg, gdep = torch.randn(2, 19, H, W), torch.randn(2, 1, H, W)

    def run(model, fused, dev, dtype):
        model.fused_train = fused
        model.zero_grad(set_to_none=True)
        f = f0.to(dev, dtype).requires_grad_(True)
        out = model.heads_forward(f, H, W)
        ((out["segmentation"] * gseg.to(dev, dtype)).sum() + (out["depth"] * gdep.to(dev, dtype)).sum()).backward()
        res = {"out." + k: v.detach().double().cpu() for k, v in out.items()}
        res["d features"] = f.grad.double().cpu()
        res.update({"grad " + n: p.grad.double().cpu() for n, p in model.named_parameters() if p.grad is not None})
        return res

    r_h, r_t, r_64 = run(m, True, "cuda", torch.float32), run(m, False, "cuda", torch.float32), run(m64, False, "cpu", torch.float64)
    assert set(r_h) == set(r_64) and "grad segmentation_head.0.weight" in r_h and "grad depth_head.depth_head.0.bias" in r_h
    gmax = max(v.abs().max().item() for k, v in r_64.items() if k.startswith("grad "))
    worst = worst_t = 0.0
    bad = []
    for k, ref in r_64.items():
        mag = ref.abs().max().item()
        if k in ("grad segmentation_head.0.bias", "grad depth_head.depth_head.0.bias", "grad depth_head.depth_head.4.bias"):
            # a convolution bias in front of batch-statistics BatchNorm has ZERO gradient in exact arithmetic
            assert r_h[k].abs().max().item() < 1e-2 * gmax and r_t[k].abs().max().item() < 1e-2 * gmax, k
            continue
        scale = max(mag, 1e-4 * gmax) if k.startswith("grad ") else mag
        e_h, e_t = (r_h[k] - ref).abs().max().item() / scale, (r_t[k] - ref).abs().max().item() / scale
        worst, worst_t = max(worst, e_h), max(worst_t, e_t)
        print(f"  {k}: HIP {e_h:.2e}  as-written f32 {e_t:.2e}  (|ref| {mag:.2e})")
        bad = bad + [k] if e_h > max(1e-4, 4 * e_t) else bad
    assert not bad, bad
    print(f"training heads {size}: worst relative error vs float64: HIP path {worst:.2e}, as-written float32 graph {worst_t:.2e} ({len(r_64)} tensors)")


def test_segformer_training_step_through_the_whole_model_runs_on_hip(P):
    """Whole-model smoke test of the same path: loss decreases over a few SGD steps and every parameter receives a finite
    gradient (values are pinned by the heads test above and by test_upconv3x3_train_forward_and_adjoint_vs_torch_autograd)."""
    torch.manual_seed(32)
    m = P.SegFormerModel(num_classes=19, include_depth=True, pretrained=False).cuda().train()
    x = torch.randn(2, 3, 128, 192, device="cuda")
    lab = torch.randint(0, 19, (2, 128, 192), device="cuda")
    opt = torch.optim.SGD(m.parameters(), lr=0.05)
    losses = []
    for _ in range(4):
        opt.zero_grad()
        out = m(x)
        loss = torch.nn.functional.cross_entropy(out["segmentation"], lab) + out["depth"].mean()
        loss.backward()
        assert all(p.grad is None or torch.isfinite(p.grad).all() for p in m.parameters())
        opt.step()
        losses.append(loss.item())
    assert losses[-1] < losses[0], losses
