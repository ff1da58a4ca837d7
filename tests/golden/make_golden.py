#!/usr/bin/env python3
"""Generate tests/golden/*.npz by IMPORTING THE REFERENCE (run in the build container only).

    python tests/golden/make_golden.py            # needs /root/reference

The reference cannot travel to the GPU box, so its outputs on fixed seeded inputs are
committed here as small data fixtures; this script is how they were made.

What is imported:
  * PKG/evaluation/metrics.py  — as is                      -> A12/A13/A14, ECE
  * PKG/data/preprocessing.py  — loaded by path with an inert, never-called `cv2`
    module entry (cv2 is absent; fog / night / depth never touch it)   -> A2/A3/A6
  * PKG/models/model.py        — loaded by path with inert `torchvision` /
    `segmentation_models_pytorch` entries (absent; never called)        -> A11/A15,
    and `SegFormerModel.forward`'s upsample+head arithmetic via the module pieces.
Nothing that fetches weights/configs by name is constructed.
rain / snow need the real cv2: no golden vectors exist for them (parity unpinned).
"""
import importlib.util
import sys
import types
from pathlib import Path

import numpy as np
import torch

REF = Path("/root/reference")
PKG = REF / "src" / "adverse_weather_semantic_segmentation_robustness_benchmark"
OUT = Path(__file__).resolve().parent


def _load(name, path):
    spec = importlib.util.spec_from_file_location(name, path)
    mod = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(mod)
    return mod


def load_reference():
    sys.path.insert(0, str(REF / "src"))
    from adverse_weather_semantic_segmentation_robustness_benchmark.evaluation import metrics
    import transformers  # noqa: F401  (must be imported before the placeholders below)
    for name in ("cv2", "torchvision", "torchvision.models", "segmentation_models_pytorch"):
        if name not in sys.modules:
            sys.modules[name] = types.ModuleType(name)
    sys.modules["torchvision"].models = sys.modules["torchvision.models"]
    pre = _load("ref_preprocessing", PKG / "data" / "preprocessing.py")
    model = _load("ref_model", PKG / "models" / "model.py")
    return metrics, pre, model


def gen_weather(pre):
    """fog / night / synthetic depth: (image, seed, intensity) -> u8 output, plus the fields
    the reference drew, recovered by replaying its RNG call order."""
    out = {}
    tf = pre.WeatherDegradationTransforms()
    cases = [(32, 64, 1, 0.5), (48, 80, 2, None), (17, 23, 3, 0.9), (64, 128, 4, None)]
    for k, (h, w, seed, inten) in enumerate(cases):
        rs = np.random.RandomState(1000 + seed)
        img = rs.randint(0, 255, (h, w, 3), dtype=np.uint8)
        out[f"img{k}"] = img
        out[f"case{k}"] = np.array([h, w, seed, -1.0 if inten is None else inten], dtype=np.float64)
        # ---- fog
        np.random.seed(seed)
        out[f"fog{k}"] = tf.apply_weather_effect(img, "fog", inten)
        np.random.seed(seed)
        noise = np.random.normal(0, 10, (h, w))
        i_used = np.random.uniform(0.3, 0.9) if inten is None else inten
        out[f"fog_noise{k}"] = noise
        out[f"fog_intensity{k}"] = np.float64(i_used)
        np.random.seed(seed)
        out[f"depth{k}"] = tf._generate_synthetic_depth(h, w)
        # ---- night
        np.random.seed(seed)
        out[f"night{k}"] = tf.apply_weather_effect(img, "night", inten)
        np.random.seed(seed)
        i_used = np.random.uniform(0.4, 0.8) if inten is None else inten
        bf = 1 - i_used * np.random.uniform(0.2, 0.6)
        nz = np.random.normal(0, 5.0 / 255.0, (h, w, 3))
        out[f"night_intensity{k}"] = np.float64(i_used)
        out[f"night_brightness{k}"] = np.float64(bf)
        out[f"night_noise{k}"] = nz
    out["n_cases"] = np.int64(len(cases))
    # clean is the identity (preprocessing.py:78-79) and unknown types raise
    try:
        tf.apply_weather_effect(out["img0"], "hail")
        out["unknown_raises"] = np.int64(0)
    except ValueError as e:
        out["unknown_raises"] = np.int64(1)
        out["unknown_msg"] = np.array(str(e))
    np.savez_compressed(OUT / "weather.npz", **out)


def gen_metrics(metrics):
    out = {}
    C = 19
    iou = metrics.IoUMetrics(C)
    rs = np.random.RandomState(7)
    n = 0
    for label_dtype in ("int64", "uint8"):
        for ignore in ("none", "some", "all"):
            pred = rs.randint(0, C, (2, 24, 40)).astype(np.int64)
            lab = rs.randint(0, C, (2, 24, 40))
            if ignore == "some":
                lab[rs.rand(*lab.shape) < 0.1] = 255
            elif ignore == "all":
                lab[:] = 255
            lab = lab.astype(label_dtype)
            res = iou.compute_iou(torch.from_numpy(pred), torch.from_numpy(lab))
            # the confusion matrix itself, by the reference's own expressions (metrics.py:54-71)
            p, t = torch.from_numpy(pred).view(-1), torch.from_numpy(lab).view(-1)
            m = t != 255
            p, t = p[m], t[m]
            cm = torch.zeros(C * C, dtype=torch.long)
            idx = t * C + p
            cm.index_add_(0, idx.long(), torch.ones_like(idx))
            out[f"pred{n}"], out[f"label{n}"] = pred, lab
            out[f"counts{n}"] = cm.numpy()
            out[f"miou{n}"] = np.float64(res["mean_iou"])
            out[f"per_class{n}"] = res["per_class_iou"]
            out[f"valid{n}"] = res["valid_classes"]
            out[f"pixacc{n}"] = np.float64(iou.compute_pixel_accuracy(torch.from_numpy(pred), torch.from_numpy(lab)))
            n += 1
    out["n_cases"] = np.int64(n)
    # 4-D logits input -> argmax path (metrics.py:50-51), incl. ties / NaN / -0.0
    logits = rs.randn(2, C, 8, 12).astype(np.float32)
    logits[0, 3, 0, 0] = logits[0, 7, 0, 0] = 9.0          # tie -> first
    logits[0, 5, 0, 1] = np.nan                            # NaN is max
    logits[0, 2, 0, 2] = np.nan; logits[0, 9, 0, 2] = np.nan
    logits[1, :, 1, 1] = 0.0; logits[1, 4, 1, 1] = -0.0    # -0.0 == 0.0 -> index 0
    logits[1, :, 2, 2] = -np.inf
    out["am_logits"] = logits
    out["am_pred"] = torch.from_numpy(logits).argmax(dim=1).numpy()
    lab = rs.randint(0, C, (2, 8, 12)).astype(np.uint8)
    out["am_label"] = lab
    out["am_miou"] = np.float64(iou.compute_iou(torch.from_numpy(logits), torch.from_numpy(lab))["mean_iou"])
    # degradation ratios (metrics.py:559-563)
    rm = metrics.RobustnessMetrics(num_classes=C)
    pairs = np.array([[0.5, 0.4], [0.0, 0.3], [0.2, 0.3], [0.0205, 0.0121], [1.0, 0.0]])
    out["deg_pairs"] = pairs
    out["deg_ratio"] = np.array([rm.compute_robustness_degradation_ratio(a, b) for a, b in pairs])
    # ECE (metrics.py:143-226)
    ece_logits = (rs.randn(2, C, 16, 20) * 3).astype(np.float32)
    ece_lab = rs.randint(0, C, (2, 16, 20)); ece_lab[rs.rand(2, 16, 20) < 0.05] = 255
    ece_lab = ece_lab.astype(np.uint8)
    cal = metrics.ConfidenceCalibration()
    det = cal.compute_ece(torch.from_numpy(ece_logits), torch.from_numpy(ece_lab), return_details=True)
    out["ece_logits"], out["ece_label"] = ece_logits, ece_lab
    out["ece"] = np.float64(det["ece"])
    out["ece_prop"] = np.array([d["proportion"] for d in det["bin_details"]])
    out["ece_acc"] = np.array([d["accuracy"] for d in det["bin_details"]])
    out["ece_conf"] = np.array([d["confidence"] for d in det["bin_details"]])
    np.savez_compressed(OUT / "metrics.npz", **out)


def gen_model(model):
    out = {}
    rs = np.random.RandomState(11)
    C = 19
    # ---- A11 combine via EnsembleModel.forward run unbound on a stand-in (no constructor)
    s1 = (rs.randn(2, C, 12, 20) * 2).astype(np.float32)
    s2 = (rs.randn(2, C, 12, 20) * 2).astype(np.float32)
    out["seg1"], out["seg2"] = s1, s2

    class Member(torch.nn.Module):
        def __init__(self, t):
            super().__init__(); self.t = t
        def forward(self, x):
            return {"segmentation": self.t}

    n = 0
    for strat in ("weighted_average", "max_confidence", "mean"):
        for ts in (True, False):
            stand = types.SimpleNamespace(
                segformer=Member(torch.from_numpy(s1)), deeplabv3plus=Member(torch.from_numpy(s2)),
                ensemble_strategy=strat, temperature_scaling=ts, include_depth=False,
                ensemble_weights=torch.tensor([0.3, -0.4]), temperature=torch.tensor([1.7]))
            with torch.no_grad():
                res = model.EnsembleModel.forward(stand, torch.zeros(2, 3, 12, 20))
            out[f"combine{n}"] = res["segmentation"].numpy()
            out[f"combine_cfg{n}"] = np.array([["weighted_average", "max_confidence", "mean"].index(strat), int(ts)])
            n += 1
    out["n_combine"] = np.int64(n)
    out["ens_w"] = torch.softmax(torch.tensor([0.3, -0.4]), dim=0).numpy()
    out["ens_t"] = np.float32(1.7)

    # ---- A15 loss
    logits = (rs.randn(2, C, 16, 24) * 2).astype(np.float32)
    lab = rs.randint(0, C, (2, 16, 24))
    dens = rs.rand(2, 16, 24).astype(np.float32)
    dpred = rs.rand(2, 1, 16, 24).astype(np.float32)
    dtgt = rs.rand(2, 16, 24).astype(np.float32)
    out.update(loss_logits=logits, loss_label=lab.astype(np.int64), loss_density=dens, loss_dpred=dpred, loss_dtgt=dtgt)
    n = 0
    for base in ("cross_entropy", "focal"):
        for ldt in (np.int64, np.uint8):
            for variant in ("density", "from_depth", "plain", "density_depth_target"):
                fn = model.FogDensityAwareLoss(base_loss=base)
                preds = {"segmentation": torch.from_numpy(logits).requires_grad_(True)}
                tg = {"label": torch.from_numpy(lab.astype(ldt))}
                fd = None
                if variant in ("density", "density_depth_target"):
                    fd = torch.from_numpy(dens)
                if variant in ("from_depth", "density_depth_target"):
                    preds["depth"] = torch.from_numpy(dpred)
                if variant == "density_depth_target":
                    tg["depth"] = torch.from_numpy(dtgt)
                r = fn(preds, tg, fd)
                r["total_loss"].backward()
                out[f"loss_total{n}"] = np.float32(r["total_loss"].item())
                out[f"loss_seg{n}"] = np.float32(r["segmentation_loss"].item())
                dl = r["depth_loss"]
                out[f"loss_depth{n}"] = np.float32(dl.item() if isinstance(dl, torch.Tensor) else dl)
                if ldt is np.int64:
                    out[f"loss_grad{n}"] = preds["segmentation"].grad.numpy()
                out[f"loss_cfg{n}"] = np.array([base, np.dtype(ldt).name, variant])
                n += 1
    out["n_loss"] = np.int64(n)
    fn = model.FogDensityAwareLoss()
    out["density_from_depth"] = fn._estimate_fog_density_from_depth(torch.from_numpy(dpred[:, 0])).numpy()

    # ---- A8 head arithmetic: F.interpolate -> segmentation_head (model.py:152-158, 209-214)
    torch.manual_seed(5)
    cin, cmid, cout = 16, 24, 7
    head = torch.nn.Sequential(
        torch.nn.Conv2d(cin, cmid, kernel_size=3, padding=1), torch.nn.BatchNorm2d(cmid),
        torch.nn.ReLU(inplace=True), torch.nn.Dropout2d(0.1), torch.nn.Conv2d(cmid, cout, kernel_size=1)).eval()
    with torch.no_grad():
        head[1].running_mean.uniform_(-0.5, 0.5); head[1].running_var.uniform_(0.5, 2.0)
        head[1].weight.uniform_(0.5, 1.5); head[1].bias.uniform_(-0.3, 0.3)
        feat = torch.randn(1, cin, 3, 5)
        H, W = 96, 160
        up = torch.nn.functional.interpolate(feat, size=(H, W), mode="bilinear", align_corners=False)
        y = head(up)
    out.update(head_feat=feat.numpy(), head_w1=head[0].weight.detach().numpy(), head_b1=head[0].bias.detach().numpy(),
               head_bn_w=head[1].weight.detach().numpy(), head_bn_b=head[1].bias.detach().numpy(),
               head_bn_mean=head[1].running_mean.numpy(), head_bn_var=head[1].running_var.numpy(),
               head_bn_eps=np.float64(head[1].eps), head_w2=head[4].weight.detach().numpy().reshape(cout, cmid),
               head_b2=head[4].bias.detach().numpy(), head_out=y.numpy(), head_size=np.array([H, W]))
    # DepthEstimationHead (model.py:42-52) on a small feature map
    torch.manual_seed(6)
    dh = model.DepthEstimationHead(in_channels=8, hidden_channels=16).eval()
    x = torch.randn(1, 8, 10, 12)
    with torch.no_grad():
        out["dhead_out"] = dh(x).numpy()
    out["dhead_in"] = x.numpy()
    for kname, v in dh.state_dict().items():
        out["dhead_sd." + kname] = v.numpy()
    # the same head at a Winograd-eligible size (in 64, hidden 128 -> 64): the shapes the HIP depth-head kernels take.
    # Its parameters are not stored (0.6 MB of noise): they are re-drawn by `dhead64_params` from a seeded generator,
    # here and in the test (tests/test_gpu_kernels.py imports this recipe by name, not the reference).
    dh64 = model.DepthEstimationHead(in_channels=64, hidden_channels=128).eval()
    dh64.load_state_dict(dhead64_params(dh64.state_dict()))
    x64 = torch.randn(2, 64, 12, 20, generator=torch.Generator().manual_seed(64))
    with torch.no_grad():
        out["dhead64_out"] = dh64(x64).numpy()
    out["dhead64_in"] = x64.numpy()
    np.savez_compressed(OUT / "model.npz", **out)


def dhead64_params(template):
    """Deterministic parameters for a DepthEstimationHead state_dict (same keys / shapes as `template`): conv weights
    N(0, 2/fan_in), small biases, BatchNorm with non-trivial affine terms and running statistics."""
    g = torch.Generator().manual_seed(640)
    sd = {}
    for k, v in template.items():
        if k.endswith("num_batches_tracked"):
            sd[k] = torch.zeros_like(v)
        elif k.endswith("running_var"):
            sd[k] = torch.rand(v.shape, generator=g) * 1.5 + 0.5
        elif k.endswith("running_mean"):
            sd[k] = torch.rand(v.shape, generator=g) - 0.5
        elif v.dim() == 4:
            sd[k] = torch.randn(v.shape, generator=g) * (2.0 / (v.shape[1] * v.shape[2] * v.shape[3])) ** 0.5
        elif k.endswith("weight"):                                  # BatchNorm gamma
            sd[k] = torch.rand(v.shape, generator=g) + 0.5
        else:                                                       # conv / BatchNorm biases
            sd[k] = (torch.rand(v.shape, generator=g) - 0.5) * 0.6
    return sd


def gen_trainer():
    """AdverseWeatherTrainer._estimate_fog_density (trainer.py:480-511): it never reads `self`, so it is called
    unbound.  trainer.py imports torch.utils.tensorboard (absent here): an inert module entry stands in for it, as for
    cv2 / smp above.  The uniform field the reference drew is recovered by replaying torch's CPU generator."""
    inert = {"torch.utils.tensorboard": ["SummaryWriter"], "torchvision.transforms": [],
             "albumentations": ["Compose", "HorizontalFlip", "RandomBrightnessContrast", "Normalize"],
             "albumentations.pytorch": ["ToTensorV2"]}
    for name, attrs in inert.items():                              # absent packages the import chain names and never calls here
        if name not in sys.modules:
            m = types.ModuleType(name)
            for a in attrs:
                setattr(m, a, object)
            sys.modules[name] = m
    sys.modules["torchvision"].transforms = sys.modules["torchvision.transforms"]
    from adverse_weather_semantic_segmentation_robustness_benchmark.training import trainer as ref_trainer
    out = {}
    conds = ["fog", "clean", "rain", "night", "snow", "fog"]
    h, w = 12, 20
    batch = {"weather_condition": conds, "image": torch.zeros(len(conds), 3, h, w)}
    torch.manual_seed(11)
    out["density"] = ref_trainer.AdverseWeatherTrainer._estimate_fog_density(None, batch).numpy()
    torch.manual_seed(11)
    out["uniform"] = torch.stack([torch.rand(h, w) for _ in conds]).numpy()
    out["conditions"] = np.array(conds)
    assert ref_trainer.AdverseWeatherTrainer._estimate_fog_density(None, {"weather_condition": [], "image": batch["image"]}) is None
    np.savez_compressed(OUT / "trainer.npz", **out)


def gen_depth_estimate():
    """DepthEstimationPreprocessor (preprocessing.py:325-367) needs the real cv2 for its first two
    steps, so the reference function cannot run here (PARITY UNPINNED for those steps).  These
    fixtures are NOT produced by the reference: gray / Laplacian restate OpenCV 4.x's published
    integer arithmetic in numpy; the float64 ladder of :340-363 is written with the reference's own
    numpy expressions and the smoothing is scipy.ndimage.gaussian_filter itself (:366), i.e. the
    libraries the reference calls.  They pin the C oracle's float64 ladder + Gaussian."""
    from scipy.ndimage import gaussian_filter
    out = {}
    for k, (h, w, seed) in enumerate([(32, 64, 1), (17, 23, 2), (64, 128, 3), (5, 9, 4)]):
        rs = np.random.RandomState(7000 + seed)
        img = rs.randint(0, 256, (h, w, 3), dtype=np.uint8)
        if k == 2:
            img[: h // 2] = 200                                    # flat sky: zero texture rows
        i32 = img.astype(np.int64)
        gray = (i32[..., 0] * 9798 + i32[..., 1] * 19235 + i32[..., 2] * 3735 + (1 << 14)) >> 15
        g = np.pad(gray, 1, mode="reflect")                       # numpy 'reflect' == BORDER_REFLECT_101
        texture = (g[:-2, 1:-1] + g[2:, 1:-1] + g[1:-1, :-2] + g[1:-1, 2:] - 4 * gray).astype(np.float64)
        sky_mask = np.zeros((h, w), dtype=np.float32); sky_mask[: h // 3, :] = 1.0
        road_mask = np.zeros((h, w), dtype=np.float32); road_mask[h // 2:, :] = 1.0
        y_coords = np.arange(h)[:, np.newaxis] / h
        base_depth = np.tile(y_coords * 0.8 + 0.2, (1, w))
        depth = base_depth.copy()
        depth[sky_mask > 0] = 1.0
        depth[road_mask > 0] *= 0.5
        texture_strength = np.abs(texture) / (np.max(np.abs(texture)) + 1e-8)
        depth = np.clip(depth + (-0.3 * texture_strength), 0, 1)
        depth = gaussian_filter(depth, sigma=2)
        out[f"img{k}"] = img
        out[f"depth{k}"] = depth
    np.savez_compressed(OUT / "depth.npz", **out)


if __name__ == "__main__":
    metrics, pre, model = load_reference()
    gen_weather(pre)
    gen_metrics(metrics)
    gen_model(model)
    gen_depth_estimate()
    gen_trainer()
    for f in sorted(OUT.glob("*.npz")):
        print(f.name, f.stat().st_size)
