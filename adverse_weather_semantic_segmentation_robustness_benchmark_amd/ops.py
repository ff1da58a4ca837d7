"""Functional host wrappers over the C ABI (``include/awseg.h``) for torch device tensors.

Each function cites the reference function it replaces (PKG = the reference package
``adverse_weather_semantic_segmentation_robustness_benchmark``).  All of them launch on the
current torch stream, never synchronise, and raise if handed CPU tensors.
"""
from __future__ import annotations

import os

import ctypes as C
from typing import Optional, Sequence

import numpy as np
import torch

from . import _native as N

IMAGENET_MEAN = np.array([0.485, 0.456, 0.406], dtype=np.float32)   # PKG/data/loader.py:196
IMAGENET_STD = np.array([0.229, 0.224, 0.225], dtype=np.float32)
NIGHT_GAINS = np.array([0.8, 0.85, 1.2], dtype=np.float32)          # PKG/data/preprocessing.py:55
CONDITIONS = ("clean", "fog", "rain", "snow", "night")               # PKG/evaluation/metrics.py:491


def gaussian_taps(sigma: float = 2.0, truncate: float = 4.0) -> np.ndarray:
    """The 17 float64 weights scipy.ndimage.gaussian_filter(sigma=2) uses
    (PKG/data/preprocessing.py:243): same numpy expressions as scipy's _gaussian_kernel1d."""
    radius = int(truncate * float(sigma) + 0.5)
    x = np.arange(-radius, radius + 1)
    phi = np.exp(-0.5 / (sigma * sigma) * x ** 2)
    return phi / phi.sum()


_TAPS = gaussian_taps()


def _mean_std(mean, std):
    m = np.ascontiguousarray(IMAGENET_MEAN if mean is None else mean, dtype=np.float32)
    s = np.ascontiguousarray(IMAGENET_STD if std is None else std, dtype=np.float32)
    return m, s


# ----------------------------------------------------------------------------- metrics
def new_counts(num_classes: int, device, n_slots: int = 1) -> torch.Tensor:
    return torch.zeros(n_slots, num_classes * num_classes, dtype=torch.int64, device=device)


def confusion_accumulate(pred: torch.Tensor, label: torch.Tensor, num_classes: int, counts: torch.Tensor,
                         oob: torch.Tensor, ignore_index: int = 255, wrap_u8: Optional[bool] = None) -> None:
    """PKG/evaluation/metrics.py:54-71.  `wrap_u8` defaults to the reference's behaviour: on
    for uint8 label tensors (their `targets*C` wraps mod 256), off for int64."""
    pred, label = pred.contiguous().view(-1), label.contiguous().view(-1)
    if wrap_u8 is None:
        wrap_u8 = label.dtype == torch.uint8
    n = pred.numel()
    ws = N.workspace.get(pred.device, N.lib().awseg_metrics_workspace(1, num_classes, n))
    N.call("awseg_confusion_accumulate", N.ptr(pred), N.label_dtype(pred), N.ptr(label), N.label_dtype(label), n,
                                               num_classes, ignore_index, int(wrap_u8), N.ptr(counts), N.ptr(oob),
                                               N.ptr(ws), N.stream())


def argmax(logits: torch.Tensor, out_dtype=torch.int64) -> torch.Tensor:
    """logits.argmax(dim=1), REF/scripts/evaluate.py:179."""
    logits = logits.contiguous()
    b, c = logits.shape[0], logits.shape[1]
    hw = logits[0, 0].numel()
    pred = torch.empty((b,) + tuple(logits.shape[2:]), dtype=out_dtype, device=logits.device)
    N.call("awseg_argmax", N.ptr(logits), b, c, hw, N.ptr(pred), N.label_dtype(pred), N.stream())
    return pred


def combine_argmax_confusion(seg1: torch.Tensor, seg2: Optional[torch.Tensor], mode: int,
                             weights: Optional[torch.Tensor] = None, temperature: Optional[torch.Tensor] = None,
                             want_logits: bool = True, want_pred: bool = False, pred_dtype=torch.int64,
                             label: Optional[torch.Tensor] = None, counts: Optional[torch.Tensor] = None,
                             oob: Optional[torch.Tensor] = None, cond: Optional[torch.Tensor] = None,
                             ignore_index: int = 255, wrap_u8: Optional[bool] = None):
    """PKG/models/model.py:443-462 (+ argmax evaluate.py:179, + confusion metrics.py:54-71).
    seg2 None -> single-model argmax(+confusion)."""
    seg1 = seg1.contiguous()
    b, c = seg1.shape[0], seg1.shape[1]
    hw = seg1[0, 0].numel()
    out = torch.empty_like(seg1) if (want_logits and seg2 is not None) else None
    pred = torch.empty((b,) + tuple(seg1.shape[2:]), dtype=pred_dtype, device=seg1.device) if want_pred else None
    ws = None
    ldt = N.U8
    if label is not None:
        label = label.contiguous()
        ldt = N.label_dtype(label)
        if wrap_u8 is None:
            wrap_u8 = label.dtype == torch.uint8
        ws = N.workspace.get(seg1.device, N.lib().awseg_metrics_workspace(b, c, hw))
    n_slots = 0 if counts is None else counts.shape[0]
    pdt = N.label_dtype(pred) if pred is not None else N.U8
    if seg2 is None:
        N.call("awseg_argmax_confusion", N.ptr(seg1), b, c, hw, N.ptr(pred), pdt, N.ptr(label), ldt, ignore_index,
                                            int(bool(wrap_u8)), N.ptr(cond), N.ptr(counts), n_slots, N.ptr(oob),
                                            N.ptr(ws), N.stream())
        return seg1, pred
    seg2 = seg2.contiguous()
    N.call("awseg_combine_argmax_confusion", N.ptr(seg1), N.ptr(seg2), b, c, hw, mode, N.ptr(weights),
                                                N.ptr(temperature), N.ptr(out), N.ptr(pred), pdt, N.ptr(label), ldt,
                                                ignore_index, int(bool(wrap_u8)), N.ptr(cond), N.ptr(counts), n_slots,
                                                N.ptr(oob), N.ptr(ws), N.stream())
    return out, pred


ECE_BIN_DTYPE = np.dtype([("count", "<i8"), ("sum_conf", "<f8"), ("sum_correct", "<i8")])


def new_ece_bins(n_bins: int, device, n_slots: int = 1) -> torch.Tensor:
    return torch.zeros(n_slots, n_bins, 3, dtype=torch.int64, device=device)   # 24 B / bin, raw


def ece_accumulate(logits: torch.Tensor, label: torch.Tensor, bins: torch.Tensor, edges: torch.Tensor,
                   cond: Optional[torch.Tensor] = None) -> None:
    """PKG/evaluation/metrics.py:161-194, per-pixel part, accumulated on device."""
    logits, label = logits.contiguous(), label.contiguous()
    b, c = logits.shape[0], logits.shape[1]
    hw = logits[0, 0].numel()
    ws = N.workspace.get(logits.device, N.lib().awseg_metrics_workspace(b, c, hw))
    N.call("awseg_ece_accumulate", N.ptr(logits), b, c, hw, N.ptr(label), N.label_dtype(label), N.ptr(cond),
                                         N.ptr(edges), bins.shape[1], N.ptr(bins), bins.shape[0], N.ptr(ws), N.stream())


def ensemble_eval_stats(seg1: torch.Tensor, seg2: torch.Tensor, mode: int, weights, temperature, label: torch.Tensor,
                        cond, edges: torch.Tensor, ece_bins: torch.Tensor, auroc_hist: torch.Tensor, lo: float, hi: float) -> None:
    """ECE accumulators of the combined logits + disagreement-score histogram by error flag, one pass
    over the member logits (REF/scripts/evaluate.py:230-255)."""
    seg1, seg2, label = seg1.contiguous(), seg2.contiguous(), label.contiguous()
    b, c = seg1.shape[0], seg1.shape[1]
    hw = seg1[0, 0].numel()
    ws = N.workspace.get(seg1.device, N.lib().awseg_metrics_workspace(b, c, hw))
    N.call("awseg_ensemble_eval_stats", N.ptr(seg1), N.ptr(seg2), b, c, hw, mode, N.ptr(weights), N.ptr(temperature), N.ptr(label),
           N.label_dtype(label), N.ptr(cond), N.ptr(edges), ece_bins.shape[1], N.ptr(ece_bins), ece_bins.shape[0],
           N.ptr(auroc_hist), auroc_hist.shape[1], float(lo), float(hi), N.ptr(ws), N.stream())


def combine_confusion_stats(seg1: torch.Tensor, seg2: torch.Tensor, mode: int, weights, temperature, label: torch.Tensor, cond,
                            counts: torch.Tensor, oob: torch.Tensor, edges: torch.Tensor, ece_bins: torch.Tensor,
                            auroc_hist: torch.Tensor, lo: float, hi: float, ignore_index: int = 255, wrap_u8: Optional[bool] = None) -> None:
    """combine_argmax_confusion (confusion counters only) + ensemble_eval_stats in one pass over the member logits."""
    seg1, seg2, label = seg1.contiguous(), seg2.contiguous(), label.contiguous()
    b, c = seg1.shape[0], seg1.shape[1]
    hw = seg1[0, 0].numel()
    if wrap_u8 is None:
        wrap_u8 = label.dtype == torch.uint8
    ws = N.workspace.get(seg1.device, N.lib().awseg_metrics_workspace(b, c, hw))
    N.call("awseg_combine_confusion_stats", N.ptr(seg1), N.ptr(seg2), b, c, hw, mode, N.ptr(weights), N.ptr(temperature), N.ptr(label),
           N.label_dtype(label), int(ignore_index), int(bool(wrap_u8)), N.ptr(cond), N.ptr(counts), counts.shape[0], N.ptr(oob),
           N.ptr(edges), ece_bins.shape[1], N.ptr(ece_bins), ece_bins.shape[0], N.ptr(auroc_hist), auroc_hist.shape[1], float(lo),
           float(hi), N.ptr(ws), N.stream())


ECE_CONF_UNIT = 2.0 ** -30     # the device keeps the confidence sums in fixed point (int64, units of 2^-30): exact, order-independent


def ece_bins_to_numpy(bins: torch.Tensor) -> np.ndarray:
    """device bins int64 [slots, n_bins, {count, sum_conf_q30, sum_correct}] -> structured array with float64 sum_conf."""
    raw = bins.cpu().numpy()
    out = np.zeros(raw.shape[:2], dtype=ECE_BIN_DTYPE)
    out["count"], out["sum_conf"], out["sum_correct"] = raw[..., 0], raw[..., 1].astype(np.float64) * ECE_CONF_UNIT, raw[..., 2]
    return out


# ----------------------------------------------------------------------------- A7
def normalize(imgs: torch.Tensor, out: Optional[torch.Tensor] = None, sel: Optional[torch.Tensor] = None,
              mean=None, std=None) -> torch.Tensor:
    """PKG/data/loader.py:195-198: uint8 [B,H,W,3] -> float32 [B,3,H,W]."""
    imgs = imgs.contiguous()
    b, h, w, _ = imgs.shape
    if out is None:
        out = torch.empty(b, 3, h, w, dtype=torch.float32, device=imgs.device)
    m, s = _mean_std(mean, std)
    N.call("awseg_normalize", N.ptr(imgs), b, h, w, N.ptr(sel), 0 if sel is None else sel.numel(), N.host(m),
                                    N.host(s), N.ptr(out), N.stream())
    return out


# ----------------------------------------------------------------------------- weather
def _jobs(dtype, n):
    return np.zeros(n, dtype=dtype)


def fog_jobs(images: Sequence[int], intensities: Sequence[float], seeds: Optional[Sequence[int]] = None) -> np.ndarray:
    """beta / A from intensity exactly as PKG/data/preprocessing.py:110-114 (Python float math)."""
    j = _jobs(N.FOG_JOB, len(images))
    for k, (im, inten) in enumerate(zip(images, intensities)):
        inten = float(inten)
        j[k]["image"] = im
        j[k]["beta"] = 0.005 + inten * (0.05 - 0.005)
        j[k]["atmos"] = 0.7 + inten * (1.0 - 0.7)
        j[k]["seed"] = 0 if seeds is None else int(seeds[k]) & 0xFFFFFFFFFFFFFFFF
    return j


def night_jobs(images, brightness, intensities, seeds=None) -> np.ndarray:
    j = _jobs(N.NIGHT_JOB, len(images))
    for k in range(len(images)):
        j[k]["image"] = images[k]
        j[k]["brightness"] = float(brightness[k])
        j[k]["intensity"] = float(intensities[k])
        j[k]["seed"] = 0 if seeds is None else int(seeds[k]) & 0xFFFFFFFFFFFFFFFF
    return j


def prim_jobs(images, intensities, prim_lists, ksizes=None):
    """-> (jobs, concatenated int32 primitive array)."""
    j = _jobs(N.PRIM_JOB, len(images))
    off = 0
    for k in range(len(images)):
        j[k]["image"] = images[k]
        j[k]["prim_offset"] = off
        j[k]["prim_count"] = len(prim_lists[k])
        j[k]["blur_ksize"] = 3 if ksizes is None else int(ksizes[k])
        j[k]["intensity"] = float(intensities[k])
        off += len(prim_lists[k])
    width = prim_lists[0].shape[1] if len(prim_lists) and prim_lists[0].ndim == 2 else 1
    prims = np.concatenate([np.asarray(p, dtype=np.int32).reshape(-1, width) for p in prim_lists], axis=0) \
        if len(prim_lists) else np.zeros((0, width), np.int32)
    if prims.shape[0] == 0:
        prims = np.zeros((1, max(width, 1)), np.int32)
    return j, np.ascontiguousarray(prims)


def synthetic_depth(h: int, w: int, jobs: np.ndarray, device, noise: Optional[torch.Tensor] = None) -> torch.Tensor:
    """PKG/data/preprocessing.py:227-248 -> float64 [n_jobs,H,W]."""
    jd = N.host_jobs(jobs)
    out = torch.empty(len(jobs), h, w, dtype=torch.float64, device=device)
    N.call("awseg_synthetic_depth", h, w, jd, len(jobs), N.ptr(noise), N.host(_TAPS), N.ptr(out), N.stream())
    return out


def lut3_apply(imgs: torch.Tensor, luts: torch.Tensor, lut_of: torch.Tensor, out: Optional[torch.Tensor] = None) -> torch.Tensor:
    """out[b,...,c] = luts[lut_of[b], c, imgs[b,...,c]] for uint8 [B,H,W,3] frames (lut_of < 0: unchanged)."""
    imgs = imgs.contiguous()
    b, h, w, _ = imgs.shape
    if out is None:
        out = torch.empty_like(imgs)
    N.call("awseg_lut3_apply", N.ptr(imgs), b, h, w, N.ptr(luts.contiguous()), int(luts.shape[0]), N.ptr(lut_of), N.ptr(out), N.stream())
    return out


def local_contrast(imgs: torch.Tensor) -> torch.Tensor:
    """sqrt(box5((gray - box5(gray))^2)) of uint8 [B,H,W,3] frames -> float32 [B,H,W]
    (PKG/data/preprocessing.py:270-278)."""
    imgs = imgs.contiguous()
    b, h, w, _ = imgs.shape
    out = torch.empty(b, h, w, dtype=torch.float32, device=imgs.device)
    N.call("awseg_local_contrast", N.ptr(imgs), b, h, w, N.ptr(out), N.stream())
    return out


def depth_estimate(imgs: torch.Tensor, out: Optional[torch.Tensor] = None, dtype=torch.float32) -> torch.Tensor:
    """DepthEstimationPreprocessor.estimate_depth (PKG/data/preprocessing.py:304-367) for a uint8
    [B,H,W,3] batch -> [B,H,W] depth target in [0,1]; float64 (reference dtype) or float32 (what
    the loader hands on, loader.py:290)."""
    imgs = imgs.contiguous()
    b, h, w, _ = imgs.shape
    if out is None:
        out = torch.empty(b, h, w, dtype=dtype, device=imgs.device)
    ws = torch.empty(max(1, b), dtype=torch.int32, device=imgs.device)
    o64 = out if out.dtype == torch.float64 else None
    o32 = out if out.dtype == torch.float32 else None
    N.call("awseg_depth_estimate", N.ptr(imgs), b, h, w, N.host(_TAPS), N.ptr(ws), N.ptr(o64), N.ptr(o32), N.stream())
    return out


def fog(imgs: torch.Tensor, jobs: np.ndarray, noise: Optional[torch.Tensor] = None, depth: Optional[torch.Tensor] = None,
        out: Optional[torch.Tensor] = None, norm_out: Optional[torch.Tensor] = None, depth_out: Optional[torch.Tensor] = None,
        mean=None, std=None) -> None:
    """_apply_fog (PKG/data/preprocessing.py:94-123).  depth given -> two-step form; else fused
    depth+fog with `noise` (parity mode) or in-kernel Philox (noise None)."""
    imgs = imgs.contiguous()
    _, h, w, _ = imgs.shape
    jd = N.host_jobs(jobs)
    m, s = _mean_std(mean, std)
    if depth is not None:
        N.call("awseg_fog_apply", N.ptr(imgs), h, w, jd, len(jobs), N.ptr(depth.contiguous()), N.ptr(out),
                                        N.ptr(norm_out), N.host(m), N.host(s), N.stream())
    else:
        N.call("awseg_fog_fused", N.ptr(imgs), h, w, jd, len(jobs), N.ptr(noise), N.host(_TAPS), N.ptr(out),
                                        N.ptr(norm_out), N.ptr(depth_out), N.host(m), N.host(s), N.stream())


def night(imgs: torch.Tensor, jobs: np.ndarray, noise: Optional[torch.Tensor] = None, out=None, norm_out=None,
          mean=None, std=None) -> None:
    """_apply_night (PKG/data/preprocessing.py:204-225)."""
    imgs = imgs.contiguous()
    _, h, w, _ = imgs.shape
    jd = N.host_jobs(jobs)
    m, s = _mean_std(mean, std)
    N.call("awseg_night_apply", N.ptr(imgs), h, w, jd, len(jobs), N.ptr(noise), N.host(NIGHT_GAINS), N.ptr(out),
                                      N.ptr(norm_out), N.host(m), N.host(s), N.stream())


WEATHER_BATCH = os.environ.get("AWSEG_WEATHER_BATCH", "1") != "0"   # one launch for a batch of mixed conditions (throughput mode)
WEATHER_BATCH_MAX = 16


def weather_batch_ok(h: int, w: int) -> bool:
    return WEATHER_BATCH and w % 4 == 0 and w >= 16


def weather_batch(imgs: torch.Tensor, jobs: np.ndarray, rain_drops: Optional[np.ndarray], snow_flakes: Optional[np.ndarray],
                  norm_out: torch.Tensor, out: Optional[torch.Tensor] = None, mean=None, std=None) -> bool:
    """apply_weather_effect + Normalize for <= 16 frames of mixed conditions in ONE launch (awseg_weather_batch; in-kernel Philox noise).
    jobs: np.ndarray of N.WEATHER_JOB.  Returns False when the launcher declines the geometry (the caller uses the per-kind calls)."""
    imgs = imgs.contiguous()
    _, h, w, _ = imgs.shape
    n_cov = int(((jobs["kind"] == N.WEATHER_RAIN) | (jobs["kind"] == N.WEATHER_SNOW)).sum())
    ws = N.workspace.get(imgs.device, N.lib().awseg_streak_workspace(n_cov, h, w), tag="streak") if n_cov else None
    rd = None if rain_drops is None else torch.from_numpy(np.ascontiguousarray(rain_drops, dtype=np.int32)).to(imgs.device, non_blocking=True)
    sf = None if snow_flakes is None else torch.from_numpy(np.ascontiguousarray(snow_flakes, dtype=np.int32)).to(imgs.device, non_blocking=True)
    m, s = _mean_std(mean, std)
    rc = N.try_call("awseg_weather_batch", N.ptr(imgs), h, w, N.host_jobs(jobs), len(jobs), N.ptr(rd), N.ptr(sf), N.host(_TAPS), N.host(NIGHT_GAINS),
                    N.ptr(out), N.ptr(norm_out), N.host(m), N.host(s), N.ptr(ws), N.stream())
    return rc == 0


def _streak_workspace(imgs: torch.Tensor, n_jobs: int, h: int, w: int, prepass: bool):
    """Coverage-map scratch of the rain / snow launchers (awseg_streak_workspace); None = rasterise inside the blur kernel."""
    if not prepass or n_jobs < 1:
        return None
    return N.workspace.get(imgs.device, N.lib().awseg_streak_workspace(n_jobs, h, w), tag="streak")


def rain(imgs: torch.Tensor, jobs: np.ndarray, drops: np.ndarray, out=None, norm_out=None, mean=None, std=None,
         prepass: bool = True) -> None:
    """_apply_rain (PKG/data/preprocessing.py:125-168).  prepass: rasterise the drops once per frame into a coverage bit map
    (a second small kernel) instead of in every tile they touch — same bytes out."""
    imgs = imgs.contiguous()
    _, h, w, _ = imgs.shape
    jd = N.host_jobs(jobs)
    pd = torch.from_numpy(np.ascontiguousarray(drops, dtype=np.int32)).to(imgs.device, non_blocking=True)
    m, s = _mean_std(mean, std)
    N.call("awseg_rain_apply", N.ptr(imgs), h, w, jd, len(jobs), N.ptr(pd), N.ptr(out), N.ptr(norm_out),
                                     N.host(m), N.host(s), N.ptr(_streak_workspace(imgs, len(jobs), h, w, prepass)), N.stream())


def snow(imgs: torch.Tensor, jobs: np.ndarray, flakes: np.ndarray, out=None, norm_out=None, mean=None, std=None,
         prepass: bool = True) -> None:
    """_apply_snow (PKG/data/preprocessing.py:170-202)."""
    imgs = imgs.contiguous()
    _, h, w, _ = imgs.shape
    jd = N.host_jobs(jobs)
    pd = torch.from_numpy(np.ascontiguousarray(flakes, dtype=np.int32)).to(imgs.device, non_blocking=True)
    m, s = _mean_std(mean, std)
    N.call("awseg_snow_apply", N.ptr(imgs), h, w, jd, len(jobs), N.ptr(pd), N.ptr(out), N.ptr(norm_out),
                                     N.host(m), N.host(s), N.ptr(_streak_workspace(imgs, len(jobs), h, w, prepass)), N.stream())


_DENSITY_TABLE = {"fog": (0.5, 0.5), "rain": (0.3, 0.2), "snow": (0.3, 0.2)}   # trainer.py:501-509, else (0.1, 0)


def fog_density_field(conditions: Sequence[str], h: int, w: int, device, seed: int,
                      uniform: Optional[torch.Tensor] = None) -> torch.Tensor:
    """AdverseWeatherTrainer._estimate_fog_density (PKG/training/trainer.py:480-511) on device.  `uniform` float32
    [B,h,w] (device): the host's torch.rand draws -> bit-identical to the reference (parity mode); None: in-kernel
    Philox keyed by `seed` (throughput mode)."""
    so = np.array([_DENSITY_TABLE.get(str(c), (0.1, 0.0)) for c in conditions], dtype=np.float32)
    sod = torch.from_numpy(so).to(device, non_blocking=True)
    out = torch.empty(len(conditions), h, w, dtype=torch.float32, device=device)
    if uniform is not None:
        uniform = uniform.to(device=device, dtype=torch.float32).contiguous()
        if tuple(uniform.shape) != (len(conditions), h, w):
            raise ValueError(f"uniform field must be [{len(conditions)},{h},{w}], got {tuple(uniform.shape)}")
    N.call("awseg_fog_density_field", N.ptr(sod), len(conditions), h * w, seed & 0xFFFFFFFFFFFFFFFF, N.ptr(uniform), N.ptr(out),
                                            N.stream())
    return out


# ----------------------------------------------------------------------------- loss
def fog_ce_forward(logits: torch.Tensor, label: torch.Tensor, density: Optional[torch.Tensor], focal: bool,
                   sensitivity: float, oob: torch.Tensor, want_pixel: bool = False):
    """FogDensityAwareLoss seg term, PKG/models/model.py:577-587 + mean :610 -> float32[1]."""
    logits, label = logits.contiguous(), label.contiguous()
    b, c = logits.shape[0], logits.shape[1]
    hw = logits[0, 0].numel()
    n_part = N.lib().awseg_loss_partials(b, hw)
    partials = N.workspace.get(logits.device, n_part * 8, "loss").view(torch.float64)
    mean = torch.empty(1, dtype=torch.float32, device=logits.device)
    pix = torch.empty((b,) + tuple(logits.shape[2:]), dtype=torch.float32, device=logits.device) if want_pixel else None
    dens = None if density is None else density.contiguous()
    N.call("awseg_fog_ce_forward", N.ptr(logits), N.ptr(label), N.label_dtype(label), N.ptr(dens), b, c, hw,
                                         N.LOSS_FOCAL if focal else N.LOSS_CE, float(sensitivity), N.ptr(pix),
                                         N.ptr(partials), N.ptr(mean), N.ptr(oob), N.stream())
    return mean, pix


def fog_ce_backward(logits, label, density, focal: bool, sensitivity: float, grad_scale: torch.Tensor) -> torch.Tensor:
    logits, label = logits.contiguous(), label.contiguous()
    b, c = logits.shape[0], logits.shape[1]
    hw = logits[0, 0].numel()
    grad = torch.empty_like(logits)
    dens = None if density is None else density.contiguous()
    gs = grad_scale.to(torch.float32).reshape(1).contiguous()
    N.call("awseg_fog_ce_backward", N.ptr(logits), N.ptr(label), N.label_dtype(label), N.ptr(dens), b, c, hw,
                                          N.LOSS_FOCAL if focal else N.LOSS_CE, float(sensitivity), N.ptr(gs), N.ptr(grad),
                                          N.stream())
    return grad


def fog_density_from_depth(depth: torch.Tensor) -> torch.Tensor:
    """FogDensityAwareLoss._estimate_fog_density_from_depth, PKG/models/model.py:644-677."""
    depth = depth.contiguous().float()
    b, h, w = depth.shape
    ws = N.workspace.get(depth.device, N.lib().awseg_density_workspace(b, h * w), "density")
    out = torch.empty_like(depth)
    N.call("awseg_fog_density_from_depth", N.ptr(depth), b, h, w, N.ptr(out), N.ptr(ws), N.stream())
    return out


# ----------------------------------------------------------------------------- heads
HEAD_SPLIT = os.environ.get("AWSEG_HEAD_SPLIT", "1") != "0"


def segformer_head_fused(g9: torch.Tensor, scale, shift, w2, b2, height: int, width: int, split: Optional[bool] = None) -> torch.Tensor:
    """Upsample-free SegFormer head (PKG/models/model.py:209-214).  g9 [B,h,w,9,Cmid].
    split: True = the split-operand f16-MFMA kernel where it applies (scale folded into g9, Cmid 128/256, tile geometry),
    False = the float32-input MFMA kernel; None takes the process default (HEAD_SPLIT, env AWSEG_HEAD_SPLIT=0/1)."""
    g9 = g9.contiguous()
    b, h, w, nine, cmid = g9.shape
    assert nine == 9
    cout = w2.shape[0]
    out = torch.empty(b, cout, height, width, dtype=torch.float32, device=g9.device)
    if (HEAD_SPLIT if split is None else split) and scale is None and cmid in (128, 256):
        if N.try_call("awseg_segformer_head_fused_split", N.ptr(g9), b, cmid, h, w, height, width, None, N.ptr(shift.contiguous()),
                      N.ptr(w2.contiguous()), N.ptr(b2.contiguous()), cout, N.ptr(out), N.stream()) == 0:
            return out
    N.call("awseg_segformer_head_fused", N.ptr(g9), b, cmid, h, w, height, width, N.ptr(None if scale is None else scale.contiguous()),
                                               N.ptr(shift.contiguous()), N.ptr(w2.contiguous()), N.ptr(b2.contiguous()),
                                               cout, N.ptr(out), N.stream())
    return out


def upconv3x3_bn_relu(g9: torch.Tensor, scale, shift, height: int, width: int, channels_last: bool = False) -> torch.Tensor:
    """relu(bn(conv3x3(interpolate(f)))) at full resolution without the upsampled tensor: the first
    stage of the fused head only (DepthEstimationHead's first 3x3 on the SegFormer branch)."""
    g9 = g9.contiguous()
    b, h, w, nine, cmid = g9.shape
    assert nine == 9
    shape = (b, height, width, cmid) if channels_last else (b, cmid, height, width)
    out = torch.empty(shape, dtype=torch.float32, device=g9.device)
    N.call("awseg_upconv3x3_bn_relu", N.ptr(g9), b, cmid, h, w, height, width, N.ptr(None if scale is None else scale.contiguous()),
           N.ptr(shift.contiguous()), N.ptr(out), int(channels_last), N.stream())
    return out.permute(0, 3, 1, 2) if channels_last else out          # logical NCHW either way


class _UpConv3x3(torch.autograd.Function):
    """conv3x3(F.interpolate(f, (H, W), bilinear, align_corners=False)) with zero padding, for TRAINING, without the
    upsampled tensor and without a full-resolution convolution (PKG/models/model.py:209-214, :219-221): forward = one small
    GEMM at the encoder's resolution + awseg_upconv3x3_linear; backward = awseg_upconv3x3_adjoint + two small GEMMs."""

    @staticmethod
    def forward(ctx, tok, weight, bias, height, width):
        B, h, w, cin = tok.shape
        cmid = weight.shape[0]
        w1r = weight.permute(1, 2, 3, 0).reshape(cin, 9 * cmid)                     # [Cin, (ky, kx, cout)]
        tok2 = tok.reshape(B * h * w, cin)
        g9 = (tok2 @ w1r).view(B, h, w, 9, cmid).contiguous()
        # NCHW out: the layers behind (BatchNorm, Conv2d) then run on MIOpen's NCHW kernels like the as-written graph — its
        # heuristic pick for channels_last weight gradients is a CK kernel 50x slower (profiles/r02_train_step_kernels.csv)
        z = torch.empty(B, cmid, height, width, dtype=torch.float32, device=tok.device)
        b = bias if bias is not None else torch.zeros(cmid, dtype=torch.float32, device=tok.device)
        N.call("awseg_upconv3x3_linear", N.ptr(g9), B, cmid, h, w, height, width, N.ptr(b.contiguous()), N.ptr(z), 0, N.stream())
        ctx.save_for_backward(tok2, w1r)
        ctx.shape = (B, h, w, cin, cmid, height, width, bias is not None)
        return z

    @staticmethod
    def backward(ctx, dz):
        tok2, w1r = ctx.saved_tensors
        B, h, w, cin, cmid, height, width, has_bias = ctx.shape
        dzl = dz.permute(0, 2, 3, 1).contiguous()                                   # NHWC for the adjoint kernel (one transpose pass)
        dg9 = torch.empty(B, h, w, 9, cmid, dtype=torch.float32, device=dz.device)
        N.call("awseg_upconv3x3_adjoint", N.ptr(dzl), B, cmid, h, w, height, width, N.ptr(dg9), N.stream())
        dg2 = dg9.view(B * h * w, 9 * cmid)
        dtok = (dg2 @ w1r.t()).view(B, h, w, cin)
        dweight = (tok2.t() @ dg2).view(cin, 3, 3, cmid).permute(3, 0, 1, 2).contiguous()
        # bias gradient = sum of dz over the pixels = sum of the CENTRE tap's low-resolution gradient: that tap is never clipped and the
        # bilinear weights of a pixel sum to one, so the adjoint has already done the 17 GB reduction (a torch sum over dz: 4 + 2 ms a step)
        dbias = dg9[:, :, :, 4, :].sum(dim=(0, 1, 2)) if has_bias else None
        return dtok, dweight, dbias, None, None


def upconv3x3_train(tok: torch.Tensor, weight: torch.Tensor, bias: Optional[torch.Tensor], height: int, width: int) -> torch.Tensor:
    """Differentiable conv3x3(interpolate(f)) for NHWC encoder tokens [B,h,w,Cin] -> [B,Cmid,H,W]."""
    return _UpConv3x3.apply(tok.contiguous(), weight, bias, int(height), int(width))


def upconv3x3_train_supported(cin: int, cmid: int, h: int, w: int, height: int, width: int) -> bool:
    return cmid % 32 == 0 and cmid <= 256 and height >= 32 * h and width >= 32 * w


def aspp_depthwise3(x_nhwc: torch.Tensor, wdw: torch.Tensor, rates) -> torch.Tensor:
    """Depthwise halves of smp's three ASPPSeparableConv branches in one pass.
    x [B,h,w,C] NHWC, wdw [3,9,C] -> [3,B,h,w,C]."""
    x = x_nhwc.contiguous()
    b, h, w, c = x.shape
    out = torch.empty(3, b, h, w, c, dtype=torch.float32, device=x.device)
    N.call("awseg_aspp_depthwise3", N.ptr(x), b, h, w, c, N.ptr(wdw.contiguous()), int(rates[0]), int(rates[1]),
                                          int(rates[2]), N.ptr(out), N.stream())
    return out


def aspp_depthwise3_mean(x_nhwc: torch.Tensor, wdw: torch.Tensor, rates):
    """aspp_depthwise3 that also returns the per-image channel means [B,C] of x (ASPPPooling's global average) from the same pass;
    the mean is None where the kernel that can do it does not take the shape (the caller then reduces x itself)."""
    x = x_nhwc.contiguous()
    b, h, w, c = x.shape
    out = torch.empty(3, b, h, w, c, dtype=torch.float32, device=x.device)
    mean = torch.empty(b, c, dtype=torch.float32, device=x.device)
    rc = N.try_call("awseg_aspp_depthwise3_mean", N.ptr(x), b, h, w, c, N.ptr(wdw.contiguous()), int(rates[0]), int(rates[1]), int(rates[2]),
                    N.ptr(out), N.ptr(mean), N.stream())
    if rc != 0:
        return aspp_depthwise3(x, wdw, rates), None
    return out, mean


# ----------------------------------------------------------------------------- backbone helpers
def dwconv3x3_nhwc(x: torch.Tensor, w9: torch.Tensor, bias: Optional[torch.Tensor] = None, act: int = 0,
                   dilation: int = 1) -> torch.Tensor:
    """Depthwise 3x3 (stride 1, zero pad = dilation) on [B,H,W,C] float32 with fused bias + activation.
    w9 [9,C] (tap-major).  act: 0 none, 1 ReLU, 2 exact GELU."""
    x = x.contiguous()
    b, h, w, c = x.shape
    out = torch.empty_like(x)
    N.call("awseg_dwconv3x3_nhwc", N.ptr(x), b, h, w, c, int(dilation), N.ptr(w9.contiguous()),
           N.ptr(None if bias is None else bias.contiguous()), int(act), N.ptr(out), N.stream())
    return out


def dwconv3x3_wgrad_nhwc(x: torch.Tensor, dy: torch.Tensor, dilation: int = 1, want_bias: bool = True):
    """Weight (and bias) gradient of the depthwise 3x3 on NHWC tensors: (dW9 [9,C] tap-major, db [C] or None)."""
    x, dy = x.contiguous(), dy.contiguous()
    b, h, w, c = x.shape
    ws = N.workspace.get(x.device, N.lib().awseg_dwconv3x3_wgrad_workspace(b, h, w, c), tag="dwgrad")
    dw9 = torch.empty(9, c, dtype=torch.float32, device=x.device)
    db = torch.empty(c, dtype=torch.float32, device=x.device) if want_bias else None
    N.call("awseg_dwconv3x3_wgrad_nhwc", N.ptr(x), N.ptr(dy), b, h, w, c, int(dilation), N.ptr(ws), N.ptr(dw9), N.ptr(db), N.stream())
    return dw9, db


DW_TRAIN = os.environ.get("AWSEG_DW_TRAIN", "1") != "0"          # depthwise 3x3 convolutions of the TRAINING graph on this repo's kernels


class _DepthwiseConv3x3NHWC(torch.autograd.Function):
    """Depthwise 3x3 (stride 1, padding = dilation, groups = channels) on an NHWC tensor under autograd: forward and input gradient on
    awseg_dwconv3x3_nhwc (the gradient with flipped taps), weight / bias gradient on awseg_dwconv3x3_wgrad_nhwc.  MIOpen's
    immediate-mode picks for these layers cost 37 ms (weight gradient), 5 ms (input gradient) and 6 ms (forward) per call at
    1024x2048 — 14 + 14 + 8 calls a training step (profiles/r03_train_step_kernels.csv)."""

    @staticmethod
    def forward(ctx, x, weight, bias, dilation):
        c = weight.shape[0]
        w9 = weight.view(c, 9).t().contiguous()
        ctx.save_for_backward(x, weight)
        ctx.dilation, ctx.has_bias = int(dilation), bias is not None
        return dwconv3x3_nhwc(x, w9, bias, 0, dilation=dilation)

    @staticmethod
    def backward(ctx, g):
        x, weight = ctx.saved_tensors
        c = weight.shape[0]
        g = g.contiguous()
        dx = dw = db = None
        if ctx.needs_input_grad[0]:
            dx = dwconv3x3_nhwc(g, weight.view(c, 9).flip(1).t().contiguous(), None, 0, dilation=ctx.dilation)
        if ctx.needs_input_grad[1] or (ctx.has_bias and ctx.needs_input_grad[2]):
            dw9, db = dwconv3x3_wgrad_nhwc(x, g, ctx.dilation, want_bias=ctx.has_bias)
            dw = dw9.t().reshape(c, 1, 3, 3)
        return dx, dw, db, None


def depthwise_conv3x3_train_ok(conv, x: torch.Tensor) -> bool:
    """An nn.Conv2d this Function computes: depthwise 3x3, stride 1, padding == dilation, float32 on the GPU, C % 4 == 0."""
    return (DW_TRAIN and x.is_cuda and x.dtype == torch.float32 and conv.kernel_size == (3, 3) and conv.stride == (1, 1)
            and conv.groups == conv.in_channels == conv.out_channels and conv.padding == conv.dilation and conv.dilation[0] == conv.dilation[1]
            and conv.in_channels % 4 == 0 and conv.padding_mode == "zeros")


def depthwise_conv3x3_nhwc_train(x_nhwc: torch.Tensor, conv) -> torch.Tensor:
    return _DepthwiseConv3x3NHWC.apply(x_nhwc, conv.weight, conv.bias, conv.dilation[0])


BN_TRAIN = os.environ.get("AWSEG_BN_TRAIN", "1") != "0"          # BatchNorm2d -> ReLU -> Dropout2d of the training heads as fused HIP passes


class _BNReLUDropout2d(torch.autograd.Function):
    """BatchNorm2d (batch statistics) -> ReLU -> Dropout2d(p) on an NCHW map under autograd (csrc/bntrain.hip): one forward pass
    and two backward passes instead of ~15, and autograd keeps x + two per-channel vectors + the [B, C] noise instead of three
    full-size maps.  The Dropout2d mask is drawn exactly as F.dropout2d draws it (a [B, C, 1, 1] Bernoulli(1 - p) / (1 - p) from the
    device generator), so a seeded run consumes the same random numbers as the module graph."""

    @staticmethod
    def forward(ctx, x, gamma, beta, running_mean, running_var, momentum, eps, p, dx_channels_last=False):
        ctx.dx_cl = bool(dx_channels_last) and x.shape[1] % 32 == 0
        b, c, h, w = x.shape
        hw = h * w
        ws = N.workspace.get(x.device, N.lib().awseg_bn_train_workspace(b, c, hw), tag="bntrain")
        mean = torch.empty(c, dtype=torch.float32, device=x.device)
        var = torch.empty_like(mean)
        N.call("awseg_bn_train_stats", N.ptr(x), b, c, hw, N.ptr(ws), N.ptr(mean), N.ptr(var), N.stream())
        invstd = torch.rsqrt(var + eps)
        if running_mean is not None and momentum is not None:
            n = b * hw
            running_mean.mul_(1.0 - momentum).add_(mean, alpha=momentum)
            running_var.mul_(1.0 - momentum).add_(var, alpha=momentum * n / max(n - 1, 1))       # running statistics keep the UNBIASED variance
        noise = None
        if p > 0.0:
            noise = x.new_empty(b, c, 1, 1).bernoulli_(1.0 - p).div_(1.0 - p)                      # F.dropout2d's own draw
        out = torch.empty_like(x)
        N.call("awseg_bn_relu_dropout_forward", N.ptr(x), b, c, hw, N.ptr(mean), N.ptr(invstd), N.ptr(gamma.contiguous()), N.ptr(beta.contiguous()),
               N.ptr(None if noise is None else noise.view(b, c)), N.ptr(out), N.stream())
        ctx.save_for_backward(x, gamma, beta, mean, invstd, noise if noise is not None else mean.new_empty(0))
        ctx.has_noise = noise is not None
        return out

    @staticmethod
    def backward(ctx, g):
        x, gamma, beta, mean, invstd, noise = ctx.saved_tensors
        b, c, h, w = x.shape
        hw = h * w
        g = g.contiguous()
        ws = N.workspace.get(x.device, N.lib().awseg_bn_train_workspace(b, c, hw), tag="bntrain")
        dgamma, dbeta = torch.empty_like(mean), torch.empty_like(mean)
        # dx_cl: the gradient leaves as an NCHW tensor over channels-last MEMORY — the layout the producer's backward reads (ops._UpConv3x3)
        dx = torch.empty(b, h, w, c, dtype=x.dtype, device=x.device) if ctx.dx_cl else torch.empty_like(x)
        N.call("awseg_bn_relu_dropout_backward", N.ptr(x), N.ptr(g), b, c, hw, N.ptr(mean), N.ptr(invstd), N.ptr(gamma.contiguous()),
               N.ptr(beta.contiguous()), N.ptr(noise.view(b, c) if ctx.has_noise else None), N.ptr(ws), N.ptr(dgamma), N.ptr(dbeta), N.ptr(dx),
               int(ctx.dx_cl), N.stream())
        if ctx.dx_cl:
            dx = dx.permute(0, 3, 1, 2)
        return dx, dgamma, dbeta, None, None, None, None, None, None


def bn_relu_dropout2d_train_ok(x: torch.Tensor, bn, relu, drop) -> bool:
    """The module triple this Function computes: a training-mode affine BatchNorm2d with running statistics, nn.ReLU, and an
    nn.Dropout2d (or None) in training mode, on a contiguous float32 NCHW map on the GPU whose planes are whole float4s."""
    import torch.nn as nn
    return (BN_TRAIN and torch.is_grad_enabled() and x.is_cuda and x.dtype == torch.float32 and x.dim() == 4 and x.is_contiguous()
            and (x.shape[2] * x.shape[3]) % 4 == 0 and isinstance(bn, nn.BatchNorm2d) and bn.training and bn.affine
            and bn.track_running_stats and bn.momentum is not None and isinstance(relu, nn.ReLU)
            and (drop is None or (isinstance(drop, nn.Dropout2d) and drop.training)) and x.shape[0] <= 65535 and x.shape[1] <= 65535)


def bn_relu_dropout2d_train(x: torch.Tensor, bn, drop=None, dx_channels_last: bool = False) -> torch.Tensor:
    """dx_channels_last: hand the input gradient back over channels-last memory (for a producer whose backward reads NHWC)."""
    if bn.num_batches_tracked is not None:
        bn.num_batches_tracked.add_(1)
    return _BNReLUDropout2d.apply(x, bn.weight, bn.bias, bn.running_mean, bn.running_var, bn.momentum, bn.eps, 0.0 if drop is None else float(drop.p),
                                  dx_channels_last)


def bias_act_nhwc_(x_nhwc: torch.Tensor, bias: Optional[torch.Tensor], residual: Optional[torch.Tensor] = None,
                   act: int = 0) -> torch.Tensor:
    """In place: x = act(x + bias[c] (+ residual)) on a contiguous [..., C] float32 tensor."""
    c = x_nhwc.shape[-1]
    N.call("awseg_bias_act_nhwc", N.ptr(x_nhwc), x_nhwc.numel() // c, c, N.ptr(bias), N.ptr(residual), int(act), N.stream())
    return x_nhwc


def winograd_weights(weight: torch.Tensor, scale: Optional[torch.Tensor] = None) -> torch.Tensor:
    """[Cout,Cin,3,3] filters (times an optional per-Cout scale) -> U = G g G^T in the kernel's LDS
    image order [Cin/8][16 positions][2][Cout][4]: input channel 8*chunk + 4*(s>>1) + 2*hk + (s&1) at
    [chunk][p][hk][n][s] (include/awseg.h), computed in float64 and rounded once.  Cin % 16 == 0."""
    g = weight.double()
    if scale is not None:
        g = g * scale.double().view(-1, 1, 1, 1)
    cout, cin = weight.shape[0], weight.shape[1]
    G = torch.tensor([[1.0, 0.0, 0.0], [0.5, 0.5, 0.5], [0.5, -0.5, 0.5], [0.0, 0.0, 1.0]], dtype=torch.float64, device=weight.device)
    # channel within a chunk = 4*s_hi + 2*hk + s_lo with s = 2*s_hi + s_lo
    u = torch.einsum("ik,ockl,jl->ijco", G, g, G).reshape(16, cin // 8, 2, 2, 2, cout)   # [p][chunk][s_hi][hk][s_lo][n]
    return u.permute(1, 0, 3, 5, 2, 4).reshape(cin // 8, 16, 2, cout, 4).float().contiguous()   # [chunk][p][hk][n][s]


def conv3x3_winograd(x: torch.Tensor, u: torch.Tensor, shift: torch.Tensor, act: int = 0, dilation: int = 1,
                     residual: Optional[torch.Tensor] = None, w2: Optional[torch.Tensor] = None,
                     b2: Optional[torch.Tensor] = None) -> torch.Tensor:
    """x float32 [B,H,W,Cin] NHWC -> [B,H,W,Cout] (or [B,H,W] with the fused 1x1 + sigmoid head)."""
    x = x.contiguous()
    b, h, w, cin = x.shape
    cout = u.shape[3]
    out = torch.empty((b, h, w) if w2 is not None else (b, h, w, cout), dtype=torch.float32, device=x.device)
    N.call("awseg_conv3x3_winograd_nhwc", N.ptr(x), b, h, w, cin, cout, dilation, N.ptr(u), N.ptr(shift.contiguous()),
           N.ptr(None if residual is None else residual.contiguous()), act, N.ptr(None if w2 is None else w2.contiguous()),
           N.ptr(None if b2 is None else b2.contiguous()), N.ptr(out), N.stream())
    return out


# 3x3 convolutions: 1 = Winograd on split-operand f16 MFMA (csrc/wino_split.hip), 0 = Winograd on float32-input MFMA
WINO_SPLIT = os.environ.get("AWSEG_WINO_SPLIT", "1") != "0"

def winograd_split_weights(weight: torch.Tensor, scale: Optional[torch.Tensor] = None, bf16: bool = False) -> torch.Tensor:
    """[Cout,Cin,3,3] filters (times an optional per-Cout scale) -> the split-operand image of U = G g G^T that
    awseg_conv3x3_winograd_split_nhwc reads (include/awseg.h): f16 high and low parts of U * 2^-eu in the B-fragment
    order [Cin/16][16][Cout/32][hi k0-7 | hi k8-15 | lo k0-7 | lo k8-15][32][8], then 2^eu as one float32.  eu brings
    max|U| into [2^13, 2^14); it is computed on the device (no host synchronisation).  int16 tensor, 16-byte aligned."""
    g = weight.double()
    if scale is not None:
        g = g * scale.double().view(-1, 1, 1, 1)
    cout, cin = weight.shape[0], weight.shape[1]
    G = torch.tensor([[1.0, 0.0, 0.0], [0.5, 0.5, 0.5], [0.5, -0.5, 0.5], [0.0, 0.0, 1.0]], dtype=torch.float64, device=weight.device)
    u = torch.einsum("ik,ockl,jl->ijco", G, g, G).reshape(16, cin, cout)                   # [p][ci][co], float64
    mx = u.abs().max()
    e = torch.where(mx > 0, torch.floor(torch.log2(mx.clamp_min(1e-300))) - 13.0, torch.zeros_like(mx))
    us = u * torch.exp2(-e)
    if bf16:
        hi = us.to(torch.bfloat16).view(torch.int16)
        lo = torch.zeros_like(hi)
    else:
        hi = us.to(torch.float16)
        lo = (us - hi.double()).to(torch.float16).view(torch.int16)
        hi = hi.view(torch.int16)
    nch, ncb = cin // 16, cout // 32

    def frag(t):                                                                          # -> [nch][16][ncb][h][32][8]
        return t.view(16, nch, 2, 8, ncb, 32).permute(1, 0, 4, 2, 5, 3)
    img = torch.stack([frag(hi), frag(lo)], dim=3).reshape(-1).contiguous()               # [nch][16][ncb][hi/lo][h][32][8]
    n = img.numel()
    buf = torch.zeros(n + 8, dtype=torch.int16, device=weight.device)
    buf[:n] = img
    buf[n:n + 2] = torch.exp2(e).to(torch.float32).reshape(1).view(torch.int16)
    return buf


def conv3x3_winograd_split(x: torch.Tensor, u_split: torch.Tensor, cout: int, shift: torch.Tensor, act: int = 0, dilation: int = 1,
                           residual: Optional[torch.Tensor] = None, w2: Optional[torch.Tensor] = None,
                           b2: Optional[torch.Tensor] = None) -> torch.Tensor:
    """conv3x3_winograd on the split-operand f16-MFMA kernel: x float32 [B,H,W,Cin] NHWC, u_split = winograd_split_weights(...)."""
    x = x.contiguous()
    b, h, w, cin = x.shape
    need = N.lib().awseg_winograd_split_weight_halfs(cin, cout)
    if need < 0 or u_split.numel() != need + 8:
        raise N.AwsegError(f"u_split does not match Cin {cin}, Cout {cout} (expected {need} + 8 int16)")
    out = torch.empty((b, h, w) if w2 is not None else (b, h, w, cout), dtype=torch.float32, device=x.device)
    N.call("awseg_conv3x3_winograd_split_nhwc", N.ptr(x), b, h, w, cin, cout, dilation, N.ptr(u_split), N.ptr(shift.contiguous()),
           N.ptr(None if residual is None else residual.contiguous()), act, N.ptr(None if w2 is None else w2.contiguous()),
           N.ptr(None if b2 is None else b2.contiguous()), N.ptr(out), N.stream())
    return out


MIXFFN_FUSED = os.environ.get("AWSEG_MIXFFN_FUSED", "1") != "0"  # MiT Mix-FFN (LN -> fc1 -> dw3x3 + GELU -> fc2 + residual) as one tile kernel


def mixffn_split_weights(w: torch.Tensor) -> torch.Tensor:
    """float32 [N,K] -> float16 [2,N,K]: f16(w) and f16(w - f16(w)) — the operand images awseg_mixffn_fused multiplies."""
    hi = w.to(torch.float16)
    return torch.stack([hi, (w - hi.float()).to(torch.float16)]).contiguous()


def mixffn_fused(tok: torch.Tensor, gamma, beta, eps: float, w1, b1, dw_taps, dw_bias, w2, b2, w1_split=None, w2_split=None,
                 checked: bool = False) -> Optional[torch.Tensor]:
    """tok + fc2(gelu(dwconv3x3(fc1(layernorm(tok))))) on NHWC tokens [B,H,W,C] in one launch (csrc/mixffn.hip); None when the
    kernel does not take the problem (C not 32 / 64, hidden != 4C, weights or LayerNorm parameters beyond the f16 operand range):
    the caller keeps its separate launches.  `checked`: the caller has verified the range conditions (mixffn_operands_ok) —
    otherwise they are checked here, which synchronises the host with the device."""
    b, h, w, c = tok.shape
    if c not in (32, 64) or w1.shape != (4 * c, c) or w2.shape != (c, 4 * c) or dw_taps.shape != (9, 4 * c):
        return None
    if not checked and not mixffn_operands_ok(gamma, beta, w1, w2):
        return None
    tok = tok.contiguous()
    out = torch.empty_like(tok)
    w1s = w1_split if w1_split is not None else mixffn_split_weights(w1)
    w2s = w2_split if w2_split is not None else mixffn_split_weights(w2)
    rc = N.try_call("awseg_mixffn_fused", N.ptr(tok), b, h, w, c, N.ptr(gamma.contiguous()), N.ptr(beta.contiguous()), float(eps),
                    N.ptr(w1s), N.ptr(b1.contiguous()), N.ptr(dw_taps.contiguous()), N.ptr(dw_bias.contiguous()),
                    N.ptr(w2s), N.ptr(w2.contiguous()), N.ptr(b2.contiguous()), N.ptr(out), N.stream())
    return out if rc == 0 else None


def mixffn_operands_ok(gamma, beta, w1, w2) -> bool:
    """The f16 operand range of awseg_mixffn_fused: |w| < 2^15 and the LayerNorm outputs' bound max|gamma| sqrt(C) + max|beta| < 2^15
    (one host synchronisation: callers cache the answer per parameter version)."""
    c = gamma.numel()
    lim = 32768.0
    vals = torch.stack([gamma.abs().max() * (c ** 0.5) + beta.abs().max(), w1.abs().max(), w2.abs().max()])
    return bool(torch.isfinite(vals).all().item()) and bool((vals < lim).all().item())


TWO_STREAMS = os.environ.get("AWSEG_TWO_STREAMS", "1") != "0"    # ensemble eval: DeepLabV3+ on a side stream beside SegFormer (0: one stream)
DEPTH_FUSED = os.environ.get("AWSEG_DEPTH_FUSED", "1") != "0"    # the SegFormer depth head as one full-resolution launch (depthfuse.hip)


def upconv_forms(g9: torch.Tensor, shift: torch.Tensor) -> torch.Tensor:
    """The bilinear forms of relu(shift + conv3x3(interpolate_x32(f))) per frame (csrc/depthfuse.hip): float32 [B, F4 | F2] from
    g9 float32 [B,h,w,9,Cmid] (the per-tap products at the encoder's resolution, BatchNorm scale folded) and the folded shift."""
    g9 = g9.contiguous()
    b, h, w, nine, cmid = g9.shape
    assert nine == 9
    per = N.lib().awseg_upconv_forms_floats(h, w, cmid)
    forms = torch.empty(b, per, dtype=torch.float32, device=g9.device)
    N.call("awseg_upconv_forms", N.ptr(g9), b, cmid, h, w, N.ptr(shift.contiguous()), N.ptr(forms), N.stream())
    return forms


def depth_head_fused(forms: torch.Tensor, h: int, w: int, cmid: int, u_image: torch.Tensor, shift2: torch.Tensor, w2: torch.Tensor,
                     b2: torch.Tensor, bf16: bool = False) -> torch.Tensor:
    """sigmoid(conv1x1(relu(bn(conv3x3(relu(bn(conv3x3(interpolate(f)))))))) at [B, 32h, 32w] in ONE launch from the forms of
    upconv_forms: the hidden map between the two 3x3 convolutions is generated tile by tile inside the Winograd kernel.
    u_image = winograd_split_weights / winograd_bf16_weights of the SECOND 3x3 (Cmid -> 64)."""
    b = forms.shape[0]
    need = N.lib().awseg_winograd_split_weight_halfs(cmid, 64)
    if need < 0 or u_image.numel() != need + 8 or forms.shape[1] != N.lib().awseg_upconv_forms_floats(h, w, cmid):
        raise N.AwsegError(f"depth_head_fused: operands do not match h {h}, w {w}, Cmid {cmid}")
    out = torch.empty(b, 32 * h, 32 * w, dtype=torch.float32, device=forms.device)
    N.call("awseg_depth_head_fused", N.ptr(forms), b, h, w, cmid, N.ptr(u_image), int(bf16), N.ptr(shift2.contiguous()),
           N.ptr(w2.contiguous()), N.ptr(b2.contiguous()), N.ptr(out), N.stream())
    return out


GEMM_WORKSPACE_BYTES = 32 << 20
GEMM_TUNE = os.environ.get("AWSEG_GEMM_TUNE", "0") != "0"      # opt-in: time hipBLASLt's candidates once per new problem shape
# (measured on the bench step: 102.0 vs 101.8 images/s — the library's first-ranked algorithm is already the fastest here)
_gemm_tuned = set()


def gemm_bias_act(x: torch.Tensor, w: torch.Tensor, bias: torch.Tensor, act: int = 0, residual: Optional[torch.Tensor] = None,
                  out: Optional[torch.Tensor] = None, w_split: Optional[torch.Tensor] = None, split: Optional[bool] = None) -> torch.Tensor:
    """act(x[M,K] @ w[N,K]^T + bias (+ residual[M,N])) in one launch with the epilogue fused; `out` may be `residual`.
    Problems gemm_wants_split() accepts run on the split-operand f16-MFMA kernel (float32-grade, csrc/gemm_split.hip;
    pass `w_split = gemm_split_weights(w)` to reuse the split weights, else they are split on the fly); the rest is one
    hipBLASLt float32 call.  split=True/False overrides the choice.  With AWSEG_GEMM_TUNE=1 the first hipBLASLt call
    of a new (M,N,K,residual,act) times the library's candidate algorithms (synchronising, once)."""
    x, w = x.contiguous(), w.contiguous()
    m, k = x.shape
    n = w.shape[0]
    bias = bias.contiguous()
    if split is None and gemm_wants_bf16(m, n, k):
        wb = w_split if (w_split is not None and getattr(w_split, "_awseg_bf16", False)) else gemm_bf16_weights(w)
        return gemm_bf16_bias_act(x, wb, bias, act, residual=residual, out=out)
    if gemm_wants_split(m, n, k) if split is None else split:
        return gemm_split_bias_act(x, w_split if w_split is not None else gemm_split_weights(w), bias, act, residual=residual, out=out)
    ws = N.workspace.get(x.device, GEMM_WORKSPACE_BYTES, tag="gemm%d" % torch.cuda.current_stream(x.device).cuda_stream)   # one per stream (TWO_STREAMS)
    key = (str(x.device), m, n, k, residual is not None, act)
    if GEMM_TUNE and m > 0 and key not in _gemm_tuned:
        _gemm_tuned.add(key)
        scratch = torch.empty(m, n, dtype=torch.float32, device=x.device)
        rc = N.lib().awseg_gemm_tune(N.ptr(x), N.ptr(w), N.ptr(bias), int(residual is not None), act, N.ptr(scratch), m, n, k,
                                     N.ptr(ws), GEMM_WORKSPACE_BYTES, N.stream())
        if rc < 0:
            N.check(rc, "awseg_gemm_tune")
        del scratch
    if out is None:
        out = torch.empty(m, n, dtype=torch.float32, device=x.device)
    N.call("awseg_gemm_bias_act", N.ptr(x), N.ptr(w), N.ptr(bias), N.ptr(residual), act, N.ptr(out), m, n, k,
           N.ptr(ws), GEMM_WORKSPACE_BYTES, N.stream())
    return out


# 1x1 convolutions / Linear layers on the split-operand f16-MFMA GEMM (csrc/gemm_split.hip): faster than hipBLASLt's
# float32 kernels on every ResNet / ASPP / decoder shape measured (tools/kernel_bench.py "gemm"), including the HBM-bound
# ones; narrow outputs (N < 128 = one block tile) stay on the library.  AWSEG_GEMM_SPLIT=0 turns the split path off.
GEMM_SPLIT = os.environ.get("AWSEG_GEMM_SPLIT", "1") != "0"
GEMM_SPLIT_MIN_M, GEMM_SPLIT_MIN_N, GEMM_SPLIT_MIN_K = 128, 128, 64
GEMM_SPLIT_N64 = os.environ.get("AWSEG_GEMM_SPLIT_N64", "1") != "0"


GEMM_SPLIT_NARROW = os.environ.get("AWSEG_GEMM_SPLIT_NARROW", "1") != "0"
STRIDED_UPSAMPLE = os.environ.get("AWSEG_STRIDED_UPSAMPLE", "0") != "0"   # off: see upsample_bilinear
KV_PACKED = os.environ.get("AWSEG_KV_PACKED", "1") != "0"              # key + value projections as one GEMM, packed rows into attention
SMALL_CONV_SPLIT = os.environ.get("AWSEG_SMALL_CONV_SPLIT", "1") != "0"  # gathered-operand GEMM for the small patch convolutions too


def gemm_wants_split(m: int, n: int, k: int) -> bool:
    if not GEMM_SPLIT or k % 8:
        return False
    # N < 64 on 10^6 rows (DeepLabV3+'s 48-channel skip projection and 19-class head, MiT stage 1's 32-channel projections): one
    # masked 64-column tile of the LDS-DMA kernel, 128 rows a block — these launches are bound by reading x once
    if GEMM_SPLIT_NARROW and 8 <= n < 64 and k >= 32 and m >= (1 << 18):
        return True
    if GEMM_SPLIT_NARROW and n % 64 == 0 and k == 32 and m >= (1 << 18):      # MiT stage 1's fc1 (32 -> 128): a single K tile
        return True
    if m < GEMM_SPLIT_MIN_M or k < GEMM_SPLIT_MIN_K:
        return False
    # N = 64 (ResNet layer1 conv1, MiT stage-2 projections): only where the LDS-DMA kernel's 256 x 64 tiles fill the chip
    return n >= GEMM_SPLIT_MIN_N or (n == 64 and GEMM_SPLIT_N64 and ((m + 255) // 256) >= 128)


def gemm_split_weights(w: torch.Tensor) -> torch.Tensor:
    """w float32 [N,K] -> int16 [2,N,K]: f16 bit patterns of the high parts and of the scaled low parts of
    w * 2^-ew (ew normalises max|w| into [2^13, 2^14)).  The returned tensor is a view of a buffer that is 16 bytes
    longer: the trailer holds {max|w| bits, ew} and is read by awseg_gemm_split_bias_act — pass the view on as it is
    (a clone would drop the trailer)."""
    w = w.contiguous()
    n, k = w.shape
    # room for the classic [2][N][K] image, the 16-byte trailer and — K % 32 == 0 — the k-blocked image of gemm_split3.hip
    buf = torch.empty(int(N.lib().awseg_gemm_split_weight_halfs(n, k)), dtype=torch.int16, device=w.device)
    N.call("awseg_gemm_split_weights", N.ptr(w), n, k, N.ptr(buf), N.stream())
    return buf[:2 * n * k].view(2, n, k)


def _split_weights_intact(w_split: torch.Tensor, n: int, k: int) -> bool:
    """the view gemm_split_weights returned still sits in front of its trailer (+ k-blocked image): a clone would drop them"""
    return w_split.untyped_storage().nbytes() - w_split.storage_offset() * 2 >= int(N.lib().awseg_gemm_split_weight_halfs(n, k)) * 2


def gemm_split_bias_act(x: torch.Tensor, w_split: torch.Tensor, bias: Optional[torch.Tensor], act: int = 0,
                        residual: Optional[torch.Tensor] = None, out: Optional[torch.Tensor] = None) -> torch.Tensor:
    """act(x[M,K] @ w[N,K]^T + bias (+ residual[M,N])) with w given as gemm_split_weights(w); `out` may be `residual`."""
    x = x.contiguous()
    m, k = x.shape
    n = w_split.shape[1]
    if not _split_weights_intact(w_split, n, k):
        raise N.AwsegError("w_split lost its 16-byte trailer (weight exponent): pass the tensor gemm_split_weights returned, not a copy")
    if out is None:
        out = torch.empty(m, n, dtype=torch.float32, device=x.device)
    N.call("awseg_gemm_split_bias_act", N.ptr(x), N.ptr(w_split), N.ptr(bias), N.ptr(residual), act, N.ptr(out), m, n, k, N.stream())
    return out


def stem_rows_weights(w: torch.Tensor) -> torch.Tensor:
    """[N, C <= 4, kh, kw <= 8] convolution weights -> the [N, kh * 32] matrix of awseg_conv_rows_gemm_split_bias_act
    (column ky * 32 + kx * 4 + c, zeros elsewhere)."""
    n, c, kh, kw = w.shape
    if c > 4 or kw > 8:
        raise N.AwsegError("stem_rows_weights: at most 4 input channels and 8 kernel columns")
    m = torch.zeros(n, kh, 8, 4, dtype=torch.float32, device=w.device)
    m[:, :, :kw, :c] = w.permute(0, 2, 3, 1)
    return m.view(n, kh * 32)


def conv_rows_gemm_split(x_padded: torch.Tensor, w_split: torch.Tensor, bias: Optional[torch.Tensor], act: int, kernel_h: int,
                         stride: int, pad_y: int, out_width: int) -> Optional[torch.Tensor]:
    """The 7x7 stem as one split-operand GEMM (awseg_conv_rows_gemm_split_bias_act): x_padded float32 [B, H, W + pads, 4] with
    zero padding columns / channel, w_split = gemm_split_weights(stem_rows_weights(w)).  Returns [B, Ho, out_width, N], or None
    when the LDS-DMA kernel does not take the shape (the caller keeps its other path)."""
    b, h, wp, pf = x_padded.shape
    n, k = w_split.shape[1], w_split.shape[2]
    if k != kernel_h * 32 or not x_padded.is_contiguous():
        raise N.AwsegError("conv_rows_gemm_split: weights must be [N, kernel_h * 32] and the image contiguous")
    if not _split_weights_intact(w_split, n, k):
        raise N.AwsegError("w_split lost its 16-byte trailer (weight exponent): pass the tensor gemm_split_weights returned, not a copy")
    ho = (h + 2 * pad_y - kernel_h) // stride + 1
    out = torch.empty(b, ho, out_width, n, dtype=torch.float32, device=x_padded.device)
    rc = N.try_call("awseg_conv_rows_gemm_split_bias_act", N.ptr(x_padded), b, h, wp, pf, kernel_h, stride, pad_y, out_width,
                    N.ptr(w_split), N.ptr(bias), None, act, N.ptr(out), n, N.stream())
    return out if rc == 0 else None


# strided / patch convolutions: gather the A operand inside the split GEMM (default) or write the im2col matrix first (AWSEG_CONV_GATHER=0)
CONV_GATHER = os.environ.get("AWSEG_CONV_GATHER", "1") != "0"


ASPP_PIECES = os.environ.get("AWSEG_ASPP_PIECES", "1") != "0"          # ASPP projection: one GEMM over the four pixel branches
DUAL_TAIL = os.environ.get("AWSEG_DUAL_TAIL", "1") != "0"             # first bottleneck of a ResNet stage: conv3 + downsample branch as one GEMM


def gemm_split_dual(x: torch.Tensor, x2: torch.Tensor, w_split: torch.Tensor, bias: Optional[torch.Tensor], act: int = 0, stride: int = 0,
                    residual: Optional[torch.Tensor] = None) -> Optional[torch.Tensor]:
    """act([x | x2] @ w^T + bias (+ residual)) with the A operand in two pieces along K: x [M,k1] rows; x2 [M,k2] rows (stride 0) or an
    NHWC image [B,H,W,k2] whose pixels (b, oy*stride, ox*stride) are the rows (a strided 1x1 convolution folded into the product).
    w_split = gemm_split_weights(w [N,k1+k2]).  Returns None when the LDS-DMA kernel does not take the shape (the caller then runs
    the two products separately)."""
    x, x2 = x.contiguous(), x2.contiguous()
    m, k1 = x.shape
    k2 = x2.shape[-1]
    n = w_split.shape[1]
    if w_split.shape[2] != k1 + k2:
        raise N.AwsegError(f"gemm_split_dual: weights have K = {w_split.shape[2]}, the operands {k1} + {k2}")
    if not _split_weights_intact(w_split, n, k1 + k2):
        raise N.AwsegError("w_split lost its 16-byte trailer (weight exponent): pass the tensor gemm_split_weights returned, not a copy")
    if stride > 0:
        b, h, w = x2.shape[0], x2.shape[1], x2.shape[2]
    else:
        b, h, w = 1, 1, 1
        if x2.numel() != m * k2:
            raise N.AwsegError("gemm_split_dual: x2 has a different row count than x")
    out = torch.empty(m, n, dtype=torch.float32, device=x.device)
    rc = N.try_call("awseg_gemm_split_dual_bias_act", N.ptr(x), k1, N.ptr(x2), k2, b, h, w, int(stride), N.ptr(w_split), N.ptr(bias),
                    N.ptr(residual), act, N.ptr(out), m, n, N.stream())
    return out if rc == 0 else None


def gemm_split_pieces(pieces, w_split: torch.Tensor, bias: Optional[torch.Tensor], act: int = 0, residual: Optional[torch.Tensor] = None,
                      out: Optional[torch.Tensor] = None) -> Optional[torch.Tensor]:
    """act(cat(pieces, dim=1) @ w^T + bias (+ residual)) for 2 .. 4 equally wide row matrices [M,k] without the concatenation (one GEMM
    whose A operand is fetched piece by piece); `out` may be `residual`.  None when the LDS-DMA kernel does not take the shape."""
    import ctypes
    pieces = [p_.contiguous() for p_ in pieces]
    m, kp = pieces[0].shape
    n = w_split.shape[1]
    if any(p_.shape != (m, kp) for p_ in pieces) or w_split.shape[2] != kp * len(pieces):
        raise N.AwsegError("gemm_split_pieces: the pieces must be equally shaped and the weights [N, pieces * k]")
    if not _split_weights_intact(w_split, n, kp * len(pieces)):
        raise N.AwsegError("w_split lost its 16-byte trailer (weight exponent): pass the tensor gemm_split_weights returned, not a copy")
    if out is None:
        out = torch.empty(m, n, dtype=torch.float32, device=pieces[0].device)
    arr = (ctypes.c_void_p * len(pieces))(*[p_.data_ptr() for p_ in pieces])
    rc = N.try_call("awseg_gemm_split_pieces_bias_act", ctypes.cast(arr, ctypes.c_void_p), len(pieces), kp, N.ptr(w_split), N.ptr(bias),
                    N.ptr(residual), act, N.ptr(out), m, n, N.stream())
    return out if rc == 0 else None


def conv_gemm_split(x: torch.Tensor, w_split: torch.Tensor, bias: Optional[torch.Tensor], act: int, kh: int, kw: int, stride: int,
                    pad: int, dilation: int = 1, residual: Optional[torch.Tensor] = None) -> torch.Tensor:
    """conv2d on a float32 NHWC tensor [B,H,W,C] (C % 32 == 0) as one split-operand GEMM whose A operand is gathered from x
    while it is staged — the im2col matrix of `im2col_nhwc` is never written.  w_split = gemm_split_weights(w2) with
    w2 [N, kh*kw*C] in (ky, kx, c) column order.  Returns [B,Ho,Wo,N]; bit-identical to im2col_nhwc + gemm_split_bias_act."""
    x = x.contiguous()
    b, h, w, c = x.shape
    n, k = w_split.shape[1], w_split.shape[2]
    if k != kh * kw * c:
        raise N.AwsegError(f"conv_gemm_split: weights have K = {k}, the convolution needs {kh * kw * c}")
    if not _split_weights_intact(w_split, n, k):
        raise N.AwsegError("w_split lost its 16-byte trailer (weight exponent): pass the tensor gemm_split_weights returned, not a copy")
    ho = (h + 2 * pad - dilation * (kh - 1) - 1) // stride + 1
    wo = (w + 2 * pad - dilation * (kw - 1) - 1) // stride + 1
    out = torch.empty(b, ho, wo, n, dtype=torch.float32, device=x.device)
    N.call("awseg_conv_gemm_split_bias_act", N.ptr(x), b, h, w, c, kh, kw, stride, pad, dilation, N.ptr(w_split), N.ptr(bias),
           N.ptr(residual), act, N.ptr(out), n, N.stream())
    return out


def im2col_nhwc(x: torch.Tensor, kh: int, kw: int, stride: int, pad: int, dilation: int = 1, k_padded: Optional[int] = None):
    """x float32 [B,H,W,C] -> (cols [B*Ho*Wo, k_padded], Ho, Wo): column (ky*kw + kx)*C + c, zero padded (awseg.h)."""
    x = x.contiguous()
    b, h, w, c = x.shape
    k = kh * kw * c
    kp = k if k_padded is None else int(k_padded)
    ho = (h + 2 * pad - dilation * (kh - 1) - 1) // stride + 1
    wo = (w + 2 * pad - dilation * (kw - 1) - 1) // stride + 1
    cols = torch.empty(b * ho * wo, kp, dtype=torch.float32, device=x.device)
    N.call("awseg_im2col_nhwc", N.ptr(x), b, h, w, c, kh, kw, stride, pad, dilation, kp, N.ptr(cols), N.stream())
    return cols, ho, wo


def dwconv3x3_upcat(a: torch.Tensor, hi: torch.Tensor, w9: torch.Tensor) -> torch.Tensor:
    """depthwise3x3(cat(bilinear_up_align_corners(a -> hi's size), hi)) on NHWC tensors: a [B,h,w,Ca], hi [B,H,W,Ch],
    w9 [9,Ca+Ch] -> [B,H,W,Ca+Ch] (DeepLabV3+ decoder, no intermediate tensors)."""
    a, hi = a.contiguous(), hi.contiguous()
    b, h, w, ca = a.shape
    _, H, W, ch = hi.shape
    out = torch.empty(b, H, W, ca + ch, dtype=torch.float32, device=a.device)
    N.call("awseg_dwconv3x3_upcat_nhwc", N.ptr(a), h, w, ca, N.ptr(hi), ch, b, H, W, N.ptr(w9.contiguous()), N.ptr(out), N.stream())
    return out


# default attention kernel: 1 = split-operand f16 MFMA (22-bit operands, float32 accumulation: float32-grade results at
# twice the speed, csrc/attn.hip), 0 = float32-input MFMA
ATTENTION_SPLIT = os.environ.get("AWSEG_ATTN_SPLIT", "1") != "0"


def attention_d32(q: torch.Tensor, k: torch.Tensor, v: torch.Tensor, heads: int, scale: float,
                  split: Optional[bool] = None) -> torch.Tensor:
    """softmax(q k^T * scale) v for head_dim 32: q [B,Nq,heads*32], k / v [B,Nkv,heads*32] (token-major) -> [B,Nq,heads*32].
    split=True runs the split-operand f16-MFMA kernel (22-bit operands, float32 accumulation), False the float32-MFMA
    kernel; None takes the process default (ATTENTION_SPLIT, env AWSEG_ATTN_SPLIT=0/1)."""
    q, k, v = q.contiguous(), k.contiguous(), v.contiguous()
    b, nq, c = q.shape
    out = torch.empty_like(q)
    if split is None and PRECISION == "bf16":
        sym = "awseg_attention_d32_bf16"
    else:
        sym = "awseg_attention_d32_split" if (ATTENTION_SPLIT if split is None else split) else "awseg_attention_d32"
    if sym == "awseg_attention_d32_split" and _wants_kv_image(nq, k.shape[1]):
        _attention_split_ws(q, k, v, 0, out, b, heads, nq, k.shape[1], scale)
        return out
    N.call(sym, N.ptr(q), N.ptr(k), N.ptr(v), N.ptr(out), b, heads, nq, k.shape[1], float(scale), N.stream())
    return out


ATTN_KV_IMAGE = os.environ.get("AWSEG_ATTN_KV_IMAGE", "1") != "0"     # split-operand attention: keys / values prepared once per launch


def _wants_kv_image(nq: int, nkv: int) -> bool:
    """The prepared image pays where many query blocks share the keys: MiT stages 1 / 2 at the bench shape (64 / 16 queries per key:
    1.148 -> 1.053 ms for stage 1); at 4 queries per key (stage 3: 0.372 -> 0.376 ms) the preparing kernel costs what it saves."""
    return ATTN_KV_IMAGE and nq >= 8 * nkv


def _attention_split_ws(q, k, v, pitch, out, b, heads, nq, nkv, scale):
    """awseg_attention_d32_split_ws on a per-stream scratch workspace (v may be a view into k's rows: packed keys | values)."""
    nbytes = int(N.lib().awseg_attention_d32_split_workspace(b, heads, nkv))
    ws = N.workspace.get(q.device, nbytes, tag="attn%d" % torch.cuda.current_stream(q.device).cuda_stream)
    N.call("awseg_attention_d32_split_ws", N.ptr(q), N.ptr_strided(k), N.ptr_strided(v), int(pitch), N.ptr(out), b, heads, nq, nkv, float(scale),
           N.ptr(ws), N.stream())


def attention_d32_packed_kv(q: torch.Tensor, kv: torch.Tensor, heads: int, scale: float, split: Optional[bool] = None) -> torch.Tensor:
    """attention_d32 with keys and values packed per token: kv [B,Nkv,2*heads*32] = [key | value], as one GEMM over the stacked key /
    value projection weights writes them.  Same kernels, same values."""
    q, kv = q.contiguous(), kv.contiguous()
    b, nq, c = q.shape
    assert kv.shape[-1] == 2 * c
    out = torch.empty_like(q)
    mode = 2 if (split is None and PRECISION == "bf16") else (1 if (ATTENTION_SPLIT if split is None else split) else 0)
    if mode == 1 and _wants_kv_image(nq, kv.shape[1]):
        _attention_split_ws(q, kv, kv[..., c:], 2 * c, out, b, heads, nq, kv.shape[1], scale)
        return out
    N.call("awseg_attention_d32_packed_kv", N.ptr(q), N.ptr(kv), N.ptr(out), b, heads, nq, kv.shape[1], float(scale), mode, N.stream())
    return out


# Compute precision of the dense contractions (1x1 convolutions / Linear layers, 3x3 Winograd convolutions, attention):
# "f32" = float32-grade results (split-operand f16 MFMA or float32-input MFMA, see set_split), "bf16" = one bf16 MFMA per
# product tile with float32 accumulation (BASELINE config 5).  Models switch it for the duration of their forward.
PRECISION = "f32"


class precision:
    """Context manager: `with ops.precision("bf16"): model(x)`."""

    def __init__(self, dtype: Optional[str]):
        self.dtype = {None: None, "f32": "f32", "fp32": "f32", "float32": "f32", "bf16": "bf16", "bfloat16": "bf16"}[dtype]

    def __enter__(self):
        global PRECISION
        self.prev = PRECISION
        if self.dtype is not None:
            PRECISION = self.dtype
        return self

    def __exit__(self, *exc):
        global PRECISION
        PRECISION = self.prev
        return False


def gemm_wants_bf16(m: int, n: int, k: int) -> bool:
    return PRECISION == "bf16" and m >= 128 and n >= 32 and k >= 16 and k % 8 == 0


def gemm_bf16_weights(w: torch.Tensor) -> torch.Tensor:
    """w float32 [N,K] -> int16 view [2,N,K] of a buffer with a 16-byte trailer (plane 0 = bf16(w), plane 1 unused)."""
    w = w.contiguous()
    n, k = w.shape
    buf = torch.zeros(int(N.lib().awseg_gemm_bf16_weight_halfs(n, k)), dtype=torch.int16, device=w.device)
    N.call("awseg_gemm_bf16_weights", N.ptr(w), n, k, N.ptr(buf), N.stream())
    return buf[:2 * n * k].view(2, n, k)


def gemm_bf16_bias_act(x: torch.Tensor, w_bf16: torch.Tensor, bias: Optional[torch.Tensor], act: int = 0,
                       residual: Optional[torch.Tensor] = None, out: Optional[torch.Tensor] = None) -> torch.Tensor:
    x = x.contiguous()
    m, k = x.shape
    n = w_bf16.shape[1]
    if out is None:
        out = torch.empty(m, n, dtype=torch.float32, device=x.device)
    N.call("awseg_gemm_bf16_bias_act", N.ptr(x), N.ptr(w_bf16), N.ptr(bias), N.ptr(residual), act, N.ptr(out), m, n, k, N.stream())
    return out


def winograd_bf16_weights(weight: torch.Tensor, scale: Optional[torch.Tensor] = None) -> torch.Tensor:
    """The image winograd_split_weights builds, with bf16(U * 2^-eu) in the high slots and zeros in the (unread) low slots."""
    return winograd_split_weights(weight, scale, bf16=True)


def conv3x3_winograd_bf16(x: torch.Tensor, u_bf16: torch.Tensor, cout: int, shift: torch.Tensor, act: int = 0, dilation: int = 1,
                          residual: Optional[torch.Tensor] = None, w2: Optional[torch.Tensor] = None,
                          b2: Optional[torch.Tensor] = None) -> torch.Tensor:
    x = x.contiguous()
    b, h, w, cin = x.shape
    out = torch.empty((b, h, w) if w2 is not None else (b, h, w, cout), dtype=torch.float32, device=x.device)
    N.call("awseg_conv3x3_winograd_bf16_nhwc", N.ptr(x), b, h, w, cin, cout, dilation, N.ptr(u_bf16), N.ptr(shift.contiguous()),
           N.ptr(None if residual is None else residual.contiguous()), act, N.ptr(None if w2 is None else w2.contiguous()),
           N.ptr(None if b2 is None else b2.contiguous()), N.ptr(out), N.stream())
    return out


def set_split(on: bool) -> None:
    """Process-wide switch between the split-operand f16-MFMA kernels (float32-grade results) and
    their float32-input MFMA counterparts, for every operator that has both."""
    global ATTENTION_SPLIT, GEMM_SPLIT, WINO_SPLIT, HEAD_SPLIT
    ATTENTION_SPLIT = GEMM_SPLIT = WINO_SPLIT = HEAD_SPLIT = bool(on)


def split_state() -> dict:
    return {"attention": ATTENTION_SPLIT, "gemm_1x1": GEMM_SPLIT, "winograd_3x3": WINO_SPLIT, "segformer_head": HEAD_SPLIT}


def restore_split(state: dict) -> None:
    global ATTENTION_SPLIT, GEMM_SPLIT, WINO_SPLIT, HEAD_SPLIT
    ATTENTION_SPLIT, GEMM_SPLIT, WINO_SPLIT = bool(state["attention"]), bool(state["gemm_1x1"]), bool(state["winograd_3x3"])
    HEAD_SPLIT = bool(state.get("segformer_head", HEAD_SPLIT))


def maxpool3x3s2_nhwc(x_nhwc: torch.Tensor, shift: Optional[torch.Tensor] = None) -> torch.Tensor:
    """nn.MaxPool2d(3, stride=2, padding=1) on a contiguous float32 [B,H,W,C] tensor -> [B,Ho,Wo,C] (no index tensor); with
    `shift` [C]: relu(pool + shift) in the same pass (the ResNet stem's BatchNorm shift + ReLU moved behind the pooling)."""
    x = x_nhwc.contiguous()
    b, h, w, c = x.shape
    out = torch.empty(b, (h - 1) // 2 + 1, (w - 1) // 2 + 1, c, dtype=torch.float32, device=x.device)
    if shift is not None:
        N.call("awseg_maxpool3x3s2_bias_relu_nhwc", N.ptr(x), b, h, w, c, N.ptr(shift.contiguous()), N.ptr(out), N.stream())
    else:
        N.call("awseg_maxpool3x3s2_nhwc", N.ptr(x), b, h, w, c, N.ptr(out), N.stream())
    return out


def upsample_bilinear(x: torch.Tensor, size, align_corners: bool) -> torch.Tensor:
    """F.interpolate(x, size, mode="bilinear", align_corners=...) / nn.UpsamplingBilinear2d on an NCHW float32 tensor, torch's
    arithmetic (bit-identical), 4 output pixels per lane."""
    b, c, h, w = x.shape
    H, W = int(size[0]), int(size[1])
    # (the 4 x 4-pixels-per-lane kernel can read the small map through its strides — awseg_upsample_bilinear_strided — but lanes that
    # walk NHWC rows 76 bytes apart keep the texture addresser busy for 0.73 ms where the planar map takes 0.32: a planar copy of
    # the small map (57 us at the bench shape) is the cheaper way, measured)
    if not (x.is_contiguous() or (STRIDED_UPSAMPLE and W % 4 == 0 and 3 * w < W and 3 * h < H and all(s_ >= 0 for s_ in x.stride()))):
        x = x.contiguous()
    out = torch.empty(b, c, H, W, dtype=torch.float32, device=x.device)
    sb, sc, sy, sx = x.stride()
    N.call("awseg_upsample_bilinear_strided", N.ptr_strided(x), b, c, h, w, sb, sc, sy, sx, H, W, int(bool(align_corners)), N.ptr(out), N.stream())
    return out


def stem_image_fill(x: torch.Tensor, image: torch.Tensor) -> None:
    """x [B,C<=4,H,W] (innermost stride 1) -> columns 3.. of the zero-padded NHWC image [B,H,Wp,4] of the 7x7 stems, one pass."""
    b, c, h, w = x.shape
    if x.stride(3) != 1:
        x = x.contiguous()
    N.call("awseg_stem_image", N.ptr_strided(x), b, c, h, w, x.stride(0), x.stride(1), x.stride(2), N.ptr(image), image.shape[2], N.stream())


def rowdot_sigmoid(x: torch.Tensor, w: torch.Tensor, bias: Optional[torch.Tensor], sigmoid: bool = True) -> torch.Tensor:
    """sigmoid(x @ w + bias) for x [rows,k], w [k] -> [rows]: a 1x1 convolution to one channel (+ Sigmoid) on NHWC rows."""
    x, w = x.contiguous(), w.contiguous().view(-1)
    out = torch.empty(x.shape[0], dtype=torch.float32, device=x.device)
    N.call("awseg_rowdot_sigmoid", N.ptr(x), x.shape[0], x.shape[1], N.ptr(w), N.ptr(bias), int(bool(sigmoid)), N.ptr(out), N.stream())
    return out


def aspp_pool_branch(mean: torch.Tensor, w1: torch.Tensor, b1: torch.Tensor, w2: torch.Tensor, b2: Optional[torch.Tensor]) -> torch.Tensor:
    """relu(mean @ w1^T + b1) @ w2^T + b2 for one row per image (the ASPP pooling branch and its slice of the projection)."""
    mean, w1, b1, w2 = mean.contiguous(), w1.contiguous(), b1.contiguous(), w2.contiguous()
    b, cin = mean.shape
    cmid, cout = w1.shape[0], w2.shape[0]
    ws = torch.empty(int(N.lib().awseg_aspp_pool_branch_workspace(b, cmid)), dtype=torch.uint8, device=mean.device)
    out = torch.empty(b, cout, dtype=torch.float32, device=mean.device)
    N.call("awseg_aspp_pool_branch", N.ptr(mean), b, cin, N.ptr(w1), N.ptr(b1), cmid, N.ptr(w2), N.ptr(None if b2 is None else b2.contiguous()),
           cout, N.ptr(ws), N.ptr(out), N.stream())
    return out


def depth_upsample_combine(d1: torch.Tensor, d2_low: torch.Tensor, weights: Optional[torch.Tensor]):
    """(d2_full, d) with d2_full = bilinear(align_corners=False) upsample of d2_low [B,1,h,w] to d1's size [B,1,H,W] and
    d = weights[0]*d1 + weights[1]*d2_full (mean when weights is None) — PKG/models/model.py:368-371, 471-478."""
    d1, d2_low = d1.contiguous(), d2_low.contiguous()
    b, _, H, W = d1.shape
    h, w = d2_low.shape[-2:]
    d2_full, d = torch.empty_like(d1), torch.empty_like(d1)
    N.call("awseg_depth_upsample_combine", N.ptr(d1), N.ptr(d2_low), b, h, w, H, W, N.ptr(None if weights is None else weights.contiguous()),
           N.ptr(d2_full), N.ptr(d), N.stream())
    return d2_full, d


def layernorm_rows(x: torch.Tensor, gamma: torch.Tensor, beta: torch.Tensor, eps: float) -> torch.Tensor:
    """torch.nn.functional.layer_norm over the last dimension for small channel counts (MiT tokens)."""
    x = x.contiguous()
    c = x.shape[-1]
    out = torch.empty_like(x)
    N.call("awseg_layernorm_rows", N.ptr(x), x.numel() // c, c, N.ptr(gamma.contiguous()), N.ptr(beta.contiguous()), float(eps),
           N.ptr(out), N.stream())
    return out
