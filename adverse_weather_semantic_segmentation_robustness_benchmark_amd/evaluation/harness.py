"""`evaluate_model` — the evaluation hot loop of REF/scripts/evaluate.py:134-274, same arguments
and result keys, with every per-pixel quantity kept on the GPU.

Reference: forward -> argmax -> `.cpu()` of predictions, labels, the FULL logits and both member
logits per batch, `torch.cat` on the host (9.5 GB for 20 frames at 1024x2048), then metrics
once at the end.  Here each batch updates additive device counters and nothing else survives it:

  * confusion  int64[1+K, C*C]   slot 0 overall, slot 1+k weather condition k  (HIP, fused with
                                  combine / temperature / argmax)
  * ECE bins   [1+K, 15] x {count, sum conf, sum correct}                      (HIP)
  * AUROC      int64[2, 2^13] histogram of the disagreement score by error/non-error

Counters are SUM-all-reduced over ranks once (RCCL), then rank-agnostic host math finishes:
the 19-element IoU divide/mean uses the reference's own torch expressions, so identical counts
give bit-identical mIoU at any GPU count.
"""
from __future__ import annotations

import logging
from typing import Any, Dict

import numpy as np
import os

import torch
import torch.nn.functional as F

from .. import _native as N
from .. import ops, parallel
from .metrics import ConfidenceCalibration, RobustnessMetrics

logger = logging.getLogger(__name__)

AUROC_BINS = 1 << 13                      # per-block LDS histogram (2 x 8192 x 4 B = 64 KB)
AUROC_LO, AUROC_HI = -1e-3, 0.70          # mutual information of two members lies in [0, ln 2]
# The device AUROC is the rank statistic of a 2^13-bin histogram of the disagreement score (ties inside a bin count half), not
# sklearn's exact ranks (metrics.py:434): asserted within AUROC_TOLERANCE of sklearn end to end (tests/test_gpu_models.py) —
# 30x looser than every other gate of the path, so the results dict says how the number was made.
AUROC_TOLERANCE = 3e-3


def _cfg(config, key, default):
    try:
        v = config.get(key, default)
    except Exception:  # noqa: BLE001
        v = default
    return default if v is None else v


class EvalState:
    """All cross-batch state of one evaluation run (device resident, additive)."""

    def __init__(self, metrics: RobustnessMetrics, conditions, device, num_bins: int = 15, ensemble: bool = False):
        self.acc = metrics.new_accumulator(device)
        self.acc.conditions = list(conditions)
        self.acc.counts = ops.new_counts(metrics.num_classes, device, 1 + len(conditions))
        self.ece = ops.new_ece_bins(num_bins, device, 1 + len(conditions))
        self.edges = torch.linspace(0, 1, num_bins + 1).to(device)
        self.auroc = torch.zeros(2, AUROC_BINS, dtype=torch.int64, device=device) if ensemble else None
        self.samples = 0

    def update_auroc(self, seg1, seg2, labels):
        """Disagreement = mutual information (metrics.py:353-367); error = argmax of the MEAN
        PROBABILITY != label (metrics.py:414-419); pixels with label 255 dropped (:426)."""
        p1, p2 = F.softmax(seg1, dim=1), F.softmax(seg2, dim=1)
        m = (p1 + p2) / 2
        h_mean = -(m * torch.log(m + 1e-8)).sum(dim=1)
        h_ind = (-(p1 * torch.log(p1 + 1e-8)).sum(dim=1) - (p2 * torch.log(p2 + 1e-8)).sum(dim=1)) / 2
        dis = (h_mean - h_ind).reshape(-1)
        lab = labels.reshape(-1).long()
        err = (m.argmax(dim=1).reshape(-1) != lab)
        valid = lab != 255
        b = ((dis - AUROC_LO) * (AUROC_BINS / (AUROC_HI - AUROC_LO))).long().clamp_(0, AUROC_BINS - 1)
        idx = b + err.long() * AUROC_BINS
        # no boolean indexing / bincount here: both would synchronise the host with the device every batch
        self.auroc.view(-1).index_add_(0, idx, valid.to(torch.int64))

    def all_reduce(self):
        """ONE SUM all-reduce of every counter — all int64 (confusion, ECE bins with fixed-point confidence sums, AUROC
        histogram): integer sums are order-independent, so the results are bit-identical at any rank count."""
        ts = [self.acc.counts, self.acc.oob, self.ece]
        if self.auroc is not None:
            ts.append(self.auroc)
        parallel.all_reduce_sum_(ts)

    def auroc_value(self) -> float:
        neg, pos = self.auroc[0].double(), self.auroc[1].double()
        P, Nn = pos.sum().item(), neg.sum().item()
        if P == 0 or Nn == 0:
            return 0.5                                                             # metrics.py:430-431
        below = torch.cumsum(neg, 0) - neg
        return float((pos * (below + 0.5 * neg)).sum().item() / (P * Nn))


# confusion + calibration + disagreement statistics in one pass over the member logits (AWSEG_STATS_ONE_PASS=0: two passes)
STATS_ONE_PASS = os.environ.get("AWSEG_STATS_ONE_PASS", "1") != "0"


@torch.no_grad()
def eval_batch(model, st: EvalState, images: torch.Tensor, labels: torch.Tensor, conds, metrics: RobustnessMetrics,
               with_stats: bool = True) -> None:
    """One batch of the evaluation loop (evaluate.py:166-200 + the per-batch share of :203-255): forward, argmax,
    confusion per condition, and (with_stats) the ECE bins and the disagreement histogram — all into `st`'s
    device counters.  Nothing per-pixel survives the call."""
    cond = st.acc.cond_ids(conds)
    if labels.dtype not in (torch.uint8, torch.int64):
        labels = labels.long()
    if st.auroc is not None:
        strategy = getattr(model, "ensemble_strategy", "weighted_average")
        fused_stats = metrics.num_classes == 19 and strategy != "max_confidence" and images[0, 0].numel() % 4 == 0
        need_logits = with_stats and not fused_stats
        one_pass = (st.edges, st.ece, st.auroc, AUROC_LO, AUROC_HI) if (with_stats and fused_stats and STATS_ONE_PASS) else None
        res = model.forward_eval(images, labels, st.acc.counts, st.acc.oob, cond, want_logits=need_logits, want_pred=False, stats=one_pass)
        if one_pass is not None and getattr(model, "_stats_fused", False):
            pass                                                  # confusion, ECE bins and the disagreement histogram came out of ONE pass
        elif with_stats and fused_stats:
            # ECE of the combined logits + disagreement histogram in ONE pass over the member logits:
            # the ensemble logits are never materialised
            mode = N.COMBINE_WEIGHTED if strategy == "weighted_average" else N.COMBINE_MEAN
            w = F.softmax(model.ensemble_weights, dim=0) if mode == N.COMBINE_WEIGHTED else None
            T = model.temperature if getattr(model, "temperature_scaling", False) else None
            ops.ensemble_eval_stats(res["segformer_seg"], res["deeplabv3plus_seg"], mode, w, T, labels, cond, st.edges, st.ece,
                                    st.auroc, AUROC_LO, AUROC_HI)
        elif with_stats:
            st.update_auroc(res["segformer_seg"], res["deeplabv3plus_seg"], labels)
            ops.ece_accumulate(res["segmentation"], labels, st.ece, st.edges, cond)
    else:
        logits = model(images)["segmentation"].float().contiguous()
        ops.combine_argmax_confusion(logits, None, 3, want_logits=False, label=labels.contiguous(), counts=st.acc.counts,
                                     oob=st.acc.oob, cond=cond)
        if with_stats:
            ops.ece_accumulate(logits, labels, st.ece, st.edges, cond)
    st.samples += images.size(0)


@torch.no_grad()
def evaluate_model(model: torch.nn.Module, test_loader, metrics: RobustnessMetrics, device, config) -> Dict[str, Any]:
    model.eval()
    conditions = list(_cfg(config, "data.weather_conditions", []))
    num_bins = int(_cfg(config, "evaluation.num_bins", 15))
    is_ensemble = hasattr(model, "segformer") and hasattr(model, "deeplabv3plus")
    st = EvalState(metrics, conditions, device, num_bins, ensemble=is_ensemble)
    for batch in test_loader:
        images = batch["image"].to(device)
        labels = batch["label"].to(device)
        eval_batch(model, st, images, labels, batch.get("weather_condition", ["clean"] * images.size(0)), metrics)
    return finalize(st, metrics)


def finalize(st: EvalState, metrics: RobustnessMetrics) -> Dict[str, Any]:
    """All-reduce the counters, then the scalar host math of evaluate.py:214-271."""
    st.all_reduce()
    st.acc.check()
    results: Dict[str, Any] = {"overall_miou": st.acc.miou(0)}
    weather_mious = {}
    for k, name in enumerate(st.acc.conditions):
        if st.acc.present(1 + k):
            weather_mious[name] = st.acc.miou(1 + k)
            results[f"miou_{name}"] = weather_mious[name]
    bins = ops.ece_bins_to_numpy(st.ece)
    results["expected_calibration_error"] = ConfidenceCalibration.ece_from_bins(bins[0])
    for k, name in enumerate(st.acc.conditions):
        if bins[1 + k]["count"].sum() > 0:
            results[f"ece_{name}"] = ConfidenceCalibration.ece_from_bins(bins[1 + k])
    if st.auroc is not None:
        results["ensemble_disagreement_auroc"] = st.auroc_value()
        results["ensemble_disagreement_auroc_bins"] = float(AUROC_BINS)             # rank histogram, not exact ranks:
        results["ensemble_disagreement_auroc_tolerance"] = AUROC_TOLERANCE           # |device - sklearn| bound the tests assert
    if "clean" in weather_mious:
        for w in ("fog", "rain", "snow", "night"):
            if w in weather_mious:
                results[f"robustness_degradation_{w}"] = metrics.compute_robustness_degradation_ratio(
                    weather_mious["clean"], weather_mious[w])
        degs = [results[f"robustness_degradation_{w}"] for w in ("fog", "rain", "snow", "night")
                if f"robustness_degradation_{w}" in results]
        if degs:
            results["robustness_degradation_ratio"] = np.mean(degs)
    return results
