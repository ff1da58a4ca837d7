"""Evaluation report files of REF/scripts/evaluate.py:277-392: the raw
`evaluation_results.json` and the markdown `evaluation_report.md` with the same sections, row
formats (three decimals) and pass marks, so downstream tooling that parses the reference's report
reads this one."""
from __future__ import annotations

import json
from pathlib import Path
from typing import Any, Dict, Optional

# target values the reference compares against when none are given (evaluate.py:303-311)
DEFAULT_TARGETS = {"miou_clean": 0.78, "miou_fog": 0.65, "miou_rain": 0.62, "robustness_degradation_ratio": 0.18,
                   "expected_calibration_error": 0.05, "ensemble_disagreement_auroc": 0.85}
ADVERSE = ("fog", "rain", "snow", "night")


def report_markdown(results: Dict[str, Any], target_metrics: Optional[Dict[str, float]] = None) -> str:
    targets = DEFAULT_TARGETS if target_metrics is None else target_metrics
    lines = ["# Adverse Weather Semantic Segmentation Evaluation Report", "", "## Summary Metrics", "",
             "| Metric | Target | Actual | Status |", "|--------|--------|--------|--------|"]
    for name, want in targets.items():
        got = results.get(name, 0.0)
        lines.append(f"| {name} | {want:.3f} | {got:.3f} | {'✓' if got >= want else '✗'} |")    # `>=` for every row, :315
    lines += ["", "## Weather-Specific Performance", ""]
    lines += [f"- **{c.title()}**: mIoU = {results[f'miou_{c}']:.3f}" for c in ("clean",) + ADVERSE if f"miou_{c}" in results]
    lines += ["", "## Robustness Analysis", ""]
    if "robustness_degradation_ratio" in results:
        lines.append(f"- **Overall Degradation Ratio**: {results['robustness_degradation_ratio']:.3f}")
    lines += [f"- **{c.title()} Degradation**: {results[f'robustness_degradation_{c}']:.3f}" for c in ADVERSE
              if f"robustness_degradation_{c}" in results]
    if "expected_calibration_error" in results:
        lines += ["", "## Confidence Calibration", "", f"- **Expected Calibration Error**: {results['expected_calibration_error']:.3f}"]
    if "ensemble_disagreement_auroc" in results:
        lines += ["", "## Ensemble Performance", "", f"- **Disagreement AUROC**: {results['ensemble_disagreement_auroc']:.3f}"]
    return "\n".join(lines)


def generate_evaluation_report(results: Dict[str, Any], output_dir, target_metrics: Optional[Dict[str, float]] = None) -> None:
    out = Path(output_dir)
    out.mkdir(parents=True, exist_ok=True)
    (out / "evaluation_results.json").write_text(json.dumps(results, indent=2))
    (out / "evaluation_report.md").write_text(report_markdown(results, target_metrics))
