#!/usr/bin/env python3
"""Evaluation entry point — counterpart of REF/scripts/evaluate.py (same positional/flag CLI:
`evaluate.py <checkpoint> [--config C] [--output-dir D] [--device DEV]`, same result keys, same
evaluation_results.json).  Under `torch.distributed.run` every rank evaluates its own shard of the
test set and the counters are all-reduced over RCCL.

    python scripts/evaluate.py checkpoints/best.pth --config configs/default.yaml
    python scripts/evaluate.py none --config configs/default.yaml        # random-init model (no checkpoint)
"""
import argparse
import json
import logging
import sys
from pathlib import Path

ROOT = Path(__file__).resolve().parent.parent
sys.path.insert(0, str(ROOT))

import torch

from adverse_weather_semantic_segmentation_robustness_benchmark_amd import parallel
from adverse_weather_semantic_segmentation_robustness_benchmark_amd.data.loader import CityscapesKITTIDataset, create_dataloader
from adverse_weather_semantic_segmentation_robustness_benchmark_amd.evaluation.harness import evaluate_model
from adverse_weather_semantic_segmentation_robustness_benchmark_amd.evaluation.metrics import RobustnessMetrics
from adverse_weather_semantic_segmentation_robustness_benchmark_amd.evaluation.report import generate_evaluation_report
from adverse_weather_semantic_segmentation_robustness_benchmark_amd.utils.checkpoint import load_model_state
from adverse_weather_semantic_segmentation_robustness_benchmark_amd.models.model import DeepLabV3PlusModel, EnsembleModel, SegFormerModel
from adverse_weather_semantic_segmentation_robustness_benchmark_amd.utils.config import (create_default_config, get_device_config, load_config,
                                                                                    setup_logging)

logger = logging.getLogger("evaluate")


def load_model(config, checkpoint_path, device):
    """REF/scripts/evaluate.py:42-86."""
    kind = config.get("model.type", "ensemble")
    nc, depth = config.get("model.num_classes", 19), config.get("model.include_depth", True)
    if kind == "segformer":
        model = SegFormerModel(num_classes=nc, include_depth=depth)
    elif kind == "deeplabv3plus":
        model = DeepLabV3PlusModel(num_classes=nc, include_depth=depth)
    elif kind == "ensemble":
        model = EnsembleModel(num_classes=nc, include_depth=depth, ensemble_strategy=config.get("model.ensemble_strategy", "weighted_average"),
                              temperature_scaling=config.get("model.temperature_scaling", True))
    else:
        raise ValueError(f"Unknown model type: {kind}")
    if checkpoint_path and str(checkpoint_path).lower() != "none":
        ckpt = torch.load(checkpoint_path, map_location=device, weights_only=False)
        load_model_state(model, ckpt)            # incl. the transformers-version key translation
    return model.to(device).eval()


def main():
    ap = argparse.ArgumentParser(description="Evaluate adverse-weather segmentation model (MI355X-native path)")
    ap.add_argument("checkpoint", type=str)
    ap.add_argument("--config", type=str, default=None)
    ap.add_argument("--output-dir", type=str, default="evaluation_results")
    ap.add_argument("--device", type=str, default="auto")
    args = ap.parse_args()
    try:
        config = load_config(args.config) if args.config else create_default_config()
        setup_logging(config)
        rank, local, world = parallel.init_from_env()
        dev = get_device_config(args.device if args.device != "auto" else config.get("device", "auto"))
        device = torch.device(dev, local) if dev.startswith("cuda") and ":" not in dev else torch.device(dev)
        model = load_model(config, args.checkpoint, device)
        ds = CityscapesKITTIDataset(data_root=config.get("data.data_root", "data"), split="test",
                                    image_size=tuple(config.get("data.image_size", [512, 1024])),
                                    weather_conditions=config.get("data.weather_conditions"), apply_augmentation=False,
                                    include_depth=config.get("data.include_depth", True), device=device)
        loader = create_dataloader(ds, batch_size=config.get("training.batch_size", 8), shuffle=False, rank=rank, world_size=world)
        metrics = RobustnessMetrics(num_classes=config.get("model.num_classes", 19), weather_conditions=config.get("data.weather_conditions"))
        results = evaluate_model(model, loader, metrics, device, config)
        if rank == 0:
            generate_evaluation_report({k: float(v) for k, v in results.items()}, Path(args.output_dir))   # json + markdown, :277-392
            for k, v in results.items():
                logger.info("%s: %.4f", k, v)
    except Exception as e:  # noqa: BLE001 - the reference converts failures to exit code 1 (evaluate.py:506-508)
        logger.error("Evaluation failed: %s", e)
        raise SystemExit(1)


if __name__ == "__main__":
    main()
