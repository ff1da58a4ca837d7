#!/usr/bin/env python3
"""Training entry point — counterpart of REF/scripts/train.py (`--config --resume --device --seed
--output-dir`, writes training_results.json).  Under `torch.distributed.run` each rank trains on
its shard and gradients are averaged in buckets over RCCL."""
import argparse
import json
import logging
import os
import random
import sys
from pathlib import Path

ROOT = Path(__file__).resolve().parent.parent
sys.path.insert(0, str(ROOT))
# Training shapes miss MIOpen's find-db on a fresh machine; its default (hybrid) find mode then BENCHMARKS every candidate solver of
# every backward convolution — the first step at 1024x2048 did not finish in 7 minutes.  FAST = find-db, else the
# heuristic pick.  Set before torch loads MIOpen; an explicit MIOPEN_FIND_MODE in the environment wins.
os.environ.setdefault("MIOPEN_FIND_MODE", "FAST")

import numpy as np
import torch

from adverse_weather_semantic_segmentation_robustness_benchmark_amd import parallel
from adverse_weather_semantic_segmentation_robustness_benchmark_amd.data.loader import CityscapesKITTIDataset, create_dataloader
from adverse_weather_semantic_segmentation_robustness_benchmark_amd.models.model import DeepLabV3PlusModel, EnsembleModel, SegFormerModel
from adverse_weather_semantic_segmentation_robustness_benchmark_amd.training.trainer import AdverseWeatherTrainer
from adverse_weather_semantic_segmentation_robustness_benchmark_amd.utils.config import (create_default_config, get_device_config, load_config,
                                                                                    setup_logging, validate_config)

logger = logging.getLogger("train")


def set_seed(seed):
    """REF/scripts/train.py:39-59."""
    random.seed(seed); np.random.seed(seed); torch.manual_seed(seed)
    if torch.cuda.is_available():
        torch.cuda.manual_seed_all(seed)


def create_model(config):
    kind = config.get("model.type", "ensemble")
    nc, depth = config.get("model.num_classes", 19), config.get("model.include_depth", True)
    if kind == "segformer":
        return SegFormerModel(num_classes=nc, include_depth=depth)
    if kind == "deeplabv3plus":
        return DeepLabV3PlusModel(num_classes=nc, include_depth=depth)
    if kind == "ensemble":
        return EnsembleModel(num_classes=nc, include_depth=depth, ensemble_strategy=config.get("model.ensemble_strategy", "weighted_average"),
                             temperature_scaling=config.get("model.temperature_scaling", True))
    raise ValueError(f"Unknown model type: {kind}")


def main():
    ap = argparse.ArgumentParser(description="Train adverse-weather segmentation model (MI355X-native path)")
    ap.add_argument("--config", type=str, default=None)
    ap.add_argument("--resume", type=str, default=None)
    ap.add_argument("--device", type=str, default=None)
    ap.add_argument("--seed", type=int, default=None)
    ap.add_argument("--output-dir", type=str, default="outputs")
    args = ap.parse_args()
    try:
        config = load_config(args.config) if args.config else create_default_config()
        if args.device:
            config.set("device", args.device)
        if args.seed is not None:
            config.set("seed", args.seed)
        setup_logging(config)
        validate_config(config)
        rank, local, world = parallel.init_from_env()
        # the SAME seed on every rank while the model is built (identical initial weights; the trainer also broadcasts
        # rank 0's parameters); the per-rank seed applies afterwards, to the host-side noise streams only
        set_seed(config.get("seed", 42))
        dev = get_device_config(config.get("device", "auto"))
        device = torch.device(dev, local) if dev.startswith("cuda") and ":" not in dev else torch.device(dev)
        model = create_model(config)
        set_seed(config.get("seed", 42) + rank)
        size = tuple(config.get("data.image_size", [512, 1024]))
        conds = config.get("data.weather_conditions")
        bs = config.get("training.batch_size", 8)
        train_ds = CityscapesKITTIDataset(split="train", image_size=size, weather_conditions=conds, device=device, seed=config.get("seed", 42))
        val_ds = CityscapesKITTIDataset(split="val", image_size=size, weather_conditions=conds, apply_augmentation=False, device=device)
        out = Path(args.output_dir)
        trainer = AdverseWeatherTrainer(model, create_dataloader(train_ds, bs, shuffle=True, rank=rank, world_size=world),
                                        create_dataloader(val_ds, bs, shuffle=False, rank=rank, world_size=world), config.to_dict(), device,
                                        checkpoint_dir=str(out / config.get("paths.checkpoints", "checkpoints")),
                                        log_dir=str(out / config.get("paths.logs", "logs")))
        results = trainer.resume_training(args.resume) if args.resume else trainer.train()
        if rank == 0:
            out.mkdir(parents=True, exist_ok=True)
            (out / "training_results.json").write_text(json.dumps(results, indent=2, default=float))
    except Exception as e:  # noqa: BLE001 - train.py:322-324
        logger.error("Training failed: %s", e)
        raise SystemExit(1)


if __name__ == "__main__":
    main()
