"""Configuration plumbing the drop-in scripts need (restated small; not on the hot path).

Semantics follow REF/src/.../utils/config.py: dot-path get/set (:30-68), deep-merge update
(:70-104), `CONFIG_SECTION__KEY=value` environment overrides with bool/int/float parsing
(:191-251), the default table of REF/configs/default.yaml, `get_device_config` (:354-375) and
`validate_config` (:402-442).  One quirk is kept on purpose and documented in DESIGN.md: the
trainer is handed `config.to_dict()` and reads TOP-LEVEL keys such as `epochs` / `grad_clip`
(SURVEY §5), so values under `training:` do not reach it.
"""
from __future__ import annotations

import copy
import logging
import os
from pathlib import Path
from typing import Any, Dict, Optional, Union

import yaml

logger = logging.getLogger(__name__)

_DEFAULTS_YAML = """
model: {type: ensemble, num_classes: 19, include_depth: true, ensemble_strategy: weighted_average, temperature_scaling: true}
data: {dataset_type: combined, data_root: data, image_size: [512, 1024], weather_conditions: [clean, fog, rain, snow, night], apply_augmentation: true, include_depth: true}
training: {batch_size: 2, epochs: 100, num_workers: 4, pin_memory: true, grad_clip: 1.0}
optimizer: {type: adamw, learning_rate: 0.001, weight_decay: 0.01, betas: [0.9, 0.999]}
scheduler: {enabled: true, type: cosine, eta_min: 0.000001}
loss: {type: fog_density_aware, base_loss: cross_entropy, depth_weight: 0.5, fog_sensitivity: 2.0, depth_loss_weight: 0.1}
early_stopping: {patience: 10, min_delta: 0.001, restore_best_weights: true}
mlflow: {enabled: true, experiment_name: adverse_weather_segmentation, run_name: null}
evaluation: {num_bins: 15, weather_conditions: [clean, fog, rain, snow, night]}
logging: {level: INFO, format: '%(asctime)s - %(name)s - %(levelname)s - %(message)s'}
paths: {checkpoints: checkpoints, logs: logs, results: results}
device: auto
seed: 42
"""


def _walk(tree: Dict[str, Any], dotted: str, create: bool = False):
    """-> (parent dict, last key) for a dot path; parent is None when a segment is missing."""
    *parents, leaf = dotted.split(".")
    node = tree
    for seg in parents:
        if not isinstance(node, dict):
            return None, leaf
        if seg not in node:
            if not create:
                return None, leaf
            node[seg] = {}
        node = node[seg]
    return (node if isinstance(node, dict) else None), leaf


def _merge(base: Dict[str, Any], extra: Dict[str, Any]) -> Dict[str, Any]:
    out = dict(base)
    for k, v in extra.items():
        out[k] = _merge(out[k], v) if isinstance(out.get(k), dict) and isinstance(v, dict) else v
    return out


class Config:
    def __init__(self, config_dict: Optional[Dict[str, Any]] = None) -> None:
        self._config = config_dict or {}

    def get(self, key: str, default: Any = None) -> Any:
        parent, leaf = _walk(self._config, key)
        if parent is None or leaf not in parent:
            return default
        return parent[leaf]

    def set(self, key: str, value: Any) -> None:
        parent, leaf = _walk(self._config, key, create=True)
        parent[leaf] = value

    def update(self, other_config: Union["Config", Dict[str, Any]]) -> None:
        other = other_config._config if isinstance(other_config, Config) else other_config
        self._config = _merge(self._config, other)

    def to_dict(self) -> Dict[str, Any]:
        return self._config.copy()

    def __getitem__(self, key: str) -> Any:
        return self.get(key)

    def __setitem__(self, key: str, value: Any) -> None:
        self.set(key, value)

    def __contains__(self, key: str) -> bool:
        return self.get(key) is not None

    def __repr__(self) -> str:
        return f"Config({self._config})"


def _parse_env_value(text: str):
    low = text.lower()
    if low in ("true", "false"):
        return low == "true"
    for cast in (int, float):
        try:
            return cast(text)
        except ValueError:
            continue
    return text


def _apply_env_overrides(tree: Dict[str, Any]) -> Dict[str, Any]:
    for name, raw in os.environ.items():
        if name.startswith("CONFIG_"):
            dotted = name[len("CONFIG_"):].lower().replace("__", ".")
            parent, leaf = _walk(tree, dotted, create=True)
            parent[leaf] = _parse_env_value(raw)
    return tree


def load_config(config_path: Union[str, Path]) -> Config:
    path = Path(config_path)
    if not path.exists():
        raise FileNotFoundError(f"Configuration file not found: {path}")
    try:
        tree = yaml.safe_load(path.read_text(encoding="utf-8"))
    except yaml.YAMLError as e:
        raise yaml.YAMLError(f"Error parsing configuration file {path}: {e}")
    return Config(_apply_env_overrides(tree or {}))


def save_config(config: Config, config_path: Union[str, Path]) -> None:
    path = Path(config_path)
    path.parent.mkdir(parents=True, exist_ok=True)
    path.write_text(yaml.safe_dump(config.to_dict(), default_flow_style=False, indent=2), encoding="utf-8")


def create_default_config() -> Config:
    return Config(copy.deepcopy(yaml.safe_load(_DEFAULTS_YAML)))


def get_device_config(device_setting: str = "auto") -> str:
    if device_setting != "auto":
        return device_setting
    import torch
    return "cuda" if torch.cuda.is_available() else "cpu"   # ROCm torch answers True on MI355X


def setup_logging(config: Config) -> None:
    lc = config.get("logging", {}) or {}
    level = getattr(logging, str(lc.get("level", "INFO")).upper(), logging.INFO)
    logging.basicConfig(level=level, format=lc.get("format", "%(asctime)s - %(name)s - %(levelname)s - %(message)s"), force=True)


def validate_config(config: Config) -> None:
    for field in ("model.num_classes", "data.image_size", "training.batch_size", "training.epochs", "optimizer.learning_rate"):
        if config.get(field) is None:
            raise ValueError(f"Required configuration field missing: {field}")
    for field in ("model.num_classes", "training.batch_size", "training.epochs", "optimizer.learning_rate"):
        if config.get(field, 0) <= 0:
            raise ValueError(f"{field} must be positive")
    size = config.get("data.image_size")
    if not isinstance(size, list) or len(size) != 2:
        raise ValueError("data.image_size must be a list of two integers [height, width]")
