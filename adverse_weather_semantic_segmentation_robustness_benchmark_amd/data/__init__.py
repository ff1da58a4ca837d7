"""Data side of the hot path (PKG/data/__init__.py:3-11 exports the same four names)."""
import importlib

_LAZY = {"CityscapesKITTIDataset": ".loader", "WeatherAugmentationPipeline": ".loader", "create_dataloader": ".loader",
         "WeatherDegradationTransforms": ".preprocessing", "DepthEstimationPreprocessor": ".preprocessing"}
__all__ = sorted(_LAZY)


def __getattr__(name):
    if name in _LAZY:
        return getattr(importlib.import_module(_LAZY[name], __name__), name)
    raise AttributeError(f"module {__name__!r} has no attribute {name!r}")
