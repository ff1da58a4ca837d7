"""MI355X-native hot path of the adverse-weather segmentation robustness benchmark.

Drop-in counterparts of the reference's public names (REF/src/.../__init__.py:12-46), backed by
hand-written HIP kernels for gfx950 behind a C ABI (include/awseg.h, libawseg_hip.so).
There is no CPU fallback: the kernels fail loudly when the library or a GPU is missing.
Public names are imported lazily so that `import <pkg>._native` stays cheap.
"""
import importlib

__version__ = "0.1.0"

_LAZY = {
    "Config": ".utils.config",
    "SegFormerModel": ".models.model", "DeepLabV3PlusModel": ".models.model", "EnsembleModel": ".models.model",
    "FogDensityAwareLoss": ".models.model", "DepthEstimationHead": ".models.model",
    "RobustnessMetrics": ".evaluation.metrics", "IoUMetrics": ".evaluation.metrics",
    "ConfidenceCalibration": ".evaluation.metrics", "EnsembleDisagreementMetrics": ".evaluation.metrics",
    "AdverseWeatherTrainer": ".training.trainer", "EarlyStopping": ".training.trainer",
    "WeatherDegradationTransforms": ".data.preprocessing", "CityscapesKITTIDataset": ".data.loader",
    "WeatherAugmentationPipeline": ".data.loader", "DepthEstimationPreprocessor": ".data.preprocessing",
}
__all__ = sorted(_LAZY)


def __getattr__(name):
    if name in _LAZY:
        return getattr(importlib.import_module(_LAZY[name], __name__), name)
    raise AttributeError(f"module {__name__!r} has no attribute {name!r}")
