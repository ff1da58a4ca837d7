from .metrics import (IoUMetrics, ConfidenceCalibration, EnsembleDisagreementMetrics, RobustnessMetrics,  # noqa: F401
                      ConfusionAccumulator, iou_from_counts)
