import os
import sys
from pathlib import Path

import numpy as np
import pytest

ROOT = Path(__file__).resolve().parent.parent
if str(ROOT) not in sys.path:
    sys.path.insert(0, str(ROOT))

GOLDEN = ROOT / "tests" / "golden"


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run by the driver with -m gpu)")


@pytest.fixture(scope="session")
def golden_weather():
    return np.load(GOLDEN / "weather.npz")


@pytest.fixture(scope="session")
def golden_metrics():
    return np.load(GOLDEN / "metrics.npz")


@pytest.fixture(scope="session")
def golden_depth():
    return np.load(GOLDEN / "depth.npz")


@pytest.fixture(scope="session")
def golden_model():
    return np.load(GOLDEN / "model.npz")


@pytest.fixture(scope="session")
def golden_trainer():
    return np.load(GOLDEN / "trainer.npz")


@pytest.fixture(scope="session")
def oracle():
    from oracle import cpu_oracle
    cpu_oracle.build()
    return cpu_oracle


@pytest.fixture(scope="session")
def native():
    """The HIP C-ABI library; GPU tests call the product through it."""
    import torch
    if not torch.cuda.is_available():
        pytest.skip("no GPU")
    from adverse_weather_semantic_segmentation_robustness_benchmark_amd import _native
    return _native
