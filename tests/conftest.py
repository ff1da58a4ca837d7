import os
import sys
from pathlib import Path

import numpy as np
import pytest

ROOT = Path(__file__).resolve().parent.parent
if str(ROOT) not in sys.path:
    sys.path.insert(0, str(ROOT))

GOLDEN = ROOT / "tests" / "golden"


# training-shape convolutions miss MIOpen's find-db on a fresh box and its default find mode then BENCHMARKS every
# candidate solver (minutes per shape at 1024x2048); FAST = find-db, else the heuristic pick.  Set before
# the library is first used (bench.py --mode train does the same).
os.environ.setdefault("MIOPEN_FIND_MODE", "FAST")


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run by the driver with -m gpu)")


def maxconf_flips_are_rounding_ties(got, ref, s1, s2, ulps=4.0):
    """max_confidence (PKG/models/model.py:447-455) selects by `softmax(s1).max > softmax(s2).max`, two float32 numbers
    whose last bit depends on the exponential's implementation (torch-CPU: Sleef; HIP: v_exp_f32; the C oracle: libm).
    Stated tolerance: the selection equals the reference's EXACTLY wherever the two confidences — computed here in
    float64 — differ by more than `ulps` float32 ulps; a pixel that differs must be such a rounding tie.  Returns the
    number of differing pixels."""
    import numpy as np
    s1, s2 = np.asarray(s1, np.float64), np.asarray(s2, np.float64)
    c1 = 1.0 / np.exp(s1 - s1.max(axis=1, keepdims=True)).sum(axis=1)
    c2 = 1.0 / np.exp(s2 - s2.max(axis=1, keepdims=True)).sum(axis=1)
    differ = (np.asarray(got) != np.asarray(ref)).any(axis=1)
    gap = np.abs(c1 - c2) / (np.maximum(c1, c2) * 2.0 ** -24)
    assert (gap[differ] <= ulps).all(), f"selection differs where the confidences are {gap[differ].max():.1f} ulp apart"
    n = int(differ.sum())
    # the flip count is part of the stated tolerance: printed with every use, visible under `pytest -s` / `-rP`
    print(f"max_confidence: {n} of {differ.size} pixels select the other member (all within {ulps:g} float32 ulp of a confidence tie)")
    return n


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
