import os
import sys

import numpy as np
import pytest

# the test process is the application here: particle groups on their own streams want more hardware queues than the HIP runtime's
# default of 4, and the runtime reads this at its first call (the package itself no longer touches the environment)
os.environ.setdefault("GPU_MAX_HW_QUEUES", "8")
REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
GOLDEN = os.path.join(REPO, "tests", "golden")
for p in (REPO, GOLDEN):
    if p not in sys.path:
        sys.path.insert(0, p)


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run on the GPU box with -m gpu)")


def load_golden(name):
    return np.load(os.path.join(GOLDEN, name))


@pytest.fixture(scope="session")
def intel_readings():
    """The bundled Intel log (910 scans x 180 beams) decoded from the compact fixture."""
    z = load_golden("intel_gfs.npz")
    rng = z["range_cm"].astype(np.float64) / 100.0
    pose = z["pose"]
    return [{"x": float(p[0]), "y": float(p[1]), "theta": float(p[2]), "range": list(map(float, r))}
            for p, r in zip(pose, rng)]


@pytest.fixture(scope="session")
def csail_readings():
    """The bundled CSAIL log (406 scans x 361 beams) decoded from the compact fixture."""
    z = load_golden("csail_gfs.npz")
    rng = z["range_cm"].astype(np.float64) / 100.0
    return [{"x": float(p[0]), "y": float(p[1]), "theta": float(p[2]), "range": list(map(float, r))}
            for p, r in zip(z["pose"], rng)]
