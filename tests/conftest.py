import os
import sys

import numpy as np
import pytest

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if REPO not in sys.path:
    sys.path.insert(0, REPO)
GOLDEN = os.path.join(REPO, "tests", "golden")


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


@pytest.fixture(scope="session")
def golden_geometry():
    return np.load(os.path.join(GOLDEN, "geometry.npz"))


@pytest.fixture(scope="session")
def golden_disparity():
    return np.load(os.path.join(GOLDEN, "disparity.npz"))


@pytest.fixture(scope="session")
def golden_correlation():
    return np.load(os.path.join(GOLDEN, "correlation.npz"))


@pytest.fixture(scope="session")
def golden_correlation_acmm():
    return np.load(os.path.join(GOLDEN, "correlation_acmm.npz"))


@pytest.fixture(scope="session")
def golden_tables():
    import json
    with open(os.path.join(GOLDEN, "tables.json")) as fh:
        return json.load(fh)
