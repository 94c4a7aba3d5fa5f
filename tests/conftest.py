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
def golden_demo_flow():
    return np.load(os.path.join(GOLDEN, "demo_flow.npz"))


def dense_flow(z, tag):
    """The (540, 960, 2) flow map of a demo_flow.npz case from its sparse storage, checked against its SHA-256."""
    import hashlib
    flow = np.zeros((540 * 960, 2))
    flow[z[f"{tag}_flow_pixels"]] = z[f"{tag}_flow_values"]
    flow = flow.reshape(540, 960, 2)
    assert hashlib.sha256(flow.tobytes()).digest() == z[f"{tag}_flow_sha256"].tobytes()
    return flow


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
