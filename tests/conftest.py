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
def golden_nn_crosspin():
    return np.load(os.path.join(GOLDEN, "nn_crosspin.npz"))


def crosspin_cases(z, tag, k):
    """Per item of a nn_crosspin.npz case: (queries (C, 784) = template cells, references (C, 784), wanted
    1-based indices (k, 784), mask (784,) of the template cells whose ranking is decided).  The reference's map
    values are float32 dot products of unit vectors scaled by one positive number per map, so a ranking counts as
    decided when every gap among its first k + 1 values exceeds 1e-4 of the best value (a few hundred float32
    ulps of a dot product, far above what the order of summation can move)."""
    f1, f2 = z[f"{tag}_f1"], z[f"{tag}_f2"]
    top, val = z[f"{tag}_top6"].astype(np.int64), z[f"{tag}_top6_values"].astype(np.float64)
    b, c = f1.shape[:2]
    for i in range(b):
        gaps = val[i, :, :k] - val[i, :, 1:k + 1]
        decided = (val[i, :, 0] > 0) & (gaps > 1e-4 * val[i, :, :1]).all(axis=1)
        yield (np.ascontiguousarray(f2[i].reshape(c, 784)), np.ascontiguousarray(f1[i].reshape(c, 784)),
               np.ascontiguousarray(top[i, :, :k].T + 1), decided)


def physical_zbuffer_case():
    """(surface samples (n, 3) of the reference's shipped bed mesh, pose, Blender's depth render of that mesh, the
    agreement measured when the fixture was made): see tests/golden/make_golden.py gen_zbuffer_physical."""
    import json
    from pose_estimation_amd import run_proj
    mesh = np.load(os.path.join(GOLDEN, "bed_mesh.npz"))
    with open(os.path.join(GOLDEN, "zbuffer_physical.json")) as fh:
        stats = json.load(fh)
    pts = run_proj.sample_surface(mesh["vertices"], mesh["faces"], stats["samples"], seed=stats["sample_seed"])
    shipped = np.load(os.path.join(GOLDEN, "demo_flow.npz"))["shipped_depth"]
    return pts, stats["pose"], shipped, stats


def physical_zbuffer_agreement(depth, shipped):
    cov, mask = np.isfinite(depth), shipped < 10.0        # Blender's depth pass is clamped to 10 behind the object
    both = cov & mask
    d = np.abs(depth[both] - shipped[both])
    return both.sum() / (cov | mask).sum(), float(np.median(d)), float((d < 0.01).mean())


@pytest.fixture(scope="session")
def golden_tables():
    import json
    with open(os.path.join(GOLDEN, "tables.json")) as fh:
        return json.load(fh)
