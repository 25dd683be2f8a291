import os
import sys

import numpy as np
import pytest

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if REPO not in sys.path:
    sys.path.insert(0, REPO)
GOLDEN = os.path.join(REPO, "tests", "golden")


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run by the driver with -m gpu)")


def golden(name):
    return np.load(os.path.join(GOLDEN, name))


@pytest.fixture(scope="session")
def ref_weights():
    """state_dict of the reference's shipped checkpoint (tests/golden/weights.npz)."""
    return dict(np.load(os.path.join(GOLDEN, "weights.npz")))


@pytest.fixture(scope="session")
def ref_params(ref_weights):
    from pointcloudsegmentation_amd.params import flatten_numpy
    return flatten_numpy(ref_weights)


def assert_close_to_max(got, ref, tol, what=""):
    """|got-ref| <= tol * max|ref| over the tensor: the tolerance form DESIGN.md
    states for fp32 features/embeddings (outputs cross zero, so a pure
    element-wise relative error is meaningless; SURVEY.md section 7 'Hard parts')."""
    got = np.asarray(got, dtype=np.float64)
    ref = np.asarray(ref, dtype=np.float64)
    assert got.shape == ref.shape, (what, got.shape, ref.shape)
    scale = max(np.abs(ref).max(), 1e-30) if ref.size else 1.0
    err = np.abs(got - ref).max() / scale if ref.size else 0.0
    assert err <= tol, f"{what}: max|diff|/max|ref| = {err:.3e} > {tol:.1e}"
    return err
