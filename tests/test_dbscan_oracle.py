"""Pins the CPU oracle of the embedding-clustering row (oracle/p2n_oracle.c: orc_normalize_rows,
orc_dbscan) against scikit-learn's DBSCAN output stored in tests/golden/dbscan_*.npz (generated
by tests/golden/make_golden_dbscan.py: the call the reference makes at inference.py:124-140)."""
import glob
import os

import numpy as np
import pytest

from conftest import GOLDEN as GOLDEN_DIR, golden

CASES = sorted(os.path.basename(p)[:-4] for p in glob.glob(os.path.join(GOLDEN_DIR, "dbscan_*.npz")))


def test_fixture_list():
    assert len(CASES) >= 11


@pytest.mark.parametrize("name", CASES)
def test_oracle_matches_sklearn_labels(name):
    from oracle import oracle
    g = golden(name + ".npz")
    emb = g["emb"]
    x = emb / (np.linalg.norm(emb, axis=1, keepdims=True) + 1e-8)        # reference inference.py:124
    xn = oracle.normalize_rows(emb)
    assert xn.dtype == np.float32 and np.array_equal(xn, x), "normalisation must be bit-identical to numpy's"
    lab = oracle.dbscan(xn, float(g["eps"]), int(g["min_samples"]))
    np.testing.assert_array_equal(lab.astype(np.int64), g["labels"])


@pytest.mark.parametrize("D", [1, 5, 8, 13, 64, 127, 128, 129, 256, 300])
def test_normalise_order_matches_numpy(D):
    from oracle import oracle
    e = np.random.default_rng(D).standard_normal((257, D)).astype(np.float32) * 3
    want = e / (np.linalg.norm(e, axis=1, keepdims=True) + 1e-8)
    assert np.array_equal(oracle.normalize_rows(e), want)
