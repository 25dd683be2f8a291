"""CPU: pin the NumPy loss oracle (values and gradient) against what the reference
produced (tests/golden/loss_*.npz, edge_*.npz).  Smoothness is compared loosely: the
reference's neighbour search (matmul-form cdist + topk, losses.py:475-479) is itself
noisy in which of 21 candidates it drops."""
import numpy as np

from conftest import golden


def _rel(a, b):
    return abs(a - b) / max(abs(b), 1e-12)


def test_loss_values_and_gradient_match_reference():
    from oracle import loss_oracle
    g = golden("loss_b2_n2048_s4.npz")
    out = loss_oracle.instance_seg_loss(g["x"][:, :, :3], g["emb"], g["labels"])
    assert _rel(out["pull_loss"], float(g["loss.pull_loss"])) < 1e-5
    assert _rel(out["push_loss"], float(g["loss.push_loss"])) < 1e-4
    assert _rel(out["smoothness_loss"], float(g["loss.smoothness_loss"])) < 2e-3
    assert _rel(out["total_loss"], float(g["loss.total_loss"])) < 1e-3
    ref = g["demb"].astype(np.float64)
    err = np.abs(out["demb_total"] - ref).max() / np.abs(ref).max()
    assert err < 2e-2, err          # dominated by the few rows whose neighbour set differs
    # rows untouched by the neighbour noise agree tightly
    row_err = np.abs(out["demb_total"] - ref).max(-1) / np.abs(ref).max()
    assert np.median(row_err) < 1e-5


def test_loss_edge_cases_match_reference():
    from oracle import loss_oracle
    g = golden("edge_b2_n256_s9.npz")
    pts = g["x"][:, :, :3]
    for name in ("allbg", "one", "two_and_bg", "int32like"):
        out = loss_oracle.instance_seg_loss(pts, g["emb"], g[f"{name}.labels"])
        for k in ("pull_loss", "push_loss"):
            ref = float(g[f"{name}.{k}"])
            assert abs(out[k] - ref) <= 1e-5 * max(abs(ref), 1e-3), (name, k, out[k], ref)
        assert _rel(out["smoothness_loss"], float(g[f"{name}.smoothness_loss"])) < 5e-3
    out = loss_oracle.instance_seg_loss(pts, g["emb"], g["allbg.labels"])
    assert out["pull_loss"] == 0.0 and out["push_loss"] == 0.0            # losses.py:395-396
    out = loss_oracle.instance_seg_loss(pts, g["emb"], g["one.labels"])
    assert out["push_loss"] == 0.0                                         # losses.py:431
