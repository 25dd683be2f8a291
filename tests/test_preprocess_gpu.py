"""GPU parity of the input normalisation (csrc/p2n_prep.hip, reference inference.py:76-104)
against its numpy restatement (oracle/preprocess_oracle.py): bit-exact features, centroid, scale."""
import numpy as np
import pytest

pytestmark = pytest.mark.gpu


def _scene(n, seed, colors):
    rng = np.random.default_rng(seed)
    pts = rng.uniform(0, 1, (n, 3)) * np.array([8.0, 6.0, 3.0]) + np.array([12.5, -3.25, 0.75])
    if colors == "unit":
        col = rng.uniform(0, 1, (n, 3))
    elif colors == "byte":
        col = rng.integers(0, 256, (n, 3)).astype(np.float64)
    else:
        col = None
    return pts, col


@pytest.mark.parametrize("n", [1, 2, 3, 2047, 2048, 2049, 40000, 166499])
@pytest.mark.parametrize("colors", [None, "unit", "byte"])
def test_bit_exact(n, colors):
    from oracle.preprocess_oracle import preprocess_point_cloud as ref
    from pointcloudsegmentation_amd.preprocess import preprocess_point_cloud
    pts, col = _scene(n, n, colors)
    with np.errstate(all="ignore"):
        wf, wc, ws = ref(pts, col)
    gf, gc, gs = preprocess_point_cloud(pts, col)
    assert gf.shape == (1, n, 6) and gf.dtype.is_floating_point and gf.is_cuda
    np.testing.assert_array_equal(gc, wc)
    np.testing.assert_array_equal(np.float64(gs), np.float64(ws))
    np.testing.assert_array_equal(gf.cpu().numpy(), wf)      # NaN == NaN positions too (n = 1: 0/0)


def test_float32_input_and_errors():
    from oracle.preprocess_oracle import preprocess_point_cloud as ref
    from pointcloudsegmentation_amd.preprocess import preprocess_point_cloud
    pts, col = _scene(1000, 5, "unit")
    p32 = pts.astype(np.float32)
    gf, gc, gs = preprocess_point_cloud(p32, col)
    wf, wc, ws = ref(p32.astype(np.float64), col)
    np.testing.assert_array_equal(gf.cpu().numpy(), wf)
    with pytest.raises(ValueError):
        preprocess_point_cloud(pts[:, :2])
    with pytest.raises(ValueError):
        preprocess_point_cloud(pts, col[:10])
