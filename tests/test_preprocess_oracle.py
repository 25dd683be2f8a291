"""The numpy restatement of reference inference.py:76-104 behaves as that code documents."""
import numpy as np


def test_preprocess_oracle_properties():
    from oracle.preprocess_oracle import preprocess_point_cloud
    rng = np.random.default_rng(0)
    pts = rng.uniform(-3, 9, (5000, 3))
    col = rng.integers(0, 256, (5000, 3)).astype(np.float64)
    f, c, s = preprocess_point_cloud(pts, col)
    assert f.shape == (1, 5000, 6) and f.dtype == np.float32
    np.testing.assert_allclose(c, pts.mean(0))
    assert abs(np.abs(f[0, :, :3]).max() - 1.0) < 1e-7 and abs(f[0, :, :3].mean()) < 1e-6
    assert f[0, :, 3:].max() <= 1.0                       # bytes were scaled by 1/255
    f2, _, _ = preprocess_point_cloud(pts, None)
    assert np.all(f2[0, :, 3:] == 0.5)
    np.testing.assert_allclose(f[0, :, :3] * s + c, pts, atol=1e-5)   # the inverse map the caller keeps
