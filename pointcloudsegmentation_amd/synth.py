"""Deterministic synthetic point clouds (SURVEY.md §8d "Synthetic inputs").

numpy-only so the same seed gives the same bytes in the golden generator, the
tests, the oracle timing leg and bench.py, on any torch version.

"ScanNet-shaped": a room box with 60 % of the points snapped to six planes
(floor / ceiling / four walls) plus a little noise, then the reference's own
input normalisation (reference inference.py:88-91: subtract the centroid,
divide by max |.|), rgb ~ U[0,1].  "LiDAR-shaped": 64 rings, range U[2,80] m.
"""
import numpy as np


def _normalise_xyz(xyz: np.ndarray) -> np.ndarray:
    # reference inference.py:88-91, in float64 like the reference, cast at the end
    xyz = xyz - xyz.mean(axis=0)
    scale = np.max(np.abs(xyz))
    if scale > 0:
        xyz = xyz / scale
    return xyz


def scannet_like_cloud(n: int, rng: np.random.Generator) -> np.ndarray:
    """One cloud, (n, 6) float32: xyz normalised to the unit inf-ball, rgb in [0,1]."""
    box = np.array([8.0, 6.0, 3.0])
    xyz = rng.random((n, 3)) * box
    snap = rng.random(n) < 0.6
    plane = rng.integers(0, 6, size=n)
    axis = plane // 2
    side = (plane % 2).astype(np.float64)
    rows = np.nonzero(snap)[0]
    xyz[rows, axis[rows]] = side[rows] * box[axis[rows]]
    xyz += rng.normal(0.0, 0.01, size=(n, 3))
    xyz = _normalise_xyz(xyz)
    rgb = rng.random((n, 3))
    return np.concatenate([xyz, rgb], axis=1).astype(np.float32)


def lidar_like_cloud(n: int, rng: np.random.Generator) -> np.ndarray:
    """One cloud, (n, 6) float32, 64-ring spinning-LiDAR geometry."""
    ring = rng.integers(0, 64, size=n)
    elev = np.deg2rad(-24.8 + ring * (26.8 / 63.0))
    azim = rng.random(n) * 2.0 * np.pi
    rng_m = 2.0 + rng.random(n) * 78.0
    xyz = np.stack([rng_m * np.cos(elev) * np.cos(azim),
                    rng_m * np.cos(elev) * np.sin(azim),
                    rng_m * np.sin(elev)], axis=1)
    xyz = _normalise_xyz(xyz)
    rgb = rng.random((n, 3))
    return np.concatenate([xyz, rgb], axis=1).astype(np.float32)


def make_batch(b: int, n: int, seed: int, kind: str = "scannet") -> np.ndarray:
    """(b, n, 6) float32 contiguous batch; cloud i uses stream (seed, i)."""
    gen = scannet_like_cloud if kind == "scannet" else lidar_like_cloud
    out = np.empty((b, n, 6), dtype=np.float32)
    for i in range(b):
        out[i] = gen(n, np.random.default_rng([seed, i]))
    return out


def make_labels(b: int, n: int, seed: int, num_instances: int = 33) -> np.ndarray:
    """(b, n) int64 instance labels in [0, num_instances) (0 = background);
    33.2 instances/scene on average in the reference's dataset
    (reference analysis_results/dataset_statistics.txt:7)."""
    rng = np.random.default_rng([seed, 7919])
    return rng.integers(0, num_instances, size=(b, n)).astype(np.int64)


def make_embeddings(n: int, d: int, seed: int, n_clusters: int = 33, noise_frac: float = 0.1,
                    spread=(0.02, 0.12), intrinsic_dim: int = 4, order: str = "random") -> np.ndarray:
    """Instance-embedding-shaped rows (n, d) fp32: `n_clusters` directions on the unit sphere,
    each with its own spread inside a random `intrinsic_dim`-dimensional subspace (density falls
    off smoothly, so at DBSCAN eps = 0.1 there are dense cores, border points, noise tails and
    clusters that touch), plus `noise_frac` of uniformly random directions; rows carry
    random positive lengths so the caller's L2 normalisation matters.  Cluster sizes are
    uneven (ScanNet instances: a few large structures, many small objects).
    order="random": rows shuffled (worst case for any locality); order="runs": rows of one
    instance come in runs of 64-2048, noise rows scattered between them -- the index coherence of
    a scanned mesh, whose vertex order follows the surface."""
    rng = np.random.default_rng(seed)
    n_noise = int(n * noise_frac)
    n_in = n - n_noise
    centres = rng.standard_normal((max(n_clusters, 1), d))
    centres /= np.linalg.norm(centres, axis=1, keepdims=True)
    w = rng.gamma(1.0, 1.0, size=max(n_clusters, 1)) + 0.05
    which = rng.choice(max(n_clusters, 1), size=n_in, p=w / w.sum())
    sig = rng.uniform(spread[0], spread[1], size=max(n_clusters, 1))
    q = max(1, min(intrinsic_dim, d))
    basis = rng.standard_normal((max(n_clusters, 1), q, d))
    basis /= np.linalg.norm(basis, axis=2, keepdims=True)
    low = rng.standard_normal((n_in, q)) * (sig[which] / np.sqrt(2.0 * q))[:, None]   # typical pair distance = sig
    pts = centres[which] + np.einsum("nq,nqd->nd", low, basis[which])
    noise = rng.standard_normal((n_noise, d))
    allp = np.concatenate([pts, noise], 0)
    allp /= np.linalg.norm(allp, axis=1, keepdims=True)
    allp *= rng.uniform(0.5, 3.0, size=(n, 1))
    if order == "random":
        perm = rng.permutation(n)
    elif order == "runs":
        src = [list(rng.permutation(np.flatnonzero(which == c))) for c in range(max(n_clusters, 1))]
        noise_ids = list(rng.permutation(np.arange(n_in, n)))
        perm = []
        live = [c for c in range(len(src)) if src[c]]
        while live or noise_ids:
            if live:
                c = live[int(rng.integers(len(live)))]
                k = int(rng.integers(64, 2049))
                perm.extend(src[c][:k]); del src[c][:k]
                if not src[c]:
                    live.remove(c)
            k = int(rng.integers(0, 1 + max(1, int(0.25 * n_noise / max(n_in / 1000, 1)))))
            perm.extend(noise_ids[:k]); del noise_ids[:k]
        perm = np.asarray(perm)
    else:
        raise ValueError(f"unknown order {order!r}")
    return np.ascontiguousarray(allp[perm], dtype=np.float32)
