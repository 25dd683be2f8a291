"""GPU parity of the embedding-clustering row (csrc/p2n_cluster.hip) through the C ABI:
bit-exact labels against scikit-learn's output stored in tests/golden/dbscan_*.npz (the
reference's call, inference.py:124-140) and against the CPU oracle on larger seeded inputs;
row normalisation bit-exact against numpy."""
import glob
import os

import numpy as np
import pytest
import torch

from conftest import GOLDEN, golden

pytestmark = pytest.mark.gpu

CASES = sorted(os.path.basename(p)[:-4] for p in glob.glob(os.path.join(GOLDEN, "dbscan_*.npz")))


@pytest.mark.parametrize("name", CASES)
def test_golden_labels(name):
    from pointcloudsegmentation_amd.clustering import cluster_embeddings
    g = golden(name + ".npz")
    got = cluster_embeddings(g["emb"], float(g["eps"]), int(g["min_samples"]))
    assert got.dtype == np.int64
    np.testing.assert_array_equal(got, g["labels"])


@pytest.mark.parametrize("D", [1, 5, 8, 13, 31, 32, 33, 48, 64, 100, 127, 128, 129, 200, 255, 256])
def test_normalize_rows_bit_exact(D):
    from pointcloudsegmentation_amd.clustering import normalize_rows
    e = np.random.default_rng(D).standard_normal((1000, D)).astype(np.float32) * 3
    e[5] = 0.0
    want = e / (np.linalg.norm(e, axis=1, keepdims=True) + 1e-8)
    assert np.array_equal(normalize_rows(e), want)


@pytest.mark.parametrize("N,D,k,ms,eps", [(20000, 64, 60, 50, 0.1), (9000, 64, 33, 50, 0.1), (5000, 128, 10, 25, 0.1),
                                          (3000, 200, 6, 20, 0.12), (4097, 16, 20, 30, 0.08), (129, 64, 2, 5, 0.1),
                                          (5000, 32, 8, 25, 0.1), (4000, 100, 8, 25, 0.1)])
def test_vs_oracle(N, D, k, ms, eps):
    from oracle import oracle
    from pointcloudsegmentation_amd.clustering import cluster_embeddings
    from pointcloudsegmentation_amd.synth import make_embeddings
    e = make_embeddings(N, D, N + D, n_clusters=k)
    want, wcnt = oracle.dbscan(oracle.normalize_rows(e), eps, ms, return_counts=True)
    got, cnt = cluster_embeddings(torch.from_numpy(e).cuda(), eps, ms, return_counts=True)
    np.testing.assert_array_equal(cnt.cpu().numpy(), wcnt)
    np.testing.assert_array_equal(got.cpu().numpy(), want.astype(np.int64))


def test_threshold_band_is_decided_exactly():
    """Pairs engineered to sit within a few ulp of eps: the fp32 MFMA estimate cannot decide
    them, the float64 recomputation must, identically to the oracle."""
    from oracle import oracle
    from pointcloudsegmentation_amd.clustering import cluster_embeddings, last_cluster_stats
    rng = np.random.default_rng(3)
    n, eps = 1536, 0.1
    ang = np.zeros(n)
    ang[1:] = np.cumsum(np.full(n - 1, 2 * np.arcsin(eps / 2)) + rng.integers(-3, 4, n - 1) * 2e-9)  # chord ~ eps +- ulps
    e = np.zeros((n, 64), np.float32)
    e[:, 0], e[:, 1] = np.cos(ang), np.sin(ang)   # consecutive points: chord = eps up to rounding
    e = np.ascontiguousarray(e)
    # second family: a lattice along one axis with spacing exactly representable near eps
    lat = np.zeros((512, 64), np.float32)
    lat[:, 0] = 1.0
    lat[:, 1] = np.arange(512) * np.float32(0.1)
    for emb, ms in ((e, 3), (lat, 2)):
        want, wcnt = oracle.dbscan(oracle.normalize_rows(emb), eps, ms, return_counts=True)
        got, cnt = cluster_embeddings(emb, eps, ms, return_counts=True)
        np.testing.assert_array_equal(cnt, wcnt)
        np.testing.assert_array_equal(got, want.astype(np.int64))
        if emb is e:
            assert last_cluster_stats(len(emb), 64, ms)["exact_pass1"] >= len(emb) // 2   # the band path really ran


def test_full_scene_size_properties():
    """Reference scene size (166,499 points, BASELINE configs[0]): the oracle is too slow there,
    so check what DBSCAN guarantees: labels are 0..K with every cluster non-empty, every core point
    is labelled, and on a random row sample the neighbourhood size and the label rule hold
    (recomputed in float64 with torch)."""
    from pointcloudsegmentation_amd.clustering import cluster_embeddings, normalize_rows
    from pointcloudsegmentation_amd.synth import make_embeddings
    N, ms, eps = 166_499, 50, 0.1
    e = torch.from_numpy(make_embeddings(N, 64, 11, n_clusters=33)).cuda()
    lab, cnt = cluster_embeddings(e, eps, ms, return_counts=True)
    K = int(lab.max())
    assert int(lab.min()) >= 0 and torch.unique(lab).numel() in (K, K + 1)
    core = cnt >= ms
    assert bool((lab[core] > 0).all())
    x = normalize_rows(e).double()
    rows = torch.from_numpy(np.random.default_rng(0).choice(N, 256, replace=False)).cuda()
    d2 = ((x[rows][:, None, :] - x[None, :, :]) ** 2).sum(-1) if False else torch.cdist(x[rows], x).square()
    # cdist in float64 differs from the sequential sum by ~1e-16: compare away from the threshold
    near = (d2 - eps * eps).abs() < 1e-12
    assert not bool(near.any())
    nb = d2 <= eps * eps
    np.testing.assert_array_equal(nb.sum(1).cpu().numpy(), cnt[rows].cpu().numpy())
    for q, r in enumerate(rows.tolist()):
        cn = nb[q] & core
        if bool(core[r]):
            assert bool((lab[cn] == lab[r]).all())          # core neighbours of a core point share its cluster
        elif bool(cn.any()):
            assert int(lab[r]) == int(lab[cn].min())        # border: lowest-numbered neighbouring cluster
        else:
            assert int(lab[r]) == 0                         # noise


@pytest.mark.parametrize("sort", ["1", "0"])
@pytest.mark.parametrize("N,D,k,ms,eps,order", [(12000, 64, 20, 40, 0.1, "runs"), (12000, 64, 20, 40, 0.1, "random"),
                                                (6000, 200, 6, 20, 0.12, "runs"), (7000, 128, 8, 25, 0.1, "runs"),
                                                (300, 64, 3, 5, 0.1, "random")])
def test_working_order_changes_nothing(monkeypatch, sort, N, D, k, ms, eps, order):
    """The library clusters the rows in the order of a code of their embeddings (P2N_CL_SORT=0: in the
    given order); neighbourhood sizes and labels -- cluster NUMBERS included, which follow the caller's
    indices -- must be the oracle's either way, on index-coherent rows (block pruning at work) as on
    shuffled ones.  The D = 200 runs case is the regression test of the block centres beyond 128
    dimensions."""
    from oracle import oracle
    from pointcloudsegmentation_amd.clustering import cluster_embeddings
    from pointcloudsegmentation_amd.synth import make_embeddings
    monkeypatch.setenv("P2N_CL_SORT", sort)
    e = make_embeddings(N, D, N + D + 1, n_clusters=k, order=order)
    want, wcnt = oracle.dbscan(oracle.normalize_rows(e), eps, ms, return_counts=True)
    got, cnt = cluster_embeddings(torch.from_numpy(e).cuda(), eps, ms, return_counts=True)
    np.testing.assert_array_equal(cnt.cpu().numpy(), wcnt)
    np.testing.assert_array_equal(got.cpu().numpy(), want.astype(np.int64))
    assert int(got.max()) >= 2   # the case really has several clusters to number


def test_shuffled_rows_get_the_same_partition_numbered_by_first_index():
    """Full scene size, where the oracle is too slow: clustering a shuffled copy must give the same
    partition, its clusters numbered by their lowest core index in the shuffled order."""
    from pointcloudsegmentation_amd.clustering import cluster_embeddings
    from pointcloudsegmentation_amd.synth import make_embeddings
    N, ms, eps = 166_499, 50, 0.1
    e = torch.from_numpy(make_embeddings(N, 64, 11, n_clusters=33, order="runs")).cuda()
    lab, cnt = cluster_embeddings(e, eps, ms, return_counts=True)
    p = torch.from_numpy(np.random.default_rng(1).permutation(N)).cuda()
    lab2, cnt2 = cluster_embeddings(e[p].contiguous(), eps, ms, return_counts=True)
    assert torch.equal(cnt2, cnt[p])
    assert torch.equal(lab2 > 0, lab[p] > 0)
    K = int(lab.max())
    assert int(lab2.max()) == K
    core2 = cnt2 >= ms
    # map: cluster number in the shuffled run -> number in the original run, one value each
    pairs = torch.unique(torch.stack([lab2[lab2 > 0], lab[p][lab2 > 0]], 1), dim=0)
    assert pairs.shape[0] == K
    # numbering rule in the shuffled order: ascending lowest core index
    first = torch.full((K + 1,), N, dtype=torch.int64, device=e.device)
    idx = torch.arange(N, device=e.device)
    first.scatter_reduce_(0, lab2[core2], idx[core2], reduce="amin")
    assert bool((first[1:-1] < first[2:]).all())


@pytest.mark.parametrize("env", [{"P2N_CL_SYM": "0"}, {"P2N_CL_REACH": "0"}, {"P2N_CL_SYM": "0", "P2N_CL_SORT": "0"}, {"P2N_CL_NOTES": "0"}])
@pytest.mark.parametrize("N,D,k,ms,eps,order", [(12000, 64, 20, 40, 0.1, "random"), (6000, 200, 6, 20, 0.12, "runs"),
                                                (7000, 128, 8, 25, 0.1, "random")])
def test_pruning_stages_switched_off_change_nothing(monkeypatch, env, N, D, k, ms, eps, order):
    """The stages that decide WHICH block pairs the passes visit -- every pair once with counts for both
    sides (P2N_CL_SYM), the point-to-sphere test (P2N_CL_REACH), the working order -- can be switched off
    one by one; counts and labels must stay the oracle's (the default, all on, is what every other test
    of this file runs)."""
    from oracle import oracle
    from pointcloudsegmentation_amd.clustering import cluster_embeddings
    from pointcloudsegmentation_amd.synth import make_embeddings
    for name, val in env.items():
        monkeypatch.setenv(name, val)
    e = make_embeddings(N, D, N + D + 2, n_clusters=k, order=order)
    want, wcnt = oracle.dbscan(oracle.normalize_rows(e), eps, ms, return_counts=True)
    got, cnt = cluster_embeddings(torch.from_numpy(e).cuda(), eps, ms, return_counts=True)
    np.testing.assert_array_equal(cnt.cpu().numpy(), wcnt)
    np.testing.assert_array_equal(got.cpu().numpy(), want.astype(np.int64))


@pytest.mark.parametrize("env", [{}, {"P2N_CL_SYM": "0"}, {"P2N_CL_SORT": "0"}])
def test_many_undecided_pairs_per_row_overflow_the_queue(monkeypatch, env):
    """Two tight groups of 520 points whose every cross pair sits within ~1e-7 of eps: each row has hundreds of
    pairs the fp32 estimate cannot decide -- far more than the 16 queue slots of a lane, so the pair pass
    decides most of them on the spot in float64, for the row and (where a block pair is visited once) for the
    other point too.  Neighbourhood sizes and labels must be the oracle's; min_samples is set so that core
    status hangs on those decisions."""
    from oracle import oracle
    from pointcloudsegmentation_amd.clustering import cluster_embeddings, last_cluster_stats
    for name, val in env.items():
        monkeypatch.setenv(name, val)
    rng = np.random.default_rng(17)
    n, eps = 520, 0.1
    e = np.zeros((2 * n + 200, 64), np.float64)
    e[:n, 0] = 1.0                                          # group A: one direction (+ jitter far below the band)
    e[:n, 2:6] = rng.standard_normal((n, 4)) * 1e-9
    th = 2 * np.arcsin(eps * (1 + rng.uniform(-1e-7, 1e-7, n)) / 2)   # group B: chords of eps (1 +- 1e-7) from A
    e[n:2 * n, 0], e[n:2 * n, 1] = np.cos(th), np.sin(th)
    e[2 * n:] = rng.standard_normal((200, 64))              # outliers
    e = np.ascontiguousarray((e * rng.uniform(0.5, 2.0, (len(e), 1))).astype(np.float32))
    e = e[rng.permutation(len(e))]
    ms = n + n // 2                                         # a group alone is not enough, the exact cross pairs decide
    want, wcnt = oracle.dbscan(oracle.normalize_rows(e), eps, ms, return_counts=True)
    got, cnt = cluster_embeddings(e, eps, ms, return_counts=True)
    np.testing.assert_array_equal(cnt, wcnt)
    np.testing.assert_array_equal(got, want.astype(np.int64))
    assert last_cluster_stats(len(e), 64, ms)["exact_pass1"] > 16 * 64 * 8      # more than the queues of a workgroup hold
    assert 0 < int((wcnt >= ms).sum()) < 2 * n                                  # some rows core, some not: the decisions mattered
