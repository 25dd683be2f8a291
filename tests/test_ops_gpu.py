"""GPU parity of the point-set operators against the CPU oracle (bit-exact indices;
oracle/p2n_oracle.c is their spec: the reference has no such operators, so this
parity is 'unpinned' except for kNN, which is also checked against the reference's
own brute-force neighbour sets stored in tests/golden/loss_b2_n2048_s4.npz)."""
import numpy as np
import pytest
import torch

from conftest import golden

pytestmark = pytest.mark.gpu


def _cloud(B, N, seed, kind="scannet"):
    from pointcloudsegmentation_amd.synth import make_batch
    return np.ascontiguousarray(make_batch(B, N, seed, kind)[:, :, :3])


def _grid_cloud(B, side):
    """Integer lattice: masses of exactly tied distances -> exercises the tie-break."""
    g = np.stack(np.meshgrid(np.arange(side), np.arange(side), np.arange(side), indexing="ij"), -1).reshape(-1, 3)
    out = np.stack([np.roll(g, b * 7, axis=0) for b in range(B)]).astype(np.float32)
    return out


# ------------------------------------------------------------------ FPS
@pytest.mark.parametrize("B,N,M,start", [(1, 1, 1, 0), (2, 5, 5, 3), (3, 1000, 64, 0), (2, 1024, 256, 17),
                                         (16, 8192, 512, 0), (2, 13000, 300, 5), (1, 20480, 200, 0),
                                         (2, 12288, 200, 1), (1, 12289, 200, 12288), (2, 3, 3, 2), (1, 600, 600, 0)])
def test_fps_single_workgroup(B, N, M, start):
    from oracle import oracle
    from pointcloudsegmentation_amd import ops
    xyz = _cloud(B, N, seed=N + M)
    want = oracle.fps(xyz, M, start)
    got = ops.farthest_point_sample(torch.from_numpy(xyz).cuda(), M, start).cpu().numpy()
    np.testing.assert_array_equal(got, want)


@pytest.mark.parametrize("B,N,M", [(1, 40000, 400), (2, 40000, 300), (1, 150000, 200), (3, 25000, 100),
                                   (128, 40000, 12),     # two 1024-thread workgroups per cloud, coordinates from global memory
                                   (40, 30000, 40),      # six 256-thread workgroups per cloud (co-residency caps G)
                                   (1, 700001, 64)])     # 64 workgroups x 20 points per thread
def test_fps_multi_workgroup(B, N, M):
    """Clouds too large for one workgroup's registers: cluster of workgroups with a
    per-iteration winner exchange."""
    from oracle import oracle
    from pointcloudsegmentation_amd import ops
    xyz = _cloud(B, N, seed=N + M)
    want = oracle.fps(xyz, M, 0)
    got = ops.farthest_point_sample(torch.from_numpy(xyz).cuda(), M, 0).cpu().numpy()
    assert ops.fps_status("cuda") == 0
    np.testing.assert_array_equal(got, want)


def test_fps_lidar_1m_to_16k_properties():
    """BASELINE configs[4] at full size (1M points -> 16,384 samples).  The oracle finishes a
    1,024-sample prefix in seconds (FPS is prefix-stable: the first m samples of a longer run are
    the m-sample run); the full run is checked through what greedy FPS guarantees: distinct
    indices, non-increasing selection distances, and the LAST pick being the exact arg-max
    (lowest index on ties) of the min-distance field of all earlier picks, recomputed
    independently with the spec's arithmetic ((dx*dx+dy*dy)+dz*dz, separate roundings)."""
    from oracle import oracle
    from pointcloudsegmentation_amd import ops
    N, M = 1_000_000, 16384
    xyz = _cloud(1, N, seed=4, kind="lidar")
    xg = torch.from_numpy(xyz).cuda()
    got = ops.farthest_point_sample(xg, M, 0)
    assert ops.fps_status("cuda") == 0
    g = got.cpu().numpy()[0]
    np.testing.assert_array_equal(g[:1024], oracle.fps(xyz, 1024, 0)[0])
    assert len(np.unique(g)) == M
    pts = xg[0]
    sel = pts[got[0].long()]

    def d2(a, b):   # a (n,3), b (m,3) -> (n,m), elementwise torch ops: each rounded once, no fma
        dx = a[:, None, 0] - b[None, :, 0]
        dy = a[:, None, 1] - b[None, :, 1]
        dz = a[:, None, 2] - b[None, :, 2]
        return (dx * dx + dy * dy) + dz * dz

    mind = torch.full((N,), float("inf"), device="cuda")
    for c in range(0, M - 1, 256):                      # min-distance field of picks 0..M-2
        blk = sel[c:min(c + 256, M - 1)]
        for r in range(0, N, 250_000):
            mind[r:r + 250_000] = torch.minimum(mind[r:r + 250_000], d2(pts[r:r + 250_000], blk).min(dim=1).values)
    best = mind.max()
    assert int(torch.nonzero(mind == best)[0, 0]) == int(g[M - 1])
    # selection distances never increase: distance of pick i to picks < i, on a strided subset
    probe = np.arange(64, M, 509)
    dist = np.array([float(d2(sel[i:i + 1], sel[:i]).min()) for i in probe])
    assert np.all(np.diff(dist) <= 0)


def test_knn_k32_against_1m_reference_points():
    """BASELINE configs[4]: k=32 neighbours of FPS centroids among 1M points; every query row is
    independent, so a 384-row slice of the 16k queries is compared bit-exactly with the oracle."""
    from oracle import oracle
    from pointcloudsegmentation_amd import ops
    N, M = 1_000_000, 16384
    xyz = _cloud(1, N, seed=4, kind="lidar")
    xg = torch.from_numpy(xyz).cuda()
    cent = ops.farthest_point_sample(xg, M, 0)
    q = xg[0][cent[0].long()][None].contiguous()
    idx, d2 = ops.knn(q, xg, 32)
    rows = np.r_[0:128, 8000:8128, M - 128:M]
    wi, wd = oracle.knn(q[:, rows].cpu().numpy(), xyz, 32)
    np.testing.assert_array_equal(idx.cpu().numpy()[:, rows], wi)
    np.testing.assert_array_equal(d2.cpu().numpy()[:, rows], wd)
    assert bool((idx[0, :, 0] == cent[0]).all()) or bool((d2[0, :, 0] == 0).all())   # a centroid finds itself first


def test_fps_ties_duplicates_and_oversampling():
    from oracle import oracle
    from pointcloudsegmentation_amd import ops
    xyz = _grid_cloud(2, 9)                     # 729 lattice points, many exact ties
    want = oracle.fps(xyz, 729, 4)
    got = ops.farthest_point_sample(torch.from_numpy(xyz).cuda(), 729, 4).cpu().numpy()
    np.testing.assert_array_equal(got, want)
    dup = np.concatenate([xyz[:, :100], xyz[:, :100]], axis=1)   # every point twice; M > #distinct
    want = oracle.fps(dup, 150, 0)
    got = ops.farthest_point_sample(torch.from_numpy(dup).cuda(), 150, 0).cpu().numpy()
    np.testing.assert_array_equal(got, want)


def test_fps_edge_cases_default_routing():
    """Exact ties (lattices), duplicates, degenerate extents, ragged member sizes, more samples than distinct points, a NaN
    coordinate, 1..N samples, more clouds than one launch takes: same indices as the oracle (tests/fps_cases.py), with the
    kernels the library picks by itself (wave clusters for 4,096..160,000 points, 256-thread round clusters above)."""
    import fps_cases
    fps_cases.run(small=False)


@pytest.mark.parametrize("mode", ["2", "1", "-1"])
def test_fps_edge_cases_forced_kernel_family(mode):
    """The same cases with one kernel family forced (read once per process, hence a child): 2 = wave clusters on every
    shape that has a plan (down to 5 points), 1 = 256-thread round clusters + the single-workgroup kernels (dense and
    bucket-pruned), -1 = the one-sample-per-exchange cluster kernel."""
    import os
    import subprocess
    import sys
    here = os.path.dirname(os.path.abspath(__file__))
    r = subprocess.run([sys.executable, os.path.join(here, "fps_cases.py")], env=dict(os.environ, P2N_FPS_ROUNDS=mode),
                       capture_output=True, text=True, timeout=600)
    assert r.returncode == 0 and "ALL OK" in r.stdout, r.stdout[-2000:] + r.stderr[-2000:]


def test_fps_errors():
    from pointcloudsegmentation_amd import ops
    x = torch.zeros(1, 10, 3, device="cuda")
    with pytest.raises(ValueError):
        ops.farthest_point_sample(x, 4, start_idx=10)
    with pytest.raises(ValueError):
        ops.farthest_point_sample(torch.zeros(1, 10, 4, device="cuda"), 4)
    assert ops.farthest_point_sample(x, 0).shape == (1, 0)


# ------------------------------------------------------------------ ball query
@pytest.mark.parametrize("B,N,M,r,ns", [(2, 1000, 100, 0.2, 16), (1, 5000, 777, 0.1, 32), (3, 2049, 300, 0.05, 64),
                                        (2, 300, 300, 10.0, 8), (1, 100, 10, 1e-6, 4),
                                        (1, 9000, 1, 0.3, 100), (2, 4097, 33, 0.15, 70), (16, 8192, 2048, 0.1, 32)])
def test_ball_query(B, N, M, r, ns):
    from oracle import oracle
    from pointcloudsegmentation_amd import ops
    xyz = _cloud(B, N, seed=N)
    centres = np.ascontiguousarray(xyz[:, ::max(1, N // M)][:, :M])
    if r < 1e-3:   # nothing in range except (maybe) nothing at all: shift the centres away
        centres = centres + 5.0
    M = centres.shape[1]
    want = oracle.ball_query(xyz, centres, r, ns)
    got = ops.ball_query(torch.from_numpy(xyz).cuda(), torch.from_numpy(centres).cuda(), r, ns).cpu().numpy()
    np.testing.assert_array_equal(got, want)


def test_ball_query_lattice_boundary():
    """Points exactly on the sphere (d2 == r2) are outside (strict <)."""
    from oracle import oracle
    from pointcloudsegmentation_amd import ops
    xyz = _grid_cloud(1, 8)
    want = oracle.ball_query(xyz, xyz[:, :50].copy(), 2.0, 40)
    got = ops.ball_query(torch.from_numpy(xyz).cuda(), torch.from_numpy(xyz[:, :50].copy()).cuda(), 2.0, 40)
    np.testing.assert_array_equal(got.cpu().numpy(), want)


# ------------------------------------------------------------------ kNN / three-NN
@pytest.mark.parametrize("B,M,N,k", [(2, 500, 500, 21), (1, 1000, 4096, 32), (2, 77, 5000, 1), (1, 64, 64, 64),
                                     (3, 100, 9000, 20), (1, 10, 21, 21)])
def test_knn(B, M, N, k):
    from oracle import oracle
    from pointcloudsegmentation_amd import ops
    ref = _cloud(B, N, seed=N + k)
    q = np.ascontiguousarray(ref[:, :M]) if M <= N else _cloud(B, M, seed=1)
    wi, wd = oracle.knn(q, ref, k)
    gi, gd = ops.knn(torch.from_numpy(q).cuda(), torch.from_numpy(ref).cuda(), k)
    np.testing.assert_array_equal(gi.cpu().numpy(), wi)
    np.testing.assert_array_equal(gd.cpu().numpy(), wd)


def test_knn_ties_lattice():
    from oracle import oracle
    from pointcloudsegmentation_amd import ops
    xyz = _grid_cloud(2, 7)
    wi, wd = oracle.knn(xyz, xyz, 27)
    gi, gd = ops.knn(torch.from_numpy(xyz).cuda(), torch.from_numpy(xyz).cuda(), 27)
    np.testing.assert_array_equal(gi.cpu().numpy(), wi)
    np.testing.assert_array_equal(gd.cpu().numpy(), wd)


@pytest.mark.parametrize("B,N,k", [(2, 500, 21), (1, 4096, 32), (3, 8192, 17), (2, 5000, 1), (1, 64, 64), (1, 21, 21),
                                   (2, 9001, 20), (1, 16384, 33), (1, 20000, 17), (4, 130, 64)])
def test_knn_self_equals_streaming_search(B, N, k):
    """The Morton-ordered self search returns what the oracle's exhaustive search returns
    (N=20000 is beyond its LDS sort and takes the streaming kernel)."""
    from oracle import oracle
    from pointcloudsegmentation_amd import ops
    xyz = _cloud(B, N, seed=N + 3 * k)
    wi, wd = oracle.knn(xyz, xyz, k)
    gi, gd = ops.knn_self(torch.from_numpy(xyz).cuda(), k)
    np.testing.assert_array_equal(gi.cpu().numpy(), wi)
    np.testing.assert_array_equal(gd.cpu().numpy(), wd)


def test_knn_of_a_tensor_against_itself_takes_the_self_search():
    """ops.knn(x, x, k) with the very same tensor routes to the self search; a copy of x as the query
    takes the streaming kernel: both must agree bit for bit."""
    from pointcloudsegmentation_amd import ops
    x = torch.from_numpy(_cloud(2, 3000, seed=77)).cuda()
    i0, d0 = ops.knn(x, x, 24)
    i1, d1 = ops.knn(x.clone(), x, 24)
    assert torch.equal(i0, i1) and torch.equal(d0, d1)


def test_knn_self_ties_duplicates_and_clusters():
    from oracle import oracle
    from pointcloudsegmentation_amd import ops
    lattice = _grid_cloud(2, 7)                                  # every distance tied many times over
    dup = np.repeat(_cloud(1, 300, seed=5), 7, axis=1)           # every point 7 times
    rng = np.random.default_rng(11)
    centres = rng.random((1, 40, 3), dtype=np.float32)
    blobs = (np.repeat(centres, 100, axis=1) + 1e-3 * rng.standard_normal((1, 4000, 3)).astype(np.float32))
    flat = _cloud(1, 3000, seed=9)
    flat[..., 2] = 0.25                                          # degenerate extent on one axis
    same = np.full((1, 200, 3), 0.5, np.float32)                 # zero extent on all axes
    for xyz, k in ((lattice, 27), (dup, 20), (blobs, 33), (flat, 17), (same, 9)):
        xyz = np.ascontiguousarray(xyz, dtype=np.float32)
        wi, wd = oracle.knn(xyz, xyz, k)
        gi, gd = ops.knn_self(torch.from_numpy(xyz).cuda(), k)
        np.testing.assert_array_equal(gi.cpu().numpy(), wi)
        np.testing.assert_array_equal(gd.cpu().numpy(), wd)


def test_knn_matches_reference_neighbour_sets():
    """Against the reference's own kNN (losses.py:475-479) on a real-valued cloud:
    the 21-neighbour SETS of the exact-arithmetic cdist agree (golden knn_exact_k21)."""
    from pointcloudsegmentation_amd import ops
    g = golden("loss_b2_n2048_s4.npz")
    pts = torch.from_numpy(np.ascontiguousarray(g["x"][:, :, :3])).cuda()
    gi = ops.knn(pts, pts, 21, return_dist=False).cpu().numpy()
    want = g["knn_exact_k21"]
    same = (np.sort(gi, axis=2) == np.sort(want, axis=2)).all(axis=2)
    assert same.mean() > 0.999, same.mean()       # sqrt in cdist can merge near-ties at the k-th place
    # and against the reference's default (matmul) cdist + topk, self column dropped
    ref_sets = np.sort(g["knn_ref_default"], axis=2)
    mine = np.sort(gi[:, :, 1:], axis=2)
    assert (ref_sets == mine).all(axis=2).mean() > 0.99


def test_knn_errors():
    from pointcloudsegmentation_amd import ops
    x = torch.zeros(1, 10, 3, device="cuda")
    with pytest.raises(ValueError):
        ops.knn(x, x, 11)        # k > N: torch.topk raises in the reference too
    with pytest.raises(ValueError):
        ops.knn(x, x, 65)
    with pytest.raises(ValueError):
        ops.knn_self(x, 11)


@pytest.mark.parametrize("B,n,m", [(2, 1000, 250), (1, 4096, 1024), (2, 10, 3)])
def test_three_nn_and_weights(B, n, m):
    from oracle import oracle
    from pointcloudsegmentation_amd import ops
    unknown = _cloud(B, n, seed=n)
    known = np.ascontiguousarray(unknown[:, ::max(1, n // m)][:, :m])
    wd, wi = oracle.three_nn(unknown, known)
    ww = oracle.three_weights(wd)
    gd, gi, gw = ops.three_nn(torch.from_numpy(unknown).cuda(), torch.from_numpy(known).cuda())
    np.testing.assert_array_equal(gi.cpu().numpy(), wi)
    np.testing.assert_array_equal(gd.cpu().numpy(), wd)
    np.testing.assert_array_equal(gw.cpu().numpy(), ww)


# ------------------------------------------------------------------ gather / interpolate
def test_group_gather_fwd_bwd():
    from oracle import oracle
    from pointcloudsegmentation_amd import ops
    rng = np.random.default_rng(0)
    B, C, N, M, ns = 2, 19, 500, 40, 16
    feat = rng.standard_normal((B, C, N)).astype(np.float32)
    idx = rng.integers(0, N, size=(B, M, ns)).astype(np.int32)
    want = oracle.group_gather(feat, idx)
    ft = torch.from_numpy(feat).cuda().requires_grad_(True)
    out = ops.group_gather(ft, torch.from_numpy(idx).cuda())
    np.testing.assert_array_equal(out.detach().cpu().numpy(), want)
    w = torch.from_numpy(rng.standard_normal(want.shape).astype(np.float32)).cuda()
    (out * w).sum().backward()
    # adjoint computed on the host in float64
    g = np.zeros((B, C, N))
    wn = w.cpu().numpy().astype(np.float64)
    for b in range(B):
        np.add.at(g[b], (slice(None), idx[b].reshape(-1)), wn[b].reshape(C, -1))
    np.testing.assert_allclose(ft.grad.cpu().numpy(), g, rtol=1e-5, atol=1e-5)


def test_three_interpolate_fwd_bwd():
    from oracle import oracle
    from pointcloudsegmentation_amd import ops
    rng = np.random.default_rng(1)
    B, C, m, n = 2, 21, 128, 700
    feat = rng.standard_normal((B, C, m)).astype(np.float32)
    unknown = _cloud(B, n, seed=2)
    known = np.ascontiguousarray(unknown[:, :m])
    d2, idx = oracle.three_nn(unknown, known)
    w = oracle.three_weights(d2)
    want = oracle.three_interpolate(feat, idx, w)
    ft = torch.from_numpy(feat).cuda().requires_grad_(True)
    out = ops.three_interpolate(ft, torch.from_numpy(idx).cuda(), torch.from_numpy(w).cuda())
    np.testing.assert_array_equal(out.detach().cpu().numpy(), want)
    go = torch.from_numpy(rng.standard_normal(want.shape).astype(np.float32)).cuda()
    (out * go).sum().backward()
    g = np.zeros((B, C, m))
    gon = go.cpu().numpy().astype(np.float64)
    for b in range(B):
        for j in range(3):
            np.add.at(g[b], (slice(None), idx[b, :, j]), gon[b] * w[b, :, j].astype(np.float64))
    np.testing.assert_allclose(ft.grad.cpu().numpy(), g, rtol=1e-4, atol=1e-5)


# Adjoints: LDS-accumulating kernels (8 / 4 / 2 / 1 channel rows per workgroup by row length, source range cut
# into slices when there are few (cloud, channel block) pairs) and the global-atomic kernels for rows beyond LDS.
@pytest.mark.parametrize("B,C,N,M,ns", [
    (2, 19, 500, 40, 16),       # 8 rows per workgroup, channel tail
    (16, 64, 8192, 512, 32),    # set-abstraction level 1 shape (M reduced): 4 rows per workgroup
    (1, 3, 8192, 2048, 32),     # one channel block: 64 slices, atomic flush
    (1, 2, 20000, 300, 8),      # one row per workgroup
    (1, 2, 40000, 300, 8),      # row beyond the LDS budget: global-atomic kernel
])
def test_group_gather_bwd_variants(B, C, N, M, ns):
    from pointcloudsegmentation_amd import ops
    rng = np.random.default_rng(N + C)
    idx = rng.integers(0, N, size=(B, M, ns)).astype(np.int32)
    idx[:, :, 0] = idx[:, :1, 0]           # many sources on one target
    ft = torch.zeros(B, C, N, device="cuda", requires_grad=True)
    out = ops.group_gather(ft, torch.from_numpy(idx).cuda())
    w = torch.from_numpy(rng.standard_normal((B, C, M, ns)).astype(np.float32)).cuda()
    (out * w).sum().backward()
    it = torch.from_numpy(idx.reshape(B, 1, -1).astype(np.int64)).cuda().expand(B, C, M * ns)
    want = torch.zeros(B, C, N, dtype=torch.float64, device="cuda").scatter_add_(2, it, w.double().reshape(B, C, -1))
    scale = float(want.abs().max())
    assert float((ft.grad.double() - want).abs().max()) <= 2e-6 * scale


@pytest.mark.parametrize("B,C,m,n", [
    (2, 21, 128, 700),
    (16, 128, 2048, 8192),      # feature-propagation level 1 shape
    (16, 256, 128, 512),        # coarsest level
    (1, 5, 512, 40000),         # one channel block: slices, atomic flush
    (1, 2, 40000, 50000),       # row beyond the LDS budget: global-atomic kernel
])
def test_three_interpolate_bwd_variants(B, C, m, n):
    from pointcloudsegmentation_amd import ops
    rng = np.random.default_rng(m + n)
    idx = rng.integers(0, m, size=(B, n, 3)).astype(np.int32)
    idx[:, ::7, 1] = idx[:, ::7, 0]        # a coarse point twice in one triple
    wt = rng.random((B, n, 3)).astype(np.float32)
    wt /= wt.sum(-1, keepdims=True)
    ft = torch.zeros(B, C, m, device="cuda", requires_grad=True)
    out = ops.three_interpolate(ft, torch.from_numpy(idx).cuda(), torch.from_numpy(wt).cuda())
    go = torch.from_numpy(rng.standard_normal((B, C, n)).astype(np.float32)).cuda()
    (out * go).sum().backward()
    want = torch.zeros(B, C, m, dtype=torch.float64, device="cuda")
    it = torch.from_numpy(idx.astype(np.int64)).cuda()
    wd = torch.from_numpy(wt).cuda()
    for j in range(3):
        contrib = (go * wd[:, None, :, j]).double()          # the kernel forms g*w in float32 as well
        want.scatter_add_(2, it[:, None, :, j].expand(B, C, n), contrib)
    scale = float(want.abs().max())
    assert float((ft.grad.double() - want).abs().max()) <= 2e-6 * scale


# LDS-staged and global-gather variants of the same operators: shapes on both sides of the
# launcher's conditions (row length / column count multiples of four, rows that fit LDS or not,
# channel tails, several column splits) must give identical bits.
@pytest.mark.parametrize("B,C,N,M,ns", [
    (2, 19, 500, 40, 16),      # staged, channel tail
    (2, 5, 501, 40, 16),       # N % 4 != 0: 16-byte global kernel
    (1, 3, 64, 7, 3),          # M*ns % 4 != 0: scalar kernel
    (3, 64, 8192, 256, 32),    # staged, two workgroups per CU, column splits
    (1, 2, 40960, 128, 8),     # one 160 KiB row per workgroup
    (1, 2, 50000, 128, 8),     # row does not fit LDS: global kernel
    (2, 130, 2048, 64, 32),    # many channel chunks
])
def test_group_gather_variants(B, C, N, M, ns):
    from oracle import oracle
    from pointcloudsegmentation_amd import ops
    rng = np.random.default_rng(C * N + M)
    feat = rng.standard_normal((B, C, N)).astype(np.float32)
    idx = rng.integers(0, N, size=(B, M, ns)).astype(np.int32)
    idx[:, 0, 0] = 0
    idx[:, -1, -1] = N - 1
    want = oracle.group_gather(feat, idx)
    out = ops.group_gather(torch.from_numpy(feat).cuda(), torch.from_numpy(idx).cuda())
    np.testing.assert_array_equal(out.cpu().numpy(), want)


@pytest.mark.parametrize("B,C,m,n", [
    (2, 21, 128, 700),         # staged, channel tail
    (2, 7, 127, 700),          # m % 4 != 0
    (1, 4, 64, 33),            # n % 4 != 0
    (3, 256, 512, 2048),       # staged (feature-propagation level 1)
    (2, 128, 2048, 8192),      # staged (feature-propagation level 2)
    (1, 3, 50000, 4096),       # row does not fit LDS
])
def test_three_interpolate_variants(B, C, m, n):
    from oracle import oracle
    from pointcloudsegmentation_amd import ops
    rng = np.random.default_rng(C + m + n)
    feat = rng.standard_normal((B, C, m)).astype(np.float32)
    idx = rng.integers(0, m, size=(B, n, 3)).astype(np.int32)
    w = rng.random((B, n, 3)).astype(np.float32)
    w /= w.sum(-1, keepdims=True)
    want = oracle.three_interpolate(feat, idx, w)
    out = ops.three_interpolate(torch.from_numpy(feat).cuda(), torch.from_numpy(idx).cuda(), torch.from_numpy(w).cuda())
    np.testing.assert_array_equal(out.cpu().numpy(), want)


@pytest.mark.parametrize("B,C,N,M,ns", [(2, 6, 512, 32, 16), (2, 6, 510, 32, 16), (1, 0, 256, 16, 8), (2, 67, 2048, 128, 32)])
def test_sa_group_variants(B, C, N, M, ns):
    rng = np.random.default_rng(N + C)
    xyz = rng.standard_normal((B, N, 3)).astype(np.float32)
    cen = rng.standard_normal((B, M, 3)).astype(np.float32)
    feat = rng.standard_normal((B, C, N)).astype(np.float32) if C else None
    idx = rng.integers(0, N, size=(B, M, ns)).astype(np.int32)
    from pointcloudsegmentation_amd import _lib, runtime as _rt
    tx, tc, ti = torch.from_numpy(xyz).cuda(), torch.from_numpy(cen).cuda(), torch.from_numpy(idx).cuda()
    tf = None if feat is None else torch.from_numpy(feat).cuda()
    out = torch.full((B, 3 + C, M, ns), float("nan"), device="cuda")
    _lib.check(_lib.lib().p2n_sa_group(tx.data_ptr(), tc.data_ptr(), _rt.ptr(tf), ti.data_ptr(), B, C, N, M, ns,
                                       out.data_ptr(), _rt.stream_handle(tx.device)), "p2n_sa_group")
    want = np.empty((B, 3 + C, M, ns), np.float32)
    for b in range(B):
        want[b, :3] = (xyz[b][idx[b]] - cen[b][:, None, :]).transpose(2, 0, 1)
        if C:
            want[b, 3:] = feat[b][:, idx[b].reshape(-1)].reshape(C, M, ns)
    np.testing.assert_array_equal(out.cpu().numpy(), want)


def test_fps_exchange_timeout_is_loud_and_indices_stay_in_range():
    """A cluster FPS needs all members of a cloud resident at once.  When an exchange
    times out (forced here: P2N_FPS_FAULT makes one member skip its publish in the round that would find sample 6,
    with a small spin limit), the call must (1) leave only valid indices behind, (2) keep the correct prefix,
    (3) raise FpsExchangeTimeout from fps_check(), and (4) keep its status out of the shared workspace:
    another workspace user in between must not clear it.  Runs in a child process because the tuning
    overrides are read once per process."""
    import os
    import subprocess
    import sys
    child = r'''
import numpy as np, torch, sys
sys.path.insert(0, %r)
from pointcloudsegmentation_amd import ops
from pointcloudsegmentation_amd.synth import make_batch
from oracle import oracle
xyz = np.ascontiguousarray(make_batch(2, 40000, 3)[:, :, :3])
xg = torch.from_numpy(xyz).cuda()
assert ops.fps_needs_check(2, 40000)
idx = ops.farthest_point_sample(xg, 64, 7)
ops.knn_self(xg[:, :4096].contiguous(), 8)            # another user of the shared per-stream workspace
got = idx.cpu().numpy()
assert got.min() >= 0 and got.max() < 40000, "indices out of range after a timeout"
want = oracle.fps(xyz, 6, 7)
assert (got[:, :6] == want).all(), "prefix before the timeout must be the real sample"
assert (got[:, 6:] == 7).all(), "rest of the row repeats the start index"
assert ops.fps_status("cuda") == 1
try:
    ops.fps_check("cuda")
except ops.FpsExchangeTimeout:
    print("RAISED")
assert ops.fps_status("cuda") == 0                    # the check resets the word
assert not ops.fps_needs_check(2, 2048)
ok = ops.farthest_point_sample(xg[:, :2048].contiguous(), 32, 0)   # single workgroup per cloud: cannot time out
ops.fps_check("cuda")
assert (ok.cpu().numpy() == oracle.fps(xyz[:, :2048], 32, 0)).all()
print("DONE")
''' % os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    env = dict(os.environ, P2N_FPS_FAULT="5", P2N_FPS_SPIN="2000")
    r = subprocess.run([sys.executable, "-c", child], env=env, capture_output=True, text=True, timeout=300)
    assert r.returncode == 0, r.stdout + r.stderr
    assert "RAISED" in r.stdout and "DONE" in r.stdout, r.stdout + r.stderr
