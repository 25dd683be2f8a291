"""FPS edge cases shared by the in-process test (default routing) and the child processes that force one kernel family
(P2N_FPS_ROUNDS is read once per process: 2 = wave clusters wherever a plan exists, 1 = 256-thread round clusters /
single-workgroup kernels, -1 = one sample per exchange / bucket-pruned kernel).  Every case: indices equal the oracle's."""
import os
import sys

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.dirname(HERE))


def _cloud(B, N, seed, kind="scannet"):
    from pointcloudsegmentation_amd.synth import make_batch
    return np.ascontiguousarray(make_batch(B, N, seed, kind)[:, :, :3])


def _grid_cloud(B, side):
    g = np.stack(np.meshgrid(np.arange(side), np.arange(side), np.arange(side), indexing="ij"), -1).reshape(-1, 3)
    return np.stack([np.roll(g, b * 7, axis=0) for b in range(B)]).astype(np.float32)


def cases(small: bool):
    """(name, xyz, M, start).  small=True adds shapes below 4,096 points (only meaningful when a mode forces clusters
    onto them)."""
    rng = np.random.default_rng(11)
    out = []
    out.append(("lattice", _grid_cloud(2, 18), 300, 5))                      # 5,832 lattice points: masses of exact ties, across members too
    base = _cloud(1, 1000, seed=3)
    dup = np.concatenate([base] * 7, axis=1)
    out.append(("duplicates", dup[:, rng.permutation(dup.shape[1])], 1200, 5))   # beyond the 1,000 distinct points: all-zero distances, lowest index
    c = rng.standard_normal((1, 12, 3)).astype(np.float32) * 3
    out.append(("blobs", (c[:, rng.integers(0, 12, 9000)] + 0.01 * rng.standard_normal((1, 9000, 3))).astype(np.float32), 300, 5))
    flat = _cloud(2, 6000, seed=8)
    flat[:, :, 2] = 0.25
    flat[1, :, 1] = -1.0
    out.append(("flat", flat, 300, 5))
    out.append(("ragged", _cloud(3, 12288 - 37, seed=9), 500, 12250))
    out.append(("oversample", _cloud(1, 5120, seed=10), 5120, 5))            # every point exactly once
    nan = _cloud(1, 7000, seed=12)
    nan[0, 4321, 1] = np.nan
    out.append(("nan", nan, 64, 5))
    out.append(("one_sample", _cloud(2, 6000, seed=13), 1, 77))
    out.append(("two_samples", _cloud(2, 6000, seed=14), 2, 0))
    out.append(("40k", _cloud(2, 40000, seed=15), 300, 0))
    out.append(("25k_x3", _cloud(3, 25000, seed=16), 100, 9))
    out.append(("30k_x40", _cloud(40, 30000, seed=17), 40, 0))               # members per cloud capped by the other clouds
    out.append(("4096_x70", _cloud(70, 4096, seed=18), 64, 1))               # more clouds than one launch takes
    out.append(("150k", _cloud(1, 150000, seed=19), 200, 0))
    out.append(("300k", _cloud(1, 300000, seed=24), 300, 7))                 # more than 64 members: several records per decision lane
    out.append(("odd_split", _cloud(1, 4099, seed=20), 4099, 4098))          # ragged last member, all points sampled
    if small:
        out.append(("n5", _cloud(2, 5, seed=21), 5, 3))
        out.append(("n1000", _cloud(3, 1000, seed=22), 64, 0))
        out.append(("n130", _cloud(1, 130, seed=23), 130, 129))
        out.append(("lattice729", _grid_cloud(2, 9), 729, 4))
    return out


def run(small: bool):
    import torch
    from oracle import oracle
    from pointcloudsegmentation_amd import ops
    for name, xyz, M, start in cases(small):
        xyz = np.ascontiguousarray(xyz)
        want = oracle.fps(xyz, M, start)
        got = ops.farthest_point_sample(torch.from_numpy(xyz).cuda(), M, start).cpu().numpy()
        assert ops.fps_status("cuda") == 0, name
        bad = np.argwhere(got != want)
        assert bad.size == 0, f"{name}: first difference at (cloud, sample) {tuple(bad[0])}: got {got[tuple(bad[0])]}, want {want[tuple(bad[0])]}"
        print("ok", name, flush=True)


if __name__ == "__main__":
    run(small=os.environ.get("P2N_FPS_ROUNDS") == "2")
    print("ALL OK")
