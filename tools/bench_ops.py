#!/usr/bin/env python3
"""Point-set operator timings against their rooflines (BASELINE north_star: 'achieved HBM GB/s for
FPS, grouping and interpolation').  Shapes: the set-abstraction / feature-propagation levels of
the hierarchical sibling at BASELINE configs[1] (B=16, N=8192) and configs[4] (1M-point LiDAR).
Algorithmic bytes per op are spelled out next to each entry.  Prints one JSON line."""
import argparse, json, os, sys
REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, REPO)
import numpy as np
import torch
from pointcloudsegmentation_amd import ops
from pointcloudsegmentation_amd.synth import make_batch

HBM_PEAK_GBS = 8000.0
ap = argparse.ArgumentParser()
ap.add_argument("--iters", type=int, default=20)
ap.add_argument("--no-lidar", action="store_true")
ap.add_argument("--only-gather", action="store_true", help="time only group gather and three-interpolate (geometry sweeps)")
a = ap.parse_args()


import ctypes
from pointcloudsegmentation_amd import _lib
L = _lib.lib()


def timed(fn, iters):
    """Mean GPU time of the library's kernels inside fn (HIP events the library records around its
    launches on the launch stream), so Python / allocator time between launches is not counted."""
    fn(); torch.cuda.synchronize()
    L.p2n_profile_enable(1)
    for _ in range(iters):
        r = fn()
    ms = (ctypes.c_float * 32)(); cnt = (ctypes.c_int * 32)()
    L.p2n_profile_collect(32, ms, cnt)
    L.p2n_profile_enable(0)
    return sum(ms) / iters, r


def fill_ms(nbytes, iters=20):
    """A plain device fill of the same number of bytes: what the memory system gives a pure
    streaming write of this size (small launches never reach the 8 TB/s figure)."""
    buf = torch.empty(int(nbytes) // 4, device="cuda")
    buf.fill_(0.0); torch.cuda.synchronize()
    e0 = torch.cuda.Event(enable_timing=True); e1 = torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(iters):
        buf.fill_(0.0)
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / iters


def entry(ms, nbytes, note):
    gbs = nbytes / (ms * 1e-3) / 1e9
    return {"ms": ms, "algorithmic_MB": nbytes / 1e6, "GB_per_s": gbs, "frac_of_hbm_peak": gbs / HBM_PEAK_GBS, "bytes": note}


out = {"metric": "point-set operators vs HBM roofline", "hbm_peak_GB_per_s": HBM_PEAK_GBS, "levels": {}}
B, N = 16, 8192
xyz = torch.from_numpy(np.ascontiguousarray(make_batch(B, N, 0)[:, :, :3])).cuda()
for (name, M, ns, C, radius) in [("sa1", 2048, 32, 64, 0.1), ("sa2", 512, 32, 128, 0.2)]:
    src = xyz if name == "sa1" else cent
    n_src = src.size(1)
    lvl = {}
    ms, idx = timed(lambda: ops.farthest_point_sample(src, M), 1 if a.only_gather else a.iters)
    lvl["fps"] = entry(ms, B * (12 * n_src + 4 * M), "12*N xyz read + 4*M indices (state lives in registers; latency-bound, see us_per_sample)")
    lvl["fps"]["us_per_sample"] = 1e3 * ms / M
    cent_new = torch.gather(src, 1, idx.long()[..., None].expand(-1, -1, 3)).contiguous()
    ms, gidx = timed(lambda: ops.ball_query(src, cent_new, radius, ns), a.iters)
    lvl["ball_query"] = entry(ms, B * (12 * n_src + 12 * M + 4 * M * ns), "xyz + centroids read, 4*M*ns indices written (VALU-bound: M*N distance tests)")
    feat = torch.randn(B, C, n_src, device="cuda")
    ms, g = timed(lambda: ops.group_gather(feat, gidx), a.iters)
    lvl["group_gather"] = entry(ms, B * (4 * C * M * ns + 4 * M * ns + 4 * C * n_src), "4*C*M*ns written + 4*M*ns indices + 4*C*N features read once")
    lvl["group_gather"]["fill_of_same_bytes_ms"] = fill_ms(B * (4 * C * M * ns + 4 * M * ns + 4 * C * n_src))
    ms, k = timed(lambda: ops.knn(cent_new, src, 32, return_dist=False), 1 if a.only_gather else a.iters)
    lvl["knn_k32"] = entry(ms, B * (12 * n_src + 12 * M + 4 * M * 32), "xyz + queries read, 4*M*k indices (VALU-bound: M*N distance tests)")
    out["levels"][name] = lvl
    cent = cent_new
# feature propagation: 512 -> 2048 -> 8192
for (name, n, m, C) in [("fp_512_to_2048", 2048, 512, 256), ("fp_2048_to_8192", 8192, 2048, 128)]:
    unk = xyz[:, :n].contiguous(); kn = xyz[:, :m].contiguous()
    lvl = {}
    ms, (d2, idx3, w3) = timed(lambda: ops.three_nn(unk, kn), a.iters)
    lvl["three_nn"] = entry(ms, B * (12 * n + 12 * m + 36 * n), "coordinates read, dist2/idx/weight (36 B per unknown) written (VALU-bound)")
    feat = torch.randn(B, C, m, device="cuda")
    ms, o = timed(lambda: ops.three_interpolate(feat, idx3, w3), a.iters)
    lvl["three_interpolate"] = entry(ms, B * (4 * C * n + 24 * n + 4 * C * m), "4*C*n written + 24*n idx/weight + 4*C*m features read once")
    lvl["three_interpolate"]["fill_of_same_bytes_ms"] = fill_ms(B * (4 * C * n + 24 * n + 4 * C * m))
    out["levels"][name] = lvl
if not a.no_lidar:
    big = torch.from_numpy(np.ascontiguousarray(make_batch(1, 1_000_000, 4, "lidar")[:, :, :3])).cuda()
    lvl = {}
    ms, idx = timed(lambda: ops.farthest_point_sample(big, 16384), 2)
    lvl["fps_1m_to_16k"] = entry(ms, 12 * 1_000_000 + 4 * 16384, "12*N + 4*M (register-resident state across a cluster of workgroups)")
    lvl["fps_1m_to_16k"]["us_per_sample"] = 1e3 * ms / 16384
    q = torch.gather(big, 1, idx.long()[..., None].expand(-1, -1, 3)).contiguous()
    ms, k = timed(lambda: ops.knn(q, big, 32, return_dist=False), 3)
    lvl["knn_k32_16k_queries"] = entry(ms, 12 * 1_000_000 + 12 * 16384 + 4 * 16384 * 32, "VALU-bound: 1.6e10 distance tests")
    lvl["knn_k32_16k_queries"]["G_distance_tests_per_s"] = 16384 * 1e6 / (ms * 1e-3) / 1e9
    out["levels"]["lidar_1m"] = lvl
if not a.only_gather:
    # the smoothness loss's neighbour search at the training shape: streaming kNN beside the Morton-ordered self search
    g = torch.Generator(device="cuda").manual_seed(3)
    cl = torch.rand(32, 8192, 3, device="cuda", generator=g)
    lvl = {}
    nbytes = 32 * (12 * 8192 + 4 * 8192 * 17)
    clq = cl.clone()   # a separate query tensor: ops.knn(x, x) with one and the same tensor would take the self search
    ms, i0 = timed(lambda: ops.knn(clq, cl, 17, return_dist=False), 10)
    lvl["knn_streaming_k17"] = entry(ms, nbytes, "xyz read, 4*N*k indices written (VALU-bound: N*N distance tests)")
    ms, i1 = timed(lambda: ops.knn_self(cl, 17, return_dist=False), 10)
    lvl["knn_self_k17"] = entry(ms, nbytes, "same result through the Morton-ordered copy (sort + pruned search)")
    lvl["knn_self_k17"]["identical_to_streaming"] = bool((i0 == i1).all())
    out["levels"]["loss_32x8192"] = lvl
if a.only_gather:
    print(json.dumps({lv: {k: round(v["GB_per_s"]) for k, v in d.items() if k in ("group_gather", "three_interpolate")}
                      for lv, d in out["levels"].items()}))
else:
    print(json.dumps(out))
