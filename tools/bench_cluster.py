#!/usr/bin/env python3
"""Embedding clustering timing: p2n_cluster_embeddings on a
reference-scene-sized embedding matrix, with scikit-learn (the reference's implementation,
inference.py:127-128) timed beside it on a subsample.  Prints one JSON line."""
import argparse, ctypes, json, os, sys, time
REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, REPO)
import numpy as np
import torch
from pointcloudsegmentation_amd import _lib
from pointcloudsegmentation_amd.clustering import cluster_embeddings, last_cluster_stats
from pointcloudsegmentation_amd.synth import make_embeddings

ap = argparse.ArgumentParser()
ap.add_argument("--n", type=int, default=166499)
ap.add_argument("--d", type=int, default=64)
ap.add_argument("--clusters", type=int, default=33)
ap.add_argument("--steps", type=int, default=5)
ap.add_argument("--order", default="runs", choices=["runs", "random"], help="row order of the synthetic embeddings")
ap.add_argument("--cpu-n", type=int, default=30000, help="subsample size for the scikit-learn leg (0 = skip)")
a = ap.parse_args()
e_np = make_embeddings(a.n, a.d, 11, n_clusters=a.clusters, order=a.order)
e = torch.from_numpy(e_np).cuda()
lab = cluster_embeddings(e)
torch.cuda.synchronize()
L = _lib.lib()
L.p2n_profile_enable(1)
t0 = time.perf_counter()
for _ in range(a.steps):
    lab = cluster_embeddings(e)
torch.cuda.synchronize()
ms = 1e3 * (time.perf_counter() - t0) / a.steps
msum = (ctypes.c_float * 32)(); cnt = (ctypes.c_int * 32)()
L.p2n_profile_collect(32, msum, cnt)
L.p2n_profile_enable(0)
k = {n: (msum[i] / max(cnt[i], 1)) * (cnt[i] // a.steps) for n, i in (("prep", 10), ("count", 11), ("link", 12))}
flop_pass = 2.0 * a.n * a.n * (8 * ((a.d + 7) // 8))
out = {"metric": "embedding clustering ms per scene", "n": a.n, "d": a.d, "order": a.order, "ms": ms, "kernels_ms": k,
       "count_pass_tflops": flop_pass / (k["count"] * 1e-3) / 1e12, "link_pass_tflops": flop_pass / (k["link"] * 1e-3) / 1e12,
       "clusters": int(lab.max()), "noise": int((lab == 0).sum()), "stats": last_cluster_stats(a.n, a.d, 50)}
if a.cpu_n:
    from sklearn.cluster import DBSCAN
    sub = e_np[:a.cpu_n]
    x = sub / (np.linalg.norm(sub, axis=1, keepdims=True) + 1e-8)
    t0 = time.perf_counter()
    DBSCAN(eps=0.1, min_samples=50, n_jobs=-1).fit_predict(x)
    t_cpu = time.perf_counter() - t0
    t0 = time.perf_counter()
    cluster_embeddings(torch.from_numpy(sub).cuda()); torch.cuda.synchronize()
    out["sklearn"] = {"n": a.cpu_n, "seconds": t_cpu, "gpu_ms_same_n": 1e3 * (time.perf_counter() - t0)}
print(json.dumps(out))
