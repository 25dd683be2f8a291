#!/usr/bin/env python3
"""BASELINE configs[0] shape: one 166,499-point scene through normalise -> Point2Net -> cluster
(reference inference.py:143-188), everything on one MI355X; the reference's CPU path (stock
PyTorch forward; scikit-learn DBSCAN on a subsample) timed beside it.  Prints one JSON line."""
import argparse, json, os, sys, time
REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, REPO)
import numpy as np
import torch
from pointcloudsegmentation_amd.clustering import cluster_embeddings
from pointcloudsegmentation_amd.inference import predict
from pointcloudsegmentation_amd.point2net import Point2Net
from pointcloudsegmentation_amd.preprocess import preprocess_point_cloud
from pointcloudsegmentation_amd.synth import make_embeddings

ap = argparse.ArgumentParser()
ap.add_argument("--n", type=int, default=166499)
ap.add_argument("--steps", type=int, default=5)
ap.add_argument("--no-cpu", action="store_true")
a = ap.parse_args()
rng = np.random.default_rng(0)
# coordinates as a scan file holds them: float32 (PLY `property float x`), read back as float64 the way open3d hands
# them to the reference (inference.py:76-80) -- the normalisation's row sum is fast on such data (csrc/p2n_prep.hip)
pts = (rng.uniform(0, 1, (a.n, 3)) * np.array([8.0, 6.0, 3.0])).astype(np.float32).astype(np.float64)
pts_full = rng.uniform(0, 1, (a.n, 3)) * np.array([8.0, 6.0, 3.0])     # full 53-bit mantissas: numpy's order everywhere
col = rng.integers(0, 256, (a.n, 3)).astype(np.float64)
w = dict(np.load(os.path.join(REPO, "tests", "golden", "weights.npz")))
m = Point2Net(feature_dim=6, embedding_dim=64)
m.load_state_dict({k: torch.from_numpy(np.asarray(v)) for k, v in w.items()}, strict=True)
m = m.cuda().eval()


def timed(fn, steps):
    fn(); torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(steps):
        r = fn()
    torch.cuda.synchronize()
    return 1e3 * (time.perf_counter() - t0) / steps, r


out = {"metric": "scene inference ms (normalise + forward + cluster)", "n": a.n}
out["end_to_end_ms_host_in_host_out"], res = timed(lambda: predict(m, pts, col), a.steps)
out["num_instances"] = res["num_instances"]
pg, cg = torch.from_numpy(pts).cuda(), torch.from_numpy(col).cuda()
out["end_to_end_ms_device_resident"], _ = timed(lambda: predict(m, pg, cg, to_host=False), a.steps)
out["normalise_ms"], (feat, _, _) = timed(lambda: preprocess_point_cloud(pg, cg), a.steps)
pf = torch.from_numpy(pts_full).cuda()
out["normalise_ms_full_mantissa_float64"], _ = timed(lambda: preprocess_point_cloud(pf, cg), a.steps)
out["input"] = "float32-origin coordinates as float64 (scan files); normalise_ms_full_mantissa_float64 = random 53-bit mantissas"
with torch.no_grad():
    out["forward_ms"], emb = timed(lambda: m(feat), a.steps)
out["cluster_ms_model_embeddings"], _ = timed(lambda: cluster_embeddings(emb[0]), a.steps)
syn = torch.from_numpy(make_embeddings(a.n, 64, 11)).cuda()
out["cluster_ms_33_cluster_embeddings"], _ = timed(lambda: cluster_embeddings(syn), a.steps)
# stream of scenes (BASELINE configs[2]: 40k-150k points): ms per scene one after the other vs pipelined
# (upload of scene i+1 and download of scene i-1 beside the compute of scene i)
from pointcloudsegmentation_amd.inference import predict_many
out["scene_stream"] = {}
NSC = 20     # scenes per call: a handful would measure the call's prologue, not the pipeline
for n_s in (40000, 80000, 150000):
    sc = []
    for k in range(NSC):
        r2 = np.random.default_rng(100 + k)
        sc.append(((r2.uniform(0, 1, (n_s, 3)) * np.array([8.0, 6.0, 3.0])).astype(np.float32).astype(np.float64), r2.integers(0, 256, (n_s, 3)).astype(np.float64)))
    row = {}
    for mode, ov in (("serial", False), ("pipelined", True)):
        for _ in range(2):      # the caching allocators (device blocks per stream, pinned blocks) reach their steady state
            list(predict_many(m, sc[:4], overlap=ov))
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        stamps = []
        for _res in predict_many(m, sc, overlap=ov):
            stamps.append(time.perf_counter())
        torch.cuda.synchronize()
        t1 = time.perf_counter()
        gaps = np.diff(np.array([t0] + stamps)) * 1e3
        row[mode + "_ms_per_scene"] = 1e3 * (t1 - t0) / len(sc)
        row[mode + "_first_result_ms"] = float(gaps[0])          # prologue + the pipeline's fill
        row[mode + "_ms_per_scene_min"] = float(gaps[1:].min())
        row[mode + "_ms_per_scene_median"] = float(np.median(gaps[1:]))
    row["scenes_per_call"] = NSC
    row["speedup"] = row["serial_ms_per_scene"] / row["pipelined_ms_per_scene"]
    out["scene_stream"][str(n_s)] = row
if not a.no_cpu:
    mc = Point2Net(feature_dim=6, embedding_dim=64)
    mc.load_state_dict(m.state_dict())
    mc.eval()
    fc = feat.cpu()
    with torch.no_grad():
        mc(fc)
        t0 = time.perf_counter(); ec = mc(fc); out["cpu_torch_forward_s"] = time.perf_counter() - t0
    from sklearn.cluster import DBSCAN
    sub = syn[:30000].cpu().numpy()
    x = sub / (np.linalg.norm(sub, axis=1, keepdims=True) + 1e-8)
    t0 = time.perf_counter(); DBSCAN(eps=0.1, min_samples=50, n_jobs=-1).fit_predict(x)
    out["cpu_sklearn_dbscan_s_30000_subsample"] = time.perf_counter() - t0
    out["cpu_threads"] = torch.get_num_threads()
print(json.dumps(out))
