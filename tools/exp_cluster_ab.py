#!/usr/bin/env python3
"""Clustering A/B on one box: the library given as argv[1] (default: the in-tree one), model embeddings of the scene
bench + the two synthetic orders.  usage: exp_cluster_ab.py [path/to/libp2n_hip.so]"""
import os, sys, time, json
REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, REPO)
import numpy as np, torch
from pointcloudsegmentation_amd import _lib
if len(sys.argv) > 1:
    _lib.LIB_PATH = os.path.abspath(sys.argv[1])
from pointcloudsegmentation_amd.clustering import cluster_embeddings
from pointcloudsegmentation_amd.synth import make_embeddings
from pointcloudsegmentation_amd.preprocess import preprocess_point_cloud
from src.models.point2net import Point2Net

def timed(fn, steps=5):
    fn(); torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(steps): r = fn()
    torch.cuda.synchronize()
    return 1e3 * (time.perf_counter() - t0) / steps, r

rng = np.random.default_rng(0)
n = 166499
pts = (rng.uniform(0, 1, (n, 3)) * np.array([8.0, 6.0, 3.0])).astype(np.float32).astype(np.float64)
col = rng.integers(0, 256, (n, 3)).astype(np.float64)
w = dict(np.load(os.path.join(REPO, "tests", "golden", "weights.npz")))
m = Point2Net(feature_dim=6, embedding_dim=64)
m.load_state_dict({k: torch.from_numpy(np.asarray(v)) for k, v in w.items()}, strict=True)
m = m.cuda().eval()
feat, _, _ = preprocess_point_cloud(torch.from_numpy(pts).cuda(), torch.from_numpy(col).cuda())
with torch.no_grad():
    emb = m(feat)
# the same scene with its points in a surface-following order (sorted along a space-filling-ish key), as a scanned mesh has them
key = (np.floor(pts[:, 0] * 4).astype(np.int64) * 64 + np.floor(pts[:, 1] * 4).astype(np.int64)) * 64 + np.floor(pts[:, 2] * 4).astype(np.int64)
perm = torch.from_numpy(np.argsort(key, kind="stable")).cuda()
emb_sorted = emb[:, perm].contiguous()
out = {"lib": os.path.basename(_lib.LIB_PATH)}
out["model_embeddings_sorted_ms"], lab2 = timed(lambda: cluster_embeddings(emb_sorted[0]))
out["model_embeddings_ms"], lab = timed(lambda: cluster_embeddings(emb[0]))
out["model_labels_checksum"] = int(np.asarray(lab.cpu() if hasattr(lab, 'cpu') else lab).astype(np.int64).dot(np.arange(len(lab)) % 1000003) % 1000000007)
for order in ("runs", "random"):
    e = torch.from_numpy(make_embeddings(166499, 64, 11, order=order)).cuda()
    out[order + "_ms"], lab = timed(lambda: cluster_embeddings(e))
    out[order + "_checksum"] = int(np.asarray(lab.cpu() if hasattr(lab, 'cpu') else lab).astype(np.int64).dot(np.arange(len(lab)) % 1000003) % 1000000007)
print(json.dumps(out))
