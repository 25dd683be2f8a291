"""Experiment: how many of the block pairs a count pass must visit are 'full' (every pair within eps by the
spheres alone: |cA - cB| + rA + rB <= eps) and how many 'interacting' at all, in code order."""
import os, sys
import numpy as np, torch
REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, REPO); sys.path.insert(0, os.path.join(REPO, "tools"))
from pointcloudsegmentation_amd.clustering import normalize_rows
from pointcloudsegmentation_amd.synth import make_embeddings
from pointcloudsegmentation_amd.inference import preprocess_point_cloud
from src.models.point2net import Point2Net
from exp_cluster_sorted import code_keys

n, eps = 166499, 0.1
cases = {"33 clusters": torch.from_numpy(make_embeddings(n, 64, 11)).cuda()}
rng = np.random.default_rng(0)
pts = (rng.uniform(0, 1, (n, 3)) * np.array([8.0, 6.0, 3.0])).astype(np.float32).astype(np.float64)
col = rng.integers(0, 256, (n, 3)).astype(np.float64)
w = dict(np.load(os.path.join(REPO, "tests", "golden", "weights.npz")))
m_ = Point2Net(feature_dim=6, embedding_dim=64)
m_.load_state_dict({k: torch.from_numpy(np.asarray(v)) for k, v in w.items()}, strict=True)
m_ = m_.cuda().eval()
feat, _, _ = preprocess_point_cloud(torch.from_numpy(pts).cuda(), torch.from_numpy(col).cuda())
with torch.no_grad():
    cases["model embeddings"] = m_(feat)[0].contiguous()
for name, e in cases.items():
    x = normalize_rows(e)
    p = torch.sort(code_keys(e), stable=True).indices
    x = x[p]
    nb = n // 128
    xb = x[:nb * 128].view(nb, 128, 64)
    c = xb.mean(1)
    r = (xb - c[:, None, :]).norm(dim=2).max(1).values
    d = torch.cdist(c, c)
    sphere = d <= r[:, None] + r[None, :] + eps
    full = d + r[:, None] + r[None, :] <= eps
    print(f"{name:18s} blocks {nb}  radius median {float(r.median()):.3f}  sphere-listed {float(sphere.float().mean()):.3f}  full {float(full.float().mean()):.4f}"
          f"  full/sphere {float(full.sum()) / float(sphere.sum()):.3f}", flush=True)
