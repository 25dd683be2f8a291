"""Experiment: after the code order, rows far from their own 128-row block's centre (relative to the block's median
distance) are moved to the end -- outliers then share blocks with each other instead of widening instance blocks.
Library in the given order (P2N_CL_SORT=0), orders applied here."""
import os, sys, time
os.environ["P2N_CL_SORT"] = "0"
import numpy as np, torch
REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, REPO); sys.path.insert(0, os.path.join(REPO, "tools"))
from pointcloudsegmentation_amd.clustering import cluster_embeddings, normalize_rows
from pointcloudsegmentation_amd.synth import make_embeddings
from pointcloudsegmentation_amd.inference import preprocess_point_cloud
from src.models.point2net import Point2Net
from exp_cluster_centred import order, timed

n = 166499
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
    p = order(e, True)
    es = e[p].contiguous()
    t0, _ = timed(lambda: cluster_embeddings(es))
    x = normalize_rows(es)
    nb = n // 128
    xb = x[:nb * 128].view(nb, 128, 64)
    d = (xb - xb.mean(1, keepdim=True)).norm(dim=2)
    med = d.median(1, keepdim=True).values
    for f in (2.0, 3.0, 4.0):
        far = torch.zeros(n, dtype=torch.bool, device=e.device)
        far[:nb * 128] = (d > f * med + 0.02).reshape(-1)
        q = torch.cat([torch.nonzero(~far).squeeze(1), torch.nonzero(far).squeeze(1)])
        e2 = es[q].contiguous()
        t1, _ = timed(lambda: cluster_embeddings(e2))
        print(f"{name:18s} code order {t0:5.2f} ms   stragglers (> {f} x median + 0.02, {100 * float(far.float().mean()):4.1f} % of rows) last {t1:5.2f} ms", flush=True)
