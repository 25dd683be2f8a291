"""One-off sanity: 1,000,000 x 64 embeddings (a LiDAR-sized scene) through the clustering, default path against
both-sides visits."""
import os, sys, time
import numpy as np, torch
REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, REPO)
from pointcloudsegmentation_amd.clustering import cluster_embeddings
from pointcloudsegmentation_amd.synth import make_embeddings
n = 1_000_000
e = torch.from_numpy(make_embeddings(n, 64, 9, n_clusters=150)).cuda()
res = {}
for name, env in (("default", {}), ("both sides", {"P2N_CL_SYM": "0"})):
    os.environ.pop("P2N_CL_SYM", None)
    os.environ.update(env)
    lab, cnt = cluster_embeddings(e, return_counts=True)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    lab, cnt = cluster_embeddings(e, return_counts=True)
    torch.cuda.synchronize()
    res[name] = (lab, cnt, 1e3 * (time.perf_counter() - t0))
    print(name, f"{res[name][2]:.2f} ms, clusters {int(lab.max())}, noise {int((lab == 0).sum())}", flush=True)
print("labels equal", bool(torch.equal(res["default"][0], res["both sides"][0])), "counts equal", bool(torch.equal(res["default"][1], res["both sides"][1])))
