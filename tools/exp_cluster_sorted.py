"""Experiment: working orders for the clustering passes, applied outside the library (P2N_CL_SORT=0 inside):
code order vs code order with rows of rare codes (runs shorter than m) moved to the end."""
import os, sys, time
os.environ["P2N_CL_SORT"] = "0"
import numpy as np, torch
REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, REPO)
from pointcloudsegmentation_amd.clustering import cluster_embeddings
from pointcloudsegmentation_amd.synth import make_embeddings
from pointcloudsegmentation_amd.inference import preprocess_point_cloud
from src.models.point2net import Point2Net


def timed(fn, steps=5):
    fn(); torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(steps):
        r = fn()
    torch.cuda.synchronize()
    return 1e3 * (time.perf_counter() - t0) / steps, r


def code_keys(e, bits=32, seed=5):
    g = torch.Generator(device="cpu").manual_seed(seed)
    r = (torch.randint(0, 2, (e.shape[1], bits), generator=g).float() * 2 - 1).to(e.device)
    s = (e @ r) > 0
    w = (2 ** torch.arange(bits - 1, -1, -1, dtype=torch.int64, device=e.device))
    return (s.long() * w).sum(1)


def orders(e, m):
    key = code_keys(e)
    ks, p = torch.sort(key, stable=True)
    # run length of every sorted position
    head = torch.ones_like(ks, dtype=torch.bool); head[1:] = ks[1:] != ks[:-1]
    rid = torch.cumsum(head.long(), 0) - 1
    rl = torch.bincount(rid)[rid]
    rare = rl < m
    p2 = torch.cat([p[~rare], p[rare]])
    return p, p2, float(rare.float().mean())


def main():
    n = 166499
    cases = {"33 clusters, random order": torch.from_numpy(make_embeddings(n, 64, 11)).cuda()}
    rng = np.random.default_rng(0)
    pts = (rng.uniform(0, 1, (n, 3)) * np.array([8.0, 6.0, 3.0])).astype(np.float32).astype(np.float64)
    col = rng.integers(0, 256, (n, 3)).astype(np.float64)
    w = dict(np.load(os.path.join(REPO, "tests", "golden", "weights.npz")))
    m_ = Point2Net(feature_dim=6, embedding_dim=64)
    m_.load_state_dict({k: torch.from_numpy(np.asarray(v)) for k, v in w.items()}, strict=True)
    m_ = m_.cuda().eval()
    feat, _, _ = preprocess_point_cloud(torch.from_numpy(pts).cuda(), torch.from_numpy(col).cuda())
    with torch.no_grad():
        cases["model embeddings of a random scene"] = m_(feat)[0].contiguous()
    for name, e in cases.items():
        for m in (4, 8, 32):
            p, p2, frac = orders(e, m)
            t1, _ = timed(lambda: cluster_embeddings(e[p].contiguous()))
            t2, _ = timed(lambda: cluster_embeddings(e[p2].contiguous()))
            print(f"{name:38s} code order {t1:6.2f} ms   rare codes (run < {m}, {100*frac:4.1f} % of rows) last {t2:6.2f} ms", flush=True)


if __name__ == "__main__":
    main()
