"""Experiment: the library's own working order against rows pre-sorted by a torch-made code (library in the given
order): per-stage kernel times of both."""
import os, sys, time, ctypes
import numpy as np, torch
REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, REPO); sys.path.insert(0, os.path.join(REPO, "tools"))
from pointcloudsegmentation_amd import _lib
from pointcloudsegmentation_amd.clustering import cluster_embeddings, last_cluster_stats
from pointcloudsegmentation_amd.synth import make_embeddings
from exp_cluster_centred import order

n = 166499
e = torch.from_numpy(make_embeddings(n, 64, 11)).cuda()
L = _lib.lib()


def stages(x, steps=5):
    cluster_embeddings(x); torch.cuda.synchronize()
    L.p2n_profile_enable(1)
    t0 = time.perf_counter()
    for _ in range(steps):
        cluster_embeddings(x)
    torch.cuda.synchronize()
    ms = 1e3 * (time.perf_counter() - t0) / steps
    msum = (ctypes.c_float * 32)(); cnt = (ctypes.c_int * 32)()
    L.p2n_profile_collect(32, msum, cnt)
    L.p2n_profile_enable(0)
    k = {nm: round((msum[i] / max(cnt[i], 1)) * (cnt[i] // steps), 3) for nm, i in (("prep", 10), ("count", 11), ("link", 12))}
    return round(ms, 2), k, last_cluster_stats(n, 64, 50)


os.environ["P2N_CL_SORT"] = "1"
print("library's order      ", *stages(e), flush=True)
for seed in (5, 6, 7):
    p = order(e, True, seed=seed)
    es = e[p].contiguous()
    os.environ["P2N_CL_SORT"] = "0"
    print(f"pre-sorted (seed {seed}), given order", *stages(es), flush=True)
    os.environ["P2N_CL_SORT"] = "1"
    print(f"pre-sorted (seed {seed}), library's order on top", *stages(es), flush=True)


def hashed_order(e):
    """the library's own code (cl_project_kernel / cl_code_kernel) restated in torch"""
    D = e.shape[1]
    S = np.zeros((D, 32), np.float32)
    for d in range(D):
        h = (d * 0x9E3779B1 + 0x7F4A7C15) & 0xffffffff
        h ^= h >> 15; h = (h * 0x2C1B3C6D) & 0xffffffff; h ^= h >> 12; h = (h * 0x297A2D39) & 0xffffffff; h ^= h >> 15
        for k in range(32):
            S[d, k] = -1.0 if (h >> k) & 1 else 1.0     # the kernel flips the sign bit where the hash bit is set
    from pointcloudsegmentation_amd.clustering import normalize_rows
    pr = normalize_rows(e) @ torch.from_numpy(S).to(e.device)
    pr = pr - pr.mean(0, keepdim=True)
    w = (2 ** torch.arange(0, 32, dtype=torch.int64, device=e.device))
    key = ((pr > 0).long() * w).sum(1)
    return torch.sort(key, stable=True).indices


p = hashed_order(e)
es = e[p].contiguous()
os.environ["P2N_CL_SORT"] = "0"
print("pre-sorted by the library's code restated in torch, given order", *stages(es), flush=True)


def sign_matrix(D, variant):
    S = np.zeros((D, 32), np.float32)
    for d in range(D):
        if variant == "hash":
            h = (d * 0x9E3779B1 + 0x7F4A7C15) & 0xffffffff
            h ^= h >> 15; h = (h * 0x2C1B3C6D) & 0xffffffff; h ^= h >> 12; h = (h * 0x297A2D39) & 0xffffffff; h ^= h >> 15
        elif variant == "murmur":
            h = (d + 1) * 0x9E3779B9 & 0xffffffff
            h ^= h >> 16; h = (h * 0x85EBCA6B) & 0xffffffff; h ^= h >> 13; h = (h * 0xC2B2AE35) & 0xffffffff; h ^= h >> 16
        for k in range(32):
            S[d, k] = -1.0 if (h >> k) & 1 else 1.0
    return S


from pointcloudsegmentation_amd.clustering import normalize_rows
xn = normalize_rows(e)
for variant in ("hash", "murmur"):
    for msb_first in (False, True):
        pr = xn @ torch.from_numpy(sign_matrix(64, variant)).to(e.device)
        pr = pr - pr.mean(0, keepdim=True)
        ex = torch.arange(31, -1, -1, device=e.device) if msb_first else torch.arange(0, 32, device=e.device)
        key = ((pr > 0).long() * (2 ** ex.long())).sum(1)
        p = torch.sort(key, stable=True).indices
        es = e[p].contiguous()
        print(f"variant {variant}, projection 0 is {'MSB' if msb_first else 'LSB'}:", *stages(es)[:2], flush=True)

print("--- adaptive bit significance: projections with the least mass near their threshold become the top bits", flush=True)
for variant, seed in (("hash", 0), ("murmur", 0), ("torch", 5), ("torch", 6), ("torch", 7)):
    if variant == "torch":
        g = torch.Generator(device="cpu").manual_seed(seed)
        S = (torch.randint(0, 2, (64, 32), generator=g).float() * 2 - 1).to(e.device)
    else:
        S = torch.from_numpy(sign_matrix(64, variant)).to(e.device)
    pr = xn @ S
    pr = pr - pr.mean(0, keepdim=True)
    for delta in (0.05, 0.1):
        near = (pr.abs() < delta).float().sum(0)
        rank = torch.argsort(near)                       # rank[0]: least near-threshold mass -> most significant bit
        ex = torch.empty(32, dtype=torch.int64, device=e.device)
        ex[rank] = torch.arange(31, -1, -1, device=e.device)
        key = ((pr > 0).long() * (2 ** ex)).sum(1)
        p = torch.sort(key, stable=True).indices
        es = e[p].contiguous()
        print(f"variant {variant}/{seed}, delta {delta}:", *stages(es)[:2], flush=True)

print("--- thresholds moved off the mean to the emptiest of a few candidates, then ranked by near mass", flush=True)
S = torch.from_numpy(sign_matrix(64, "hash")).to(e.device)
pr0 = xn @ S
for ncand, span in ((1, 0.0), (5, 0.5), (9, 0.5), (9, 1.0)):
    mu, sd = pr0.mean(0), pr0.std(0)
    best_t, best_near = None, None
    for c in range(ncand):
        off = 0.0 if ncand == 1 else (c / (ncand - 1) - 0.5) * 2 * span
        t = mu + off * sd
        near = ((pr0 - t).abs() < 0.1).float().sum(0) + 1e-3 * abs(off) * pr0.shape[0]     # slight preference for balanced cuts
        if best_t is None:
            best_t, best_near = t.clone(), near.clone()
        else:
            take = near < best_near
            best_t = torch.where(take, t, best_t); best_near = torch.where(take, near, best_near)
    rank = torch.argsort(best_near)
    ex = torch.empty(32, dtype=torch.int64, device=e.device)
    ex[rank] = torch.arange(31, -1, -1, device=e.device)
    key = (((pr0 - best_t) > 0).long() * (2 ** ex)).sum(1)
    p = torch.sort(key, stable=True).indices
    es = e[p].contiguous()
    print(f"candidates {ncand}, span {span} sd:", *stages(es)[:2], flush=True)

print("--- more candidate projections than bits: the 32 with the least near mass, ranked", flush=True)
for ncand in (32, 64, 128):
    g = torch.Generator(device="cpu").manual_seed(11)
    S = (torch.randint(0, 2, (64, ncand), generator=g).float() * 2 - 1).to(e.device)
    pr = xn @ S
    pr = pr - pr.mean(0, keepdim=True)
    near = (pr.abs() < 0.1).float().sum(0)
    rank = torch.argsort(near)[:32]
    key = ((pr[:, rank] > 0).long() * (2 ** torch.arange(31, -1, -1, device=e.device))).sum(1)
    p = torch.sort(key, stable=True).indices
    es = e[p].contiguous()
    print(f"candidates {ncand}:", *stages(es)[:2], flush=True)
