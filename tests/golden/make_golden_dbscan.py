#!/usr/bin/env python3
"""Golden vectors for the embedding-clustering row.

The reference's cluster_embeddings (inference.py:106-142) cannot be imported here
(inference.py needs open3d, which is not installed), so this script drives the same
third-party call the reference makes -- scikit-learn's DBSCAN(eps, min_samples,
n_jobs=-1).fit_predict on the L2-normalised rows -- and applies the reference's relabelling
rule (noise -> 0, clusters -> 1.. in ascending label order) to its output.  scikit-learn
version used is recorded in each file.  Stored: inputs and the labels only.
Usage:  python tests/golden/make_golden_dbscan.py
"""
import os
import sys

import numpy as np
import sklearn
from sklearn.cluster import DBSCAN

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.dirname(os.path.dirname(HERE)))
from pointcloudsegmentation_amd.synth import make_embeddings  # noqa: E402


def run(emb, eps, min_samples):
    x = emb / (np.linalg.norm(emb, axis=1, keepdims=True) + 1e-8)
    lab = DBSCAN(eps=eps, min_samples=min_samples, n_jobs=-1).fit_predict(x)
    out = np.zeros_like(lab)
    nxt = 1
    for u in np.unique(lab):
        if u == -1:
            out[lab == u] = 0
        else:
            out[lab == u] = nxt
            nxt += 1
    return x, out


def save(name, emb, eps=0.1, min_samples=50):
    emb = np.ascontiguousarray(emb, dtype=np.float32)
    x, lab = run(emb, eps, min_samples)
    np.savez_compressed(os.path.join(HERE, name + ".npz"), emb=emb, labels=lab.astype(np.int64),
                        eps=np.float64(eps), min_samples=np.int64(min_samples), sklearn_version=np.array(sklearn.__version__))
    print(f"{name}: N={len(emb)} D={emb.shape[1]} clusters={lab.max()} noise={(lab == 0).sum()}")


def main():
    rng = np.random.default_rng(0)
    save("dbscan_n3000_d64", make_embeddings(3000, 64, 3000, n_clusters=8))
    save("dbscan_n6000_d64_ms10", make_embeddings(6000, 64, 6000, n_clusters=40), min_samples=10)
    save("dbscan_n2500_d32", make_embeddings(2500, 32, 7, n_clusters=12), min_samples=20)
    save("dbscan_n2000_d3", make_embeddings(2000, 3, 8, n_clusters=6, intrinsic_dim=2), eps=0.05, min_samples=15)
    save("dbscan_n1500_d100", make_embeddings(1500, 100, 9, n_clusters=5), min_samples=30)
    save("dbscan_all_noise", rng.standard_normal((500, 64)))
    one = np.ones((300, 64)) + rng.standard_normal((300, 64)) * 1e-3
    save("dbscan_one_cluster", one)
    save("dbscan_fewer_than_min_samples", one[:20])
    dup = np.repeat(rng.standard_normal((3, 64)), 200, axis=0)
    save("dbscan_duplicates", dup[rng.permutation(600)])
    # a chain on the unit circle (embedded in 64-D): two dense arcs joined by sparse points ->
    # border points contested between clusters, order-dependent in a naive implementation
    ang = np.concatenate([rng.uniform(0.0, 0.5, 400), rng.uniform(0.62, 1.1, 400), rng.uniform(0.5, 0.62, 6)])
    chain = np.zeros((len(ang), 64))
    chain[:, 0], chain[:, 1] = np.cos(ang), np.sin(ang)
    save("dbscan_chain_borders", chain[rng.permutation(len(ang))], eps=0.02, min_samples=12)
    zero = make_embeddings(400, 64, 5, n_clusters=2)
    zero[::7] = 0.0            # zero rows stay zero after x / (0 + 1e-8)
    save("dbscan_zero_rows", zero, min_samples=20)


if __name__ == "__main__":
    main()
