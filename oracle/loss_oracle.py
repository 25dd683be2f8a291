"""NumPy (float64) restatement of the reference's InstanceSegLoss and of its gradient
w.r.t. the embeddings.  TEST INFRASTRUCTURE ONLY (see oracle/p2n_oracle.c header).

Follows reference src/models/losses.py: InstanceSegLoss.forward :350-373,
DiscriminativeLoss.forward :381-452, SmoothnessLoss.forward :459-489.  Pinned against
the loss values and d loss / d embeddings the reference itself produced
(tests/golden/loss_*.npz, edge_*.npz) by tests/test_loss_oracle.py.  The neighbour
search is oracle.knn (exact fp32 distances, lowest-index ties), column 0 dropped as
the reference drops it (:479).
"""
import numpy as np

from . import oracle


def _normalize(v, axis=-1, eps=1e-12):
    n = np.sqrt((v * v).sum(axis=axis, keepdims=True))
    return v / np.maximum(n, eps), np.maximum(n, eps)


def _normalize_bwd(dy, y, n):
    return (dy - y * (y * dy).sum(axis=-1, keepdims=True)) / n


def instance_seg_loss(points, emb, labels, delta_v=0.1, delta_d=0.5, k=20, want_grad=True):
    """points (B,N,3) f32, emb (B,N,E), labels (B,N) int -> dict of scalars
    (+ 'demb_pull', 'demb_push', 'demb_smooth' in float64)."""
    points = np.asarray(points, np.float32)
    emb = np.asarray(emb, np.float64)
    labels = np.asarray(labels)
    B, N, E = emb.shape
    en, nrm = _normalize(emb)                                   # losses.py:357
    pull = push = smooth = 0.0
    dpull = np.zeros_like(en)
    dpush = np.zeros_like(en)
    dsm = np.zeros_like(en)
    for b in range(B):
        e, lab = en[b], labels[b]
        uniq = np.unique(lab)
        uniq = uniq[uniq != 0]                                  # :392-393
        I = len(uniq)
        if I > 0:
            masks = [lab == u for u in uniq]
            means = np.stack([e[m].mean(0) for m in masks])    # :398-410
            # pull :413-428
            cpull = 1.0 / (I + 1e-6) / B
            for i, m in enumerate(masks):
                diff = e[m] - means[i]
                dist = np.sqrt((diff * diff).sum(1))
                act = dist - delta_v > 0
                pull += np.where(act, dist - delta_v, 0.0).mean() * cpull * B / B
                if want_grad:
                    n_i = m.sum()
                    u = np.where((act & (dist > 0))[:, None], diff / np.maximum(dist, 1e-300)[:, None], 0.0)
                    dpull[b][m] = cpull / n_i * (u - u.sum(0) / n_i)
            # push :431-441
            if I > 1:
                nm, nmn = _normalize(means)
                d = np.sqrt(((nm[:, None, :] - nm[None, :, :]) ** 2).sum(-1))
                iu = np.triu(np.ones((I, I), bool), 1)
                term = np.where(iu, np.maximum(2 * delta_d - d, 0.0), 0.0)
                pairs = np.float32(I * (I - 1) / 2)
                w = 1.0 / (float(pairs) + 1e-6)
                push += term.sum() * w / B
                if want_grad:
                    active = (2 * delta_d - d > 0) & (d > 0) & ~np.eye(I, dtype=bool)
                    coef = np.where(active, -w / B / np.where(d > 0, d, 1.0), 0.0)
                    dnm = (coef[:, :, None] * (nm[:, None, :] - nm[None, :, :])).sum(1)
                    dmean = _normalize_bwd(dnm, nm, nmn)
                    for i, m in enumerate(masks):
                        dpush[b][m] += dmean[i] / m.sum()
        # smoothness :470-487
        idx, _ = oracle.knn(points[b:b + 1], points[b:b + 1], k + 1)
        nbr = idx[0][:, 1:]
        diff = e[nbr] - e[:, None, :]
        dist = np.sqrt((diff * diff).sum(-1))
        smooth += dist.mean() / B
        if want_grad:
            g = np.where((dist > 0)[..., None], diff / np.where(dist > 0, dist, 1.0)[..., None], 0.0) / (N * k * B)
            np.add.at(dsm[b], nbr.reshape(-1), g.reshape(-1, E))
            dsm[b] -= g.sum(1)
    out = {"pull_loss": pull, "push_loss": push, "smoothness_loss": smooth,
           "instance_loss": pull + push, "total_loss": pull + push + 0.1 * smooth}
    if want_grad:
        out["demb_pull"] = _normalize_bwd(dpull, en, nrm)
        out["demb_push"] = _normalize_bwd(dpush, en, nrm)
        out["demb_smooth"] = _normalize_bwd(dsm, en, nrm)
        out["demb_total"] = out["demb_pull"] + out["demb_push"] + 0.1 * out["demb_smooth"]
    return out
