#!/usr/bin/env python3
"""Generate the golden vectors under tests/golden/ by running the REFERENCE.

Runs only in the build container (needs /root/reference); the GPU box gets the
.npz files this script wrote, never the reference.  What it records
(SURVEY.md §8c):

  G1  weights.npz         the 58 state_dict tensors of the shipped checkpoint
                          (reference outputs/best_model_test.pth, loaded with
                          weights_only=True)
  G2  eval_*.npz          Point2Net eval forward (reference point2net.py:67-87)
  G3  train_*.npz         train-mode forward, BN running stats after 1 and 2
                          steps, parameter grads for two losses
  G4  loss_*.npz          InstanceSegLoss terms (reference losses.py:332-489)
                          and kNN neighbour sets
  G5  edge_*.npz          edge cases (all background, one instance, duplicates)
  G6  train_gapsafe_*.npz full training step on a cloud whose neighbour sets are stable
                          under the reference's cdist rounding (noise-free gradients)

No reference source text is stored: only inputs and the numbers it produced.
Usage:  PYTHONDONTWRITEBYTECODE=1 python tests/golden/make_golden.py
"""
import os
import sys

import numpy as np
import torch

HERE = os.path.dirname(os.path.abspath(__file__))
REPO = os.path.dirname(os.path.dirname(HERE))
REF = os.environ.get("P2N_REFERENCE", "/root/reference")
sys.dont_write_bytecode = True
sys.path.insert(0, REPO)
sys.path.insert(0, REF)

from pointcloudsegmentation_amd.synth import make_batch, make_labels  # noqa: E402
from src.models.point2net import Point2Net  # noqa: E402  (the reference's)
import importlib.util  # noqa: E402

_spec = importlib.util.spec_from_file_location("ref_losses", os.path.join(REF, "src/models/losses.py"))
ref_losses = importlib.util.module_from_spec(_spec)
_spec.loader.exec_module(ref_losses)

torch.set_num_threads(8)
torch.manual_seed(0)


def load_ref_model():
    ck = torch.load(os.path.join(REF, "outputs/best_model_test.pth"), map_location="cpu", weights_only=True)
    m = Point2Net(feature_dim=6, embedding_dim=64)
    m.load_state_dict(ck["model_state_dict"], strict=True)
    return m, ck["model_state_dict"]


def save(name, **arrs):
    path = os.path.join(HERE, name)
    np.savez_compressed(path, **arrs)
    print(f"{name}: {os.path.getsize(path)/1e6:.2f} MB  [{', '.join(arrs)}]")


def g1_weights(sd):
    save("weights.npz", **{k: v.detach().cpu().numpy() for k, v in sd.items()})


def g2_eval(model):
    model.eval()
    for (b, n, seed) in [(1, 1, 0), (1, 7, 1), (2, 1000, 0), (3, 4099, 1)]:
        x = make_batch(b, n, seed)
        with torch.no_grad():
            xt = torch.from_numpy(x)
            f1, f2, f3, g = model.encoder(xt)
            emb = model(xt)
        arrs = dict(x=x, global_f=g.numpy(), emb=emb.contiguous().numpy())
        if (b, n) == (2, 1000):   # per-layer activations to localise a mismatch
            arrs.update(f1=f1.numpy(), f2=f2.numpy(), f3=f3.numpy())
        save(f"eval_b{b}_n{n}_s{seed}.npz", **arrs)
    # headline shape: inputs are regenerated from the seed, outputs subsampled 1/64
    b, n, seed = 16, 8192, 0
    x = make_batch(b, n, seed)
    with torch.no_grad():
        xt = torch.from_numpy(x)
        _, _, _, g = model.encoder(xt)
        emb = model(xt).contiguous().numpy()
    save("eval_b16_n8192_s0.npz", seed=np.int64(seed), global_f=g.numpy(),
         emb_stride64=emb[:, ::64, :].copy(), x_head=x[:, :4, :].copy())
    # a random-weight model with a non-default embedding_dim / feature_dim = 6
    torch.manual_seed(123)
    m2 = Point2Net(feature_dim=6, embedding_dim=32)
    # make the BN running stats non-trivial
    with torch.no_grad():
        for mod in m2.modules():
            if isinstance(mod, torch.nn.BatchNorm1d):
                mod.running_mean.uniform_(-0.2, 0.2)
                mod.running_var.uniform_(0.5, 1.5)
                mod.weight.uniform_(0.5, 1.5)
                mod.bias.uniform_(-0.2, 0.2)
    m2.eval()
    x = make_batch(2, 333, 5)
    with torch.no_grad():
        emb = m2(torch.from_numpy(x)).contiguous().numpy()
    save("eval_e32_b2_n333_s5.npz", x=x, emb=emb,
         **{"sd." + k: v.numpy() for k, v in m2.state_dict().items()})


def bn_stats(model):
    out = {}
    for k, v in model.state_dict().items():
        if "running" in k or "num_batches" in k:
            out[k] = v.detach().numpy().copy()
    return out


def g3_train(sd):
    b, n, seed = 4, 2048, 3
    x = make_batch(b, n, seed)
    labels = make_labels(b, n, seed)
    xt = torch.from_numpy(x)
    # (a) loss = emb.sum(), two steps of running-stat updates
    m = Point2Net(6, 64)
    m.load_state_dict(sd)
    m.train()
    emb = m(xt)
    emb.sum().backward()
    arrs = dict(x=x, labels=labels, emb=emb.detach().contiguous().numpy())
    arrs.update({"step1." + k: v for k, v in bn_stats(m).items()})
    arrs.update({"gradsum." + k: p.grad.numpy().copy() for k, p in m.named_parameters()})
    m.zero_grad()
    _ = m(xt)
    arrs.update({"step2." + k: v for k, v in bn_stats(m).items()})
    # (b) loss = InstanceSegLoss total
    m = Point2Net(6, 64)
    m.load_state_dict(sd)
    m.train()
    crit = ref_losses.InstanceSegLoss(feature_dim=64, delta_v=0.1, delta_d=0.5)
    emb = m(xt)
    ld = crit(xt[:, :, :3], emb, torch.from_numpy(labels))
    ld["total_loss"].backward()
    arrs.update({"loss." + k: np.float32(v.item()) for k, v in ld.items()})
    arrs.update({"gradloss." + k: p.grad.numpy().copy() for k, p in m.named_parameters()})
    save("train_b4_n2048_s3.npz", **arrs)

    # small random-weight case incl. dL/dx-free check of a different shape (B*N not mult of 32)
    torch.manual_seed(7)
    m = Point2Net(6, 64)
    m.train()
    x = make_batch(3, 77, 11)
    w = torch.randn(3, 77, 64)
    xt = torch.from_numpy(x)
    emb = m(xt)
    (emb * w).sum().backward()
    arrs = dict(x=x, w=w.numpy(), emb=emb.detach().contiguous().numpy())
    arrs.update({"sd." + k: v.numpy().copy() for k, v in Point2Net(6, 64).state_dict().items()} if False else {})
    torch.manual_seed(7)
    m0 = Point2Net(6, 64)
    arrs.update({"sd." + k: v.numpy().copy() for k, v in m0.state_dict().items()})
    arrs.update({"step1." + k: v for k, v in bn_stats(m).items()})
    arrs.update({"grad." + k: p.grad.numpy().copy() for k, p in m.named_parameters()})
    save("train_b3_n77_s11.npz", **arrs)


def knn_sets_exact(pts, k):
    d = torch.cdist(pts, pts, compute_mode="donot_use_mm_for_euclid_dist")
    idx = torch.argsort(d, dim=1, stable=True)[:, : k + 1]
    return idx


def g4_losses(model):
    model.eval()
    crit = ref_losses.InstanceSegLoss(feature_dim=64, delta_v=0.1, delta_d=0.5)
    for (b, n, seed) in [(2, 2048, 4), (2, 8192, 5)]:
        x = make_batch(b, n, seed)
        labels = make_labels(b, n, seed)
        xt = torch.from_numpy(x)
        with torch.no_grad():
            emb = model(xt).contiguous()
        embg = emb.clone().requires_grad_(True)
        ld = crit(xt[:, :, :3], embg, torch.from_numpy(labels))
        ld["total_loss"].backward()
        arrs = dict(seed=np.int64(seed), labels=labels,
                    **{"loss." + k: np.float32(v.item()) for k, v in ld.items()})
        if n == 2048:
            arrs.update(x=x, emb=emb.numpy(), demb=embg.grad.numpy())
            # reference default-mode topk (ordered, col 0 dropped as the reference does)
            ref_idx, exact_idx = [], []
            for i in range(b):
                p = xt[i, :, :3]
                _, idx = torch.cdist(p, p).topk(k=21, dim=1, largest=False)
                ref_idx.append(idx[:, 1:].numpy().astype(np.int32))
                exact_idx.append(knn_sets_exact(p, 20).numpy().astype(np.int32))
            arrs.update(knn_ref_default=np.stack(ref_idx), knn_exact_k21=np.stack(exact_idx))
        else:
            arrs.update(emb_stride64=emb.numpy()[:, ::64].copy(), demb_stride64=embg.grad.numpy()[:, ::64].copy())
        save(f"loss_b{b}_n{n}_s{seed}.npz", **arrs)


def g5_edges(model):
    model.eval()
    crit = ref_losses.InstanceSegLoss(feature_dim=64, delta_v=0.1, delta_d=0.5)
    b, n, seed = 2, 256, 9
    x = make_batch(b, n, seed)
    xt = torch.from_numpy(x)
    with torch.no_grad():
        emb = model(xt).contiguous()
    cases = {
        "allbg": np.zeros((b, n), np.int64),
        "one": np.full((b, n), 5, np.int64),
        "two_and_bg": (np.arange(b * n).reshape(b, n) % 3).astype(np.int64),
        "int32like": (np.arange(b * n).reshape(b, n) % 7 * 1000).astype(np.int64),
    }
    arrs = dict(x=x, emb=emb.numpy())
    for name, lab in cases.items():
        embg = emb.clone().requires_grad_(True)
        ld = crit(xt[:, :, :3], embg, torch.from_numpy(lab))
        ld["total_loss"].backward()
        arrs[f"{name}.labels"] = lab
        arrs[f"{name}.demb"] = embg.grad.numpy()
        for k, v in ld.items():
            arrs[f"{name}.{k}"] = np.float32(v.item())
    # duplicated points: every point appears twice -> self is not necessarily column 0
    xd = x.copy()
    xd[:, n // 2:, :3] = xd[:, : n // 2, :3]
    xdt = torch.from_numpy(xd)
    sm = ref_losses.SmoothnessLoss()
    arrs["dup.x"] = xd
    arrs["dup.smoothness"] = np.float32(sm(xdt[:, :, :3], emb).item())
    save("edge_b2_n256_s9.npz", **arrs)


def gap_safe_cloud(b, side, seed, min_gap=2e-4, k=20):
    """Jittered anisotropic lattice, normalised like reference inference.py:88-91, re-jittered point by point
    until every point's (k+1)-th and (k+2)-th nearest distances (self included) differ by more than min_gap:
    the reference's matmul-form cdist (errors ~1e-6 at these distances) then picks exactly the neighbour
    SETS exact arithmetic picks, so its loss and gradients carry no neighbour-selection noise."""
    rng = np.random.default_rng(seed)
    n = side ** 3
    g = np.stack(np.meshgrid(*[np.arange(side)] * 3, indexing="ij"), -1).reshape(-1, 3).astype(np.float64)
    out = np.zeros((b, n, 3), np.float32)
    for i in range(b):
        step = np.array([1.0, 1.13, 1.29]) * (1.0 + 0.05 * i)
        base = g * step
        pts = base + rng.uniform(-0.18, 0.18, base.shape)
        for _ in range(200):
            q = pts - pts.mean(0)
            q = (q / np.abs(q).max()).astype(np.float32)
            d = np.sqrt(((q[:, None, :].astype(np.float64) - q[None, :, :]) ** 2).sum(-1))
            ds = np.sort(d, axis=1)
            bad = np.where(ds[:, k + 1] - ds[:, k] < min_gap)[0]
            if len(bad) == 0:
                break
            pts[bad] = base[bad] + rng.uniform(-0.18, 0.18, (len(bad), 3))
        else:
            raise RuntimeError("no gap-safe jitter found")
        out[i] = q[rng.permutation(n)]
    return out


def g6_gap_safe_training_step(sd):
    """Full training step (train-mode forward, InstanceSegLoss, backward) on a cloud whose 20-neighbour sets
    are stable under the reference's cdist rounding: pins model-backward o loss-backward end to end."""
    b, side, seed = 2, 12, 21
    xyz = gap_safe_cloud(b, side, seed)
    n = xyz.shape[1]
    rng = np.random.default_rng(seed + 1)
    x = np.concatenate([xyz, rng.uniform(0, 1, (b, n, 3)).astype(np.float32)], axis=2)
    labels = make_labels(b, n, seed)
    xt = torch.from_numpy(x)
    m = Point2Net(6, 64)
    m.load_state_dict(sd)
    m.train()
    crit = ref_losses.InstanceSegLoss(feature_dim=64, delta_v=0.1, delta_d=0.5)
    emb = m(xt)
    emb.retain_grad()
    ld = crit(xt[:, :, :3], emb, torch.from_numpy(labels))
    ld["total_loss"].backward()
    # the neighbour sets the reference really used (default matmul-form cdist) must equal the exact ones
    same = 0
    for i in range(b):
        p = xt[i, :, :3]
        _, idx = torch.cdist(p, p).topk(k=21, dim=1, largest=False)
        ex = knn_sets_exact(p, 20)
        same += int((np.sort(idx.numpy(), 1) == np.sort(ex.numpy(), 1)).all(1).sum())
    assert same == b * n, f"reference neighbour sets differ from the exact ones on {b * n - same} rows"
    arrs = dict(x=x, labels=labels, emb=emb.detach().contiguous().numpy(), demb=emb.grad.contiguous().numpy())
    arrs.update({"loss." + k: np.float32(v.item()) for k, v in ld.items()})
    arrs.update({"gradloss." + k: p.grad.numpy().copy() for k, p in m.named_parameters()})
    arrs.update({"step1." + k: v for k, v in bn_stats(m).items()})
    # the same step by the same reference code in float64: separates the two float32 computations' own
    # rounding (the reference's CPU float32 run above, the GPU's) from a real defect
    m64 = Point2Net(6, 64)
    m64.load_state_dict(sd)
    m64 = m64.double().train()
    emb64 = m64(xt.double())
    emb64.retain_grad()
    ld64 = crit(xt[:, :, :3].double(), emb64, torch.from_numpy(labels))
    ld64["total_loss"].backward()
    arrs.update({"loss64." + k: np.float64(v.item()) for k, v in ld64.items()})
    arrs.update({"gradloss64." + k: p.grad.numpy().astype(np.float32) for k, p in m64.named_parameters()})
    arrs["demb64"] = emb64.grad.contiguous().numpy().astype(np.float32)
    save(f"train_gapsafe_b{b}_n{n}_s{seed}.npz", **arrs)


if __name__ == "__main__":
    only = set(sys.argv[1:])
    model, sd = load_ref_model()
    if not only or "g1" in only:
        g1_weights(sd)
    if not only or "g2" in only:
        g2_eval(model)
    if not only or "g3" in only:
        g3_train(sd)
    if not only or "g4" in only:
        g4_losses(model)
    if not only or "g5" in only:
        g5_edges(model)
    if not only or "g6" in only:
        g6_gap_safe_training_step(sd)
