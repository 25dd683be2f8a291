#!/usr/bin/env python3
"""Per-tensor gradient errors of the training goldens in one precision (P2N_X3_MASK selects which GEMM families run in
fp32x3).  usage: debug_x3_train.py <fp32|fp32x3>"""
import os, sys
REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, REPO); sys.path.insert(0, os.path.join(REPO, "tests"))
import numpy as np, torch
from conftest import golden
from test_train_gpu import _grad_scale
from src.models.point2net import Point2Net
prec = sys.argv[1]
w = dict(np.load(os.path.join(REPO, "tests", "golden", "weights.npz")))
g = golden("train_b4_n2048_s3.npz")
res = []
for rep in range(3):
    m = Point2Net(6, 64); m.load_state_dict({k: torch.from_numpy(np.asarray(v)) for k, v in w.items()}); m = m.cuda().train(); m.precision = prec
    x = torch.from_numpy(g["x"]).cuda()
    emb = m(x)
    e_emb = np.abs(emb.detach().cpu().numpy() - g["emb"]).max() / np.abs(g["emb"]).max()
    emb.sum().backward()
    errs = {}
    for k, p in m.named_parameters():
        scale = _grad_scale(k, lambda q: g["gradsum." + q])
        errs[k] = float(np.abs(p.grad.cpu().numpy() - g["gradsum." + k]).max() / scale)
    top = sorted(errs.items(), key=lambda kv: -kv[1])[:4]
    res.append((e_emb, top))
print(prec, "mask", os.environ.get("P2N_X3_MASK", "7"))
for e_emb, top in res:
    print(f"  emb {e_emb:.2e} | " + ", ".join(f"{k.replace('encoder.','e.').replace('embedding_net.','h.')} {v:.2e}" for k, v in top))
