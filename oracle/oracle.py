"""ctypes front-end of the CPU oracle (oracle/p2n_oracle.c).  TEST INFRASTRUCTURE ONLY.

Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg import this
module; nothing under pointcloudsegmentation_amd/ or src/ does.
"""
import ctypes
import os
import subprocess

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
_SO = os.path.join(_HERE, "libp2n_oracle.so")
_lib = None

_f = ctypes.POINTER(ctypes.c_float)
_i = ctypes.POINTER(ctypes.c_int32)


def build(force: bool = False) -> str:
    src = os.path.join(_HERE, "p2n_oracle.c")
    if force or not os.path.exists(_SO) or os.path.getmtime(_SO) < os.path.getmtime(src):
        subprocess.check_call(["make", "-C", _HERE, "-B", "libp2n_oracle.so"], stdout=subprocess.DEVNULL)
    return _SO


def lib():
    global _lib
    if _lib is None:
        if not os.path.exists(_SO):
            build()
        _lib = ctypes.CDLL(_SO)
        _lib.orc_param_count.restype = ctypes.c_long
    return _lib


def _fp(a):
    return a.ctypes.data_as(_f) if a is not None else None


def _ip(a):
    return a.ctypes.data_as(_i) if a is not None else None


def _c32(a):
    a = np.ascontiguousarray(a, dtype=np.float32)
    return a


def forward_eval(x, params, embedding_dim=64, want_features=False):
    """x (B,N,F) float32, params flat float32 -> dict(emb, global_f[, f1, f2, f3])."""
    x = _c32(x)
    params = _c32(params)
    B, N, F = x.shape
    E = embedding_dim
    assert params.size == lib().orc_param_count(F, E), (params.size, lib().orc_param_count(F, E))
    emb = np.empty((B, N, E), np.float32)
    g = np.empty((B, 256), np.float32)
    f1 = np.empty((B, 64, N), np.float32) if want_features else None
    f2 = np.empty((B, 128, N), np.float32) if want_features else None
    f3 = np.empty((B, 256, N), np.float32) if want_features else None
    rc = lib().orc_forward_eval(_fp(x), B, N, F, E, _fp(params), _fp(emb), _fp(g), _fp(f1), _fp(f2), _fp(f3))
    if rc:
        raise RuntimeError(f"orc_forward_eval rc={rc}")
    out = dict(emb=emb, global_f=g)
    if want_features:
        out.update(f1=f1, f2=f2, f3=f3)
    return out


def fps(xyz, m, start_idx=0):
    xyz = _c32(xyz)
    B, N, _ = xyz.shape
    idx = np.empty((B, m), np.int32)
    rc = lib().orc_fps(_fp(xyz), B, N, m, start_idx, _ip(idx))
    if rc:
        raise ValueError(f"orc_fps rc={rc}")
    return idx


def ball_query(xyz, new_xyz, radius, nsample):
    xyz, new_xyz = _c32(xyz), _c32(new_xyz)
    B, N, _ = xyz.shape
    M = new_xyz.shape[1]
    idx = np.empty((B, M, nsample), np.int32)
    rc = lib().orc_ball_query(_fp(xyz), _fp(new_xyz), B, N, M, ctypes.c_float(radius), nsample, _ip(idx))
    if rc:
        raise ValueError(f"orc_ball_query rc={rc}")
    return idx


def knn(query, ref, k):
    query, ref = _c32(query), _c32(ref)
    B, M, _ = query.shape
    N = ref.shape[1]
    if k > 256:
        raise ValueError("oracle kNN supports k <= 256")
    idx = np.empty((B, M, k), np.int32)
    d2 = np.empty((B, M, k), np.float32)
    rc = lib().orc_knn(_fp(query), _fp(ref), B, M, N, k, _ip(idx), _fp(d2))
    if rc:
        raise ValueError(f"orc_knn rc={rc} (k > N?)")
    return idx, d2


def group_gather(feat, idx):
    feat = _c32(feat)
    idx = np.ascontiguousarray(idx, dtype=np.int32)
    B, C, N = feat.shape
    _, M, ns = idx.shape
    out = np.empty((B, C, M, ns), np.float32)
    rc = lib().orc_group_gather(_fp(feat), _ip(idx), B, C, N, M, ns, _fp(out))
    if rc:
        raise ValueError("orc_group_gather: index out of range")
    return out


def three_nn(unknown, known):
    unknown, known = _c32(unknown), _c32(known)
    B, n, _ = unknown.shape
    m = known.shape[1]
    d2 = np.empty((B, n, 3), np.float32)
    idx = np.empty((B, n, 3), np.int32)
    rc = lib().orc_three_nn(_fp(unknown), _fp(known), B, n, m, _fp(d2), _ip(idx))
    if rc:
        raise ValueError("orc_three_nn: need at least 3 known points")
    return d2, idx


def three_weights(d2):
    d2 = _c32(d2)
    w = np.empty_like(d2)
    lib().orc_three_weights(_fp(d2), ctypes.c_long(d2.size // 3), _fp(w))
    return w


def three_interpolate(feat, idx, w):
    feat, w = _c32(feat), _c32(w)
    idx = np.ascontiguousarray(idx, dtype=np.int32)
    B, C, m = feat.shape
    n = idx.shape[1]
    out = np.empty((B, C, n), np.float32)
    rc = lib().orc_three_interpolate(_fp(feat), _ip(idx), _fp(w), B, C, m, n, _fp(out))
    if rc:
        raise ValueError("orc_three_interpolate: index out of range")
    return out


def normalize_rows(emb):
    """reference inference.py:124 in numpy's float32 evaluation order."""
    emb = _c32(emb)
    out = np.empty_like(emb)
    rc = lib().orc_normalize_rows(_fp(emb), ctypes.c_long(emb.shape[0]), emb.shape[1], _fp(out))
    if rc:
        raise ValueError(f"orc_normalize_rows rc={rc}")
    return out


def dbscan(x_normalised, eps=0.1, min_samples=50, return_counts=False):
    """scikit-learn DBSCAN semantics on already-normalised rows; labels 0 = noise, 1.. clusters
    (reference inference.py:127-140)."""
    x = _c32(x_normalised)
    n, d = x.shape
    labels = np.empty(n, np.int32)
    counts = np.empty(n, np.int32)
    rc = lib().orc_dbscan(_fp(x), ctypes.c_long(n), d, ctypes.c_double(eps), int(min_samples), _ip(labels), _ip(counts))
    if rc:
        raise ValueError(f"orc_dbscan rc={rc}")
    return (labels, counts) if return_counts else labels


def cluster_embeddings(emb, eps=0.1, min_samples=50):
    """reference inference.py:106-142 end to end."""
    return dbscan(normalize_rows(emb), eps, min_samples)
