"""Embedding clustering on the GPU: the drop-in for the reference's
`PointCloudInference.cluster_embeddings` (reference inference.py:106-142): L2-normalise
the rows, scikit-learn DBSCAN(eps, min_samples) semantics, noise -> 0 and clusters -> 1..
in scikit-learn's label order.  Runs in libp2n_hip.so (csrc/p2n_cluster.hip); there is no
CPU path here: without the library or a GPU the call raises.
"""
import numpy as np
import torch

from . import _lib
from . import runtime as _rt


def _as_device_rows(embeddings, what):
    if isinstance(embeddings, np.ndarray):
        if not torch.cuda.is_available():
            raise RuntimeError(f"{what}: needs a GPU (the HIP library is the only implementation)")
        t = torch.from_numpy(np.ascontiguousarray(embeddings, dtype=np.float32)).cuda()
        return t, True
    t = _rt.require_cuda_f32(embeddings, "embeddings")
    return t, False


def normalize_rows(embeddings):
    """embeddings / (||embeddings||_2 + 1e-8) row-wise in float32, bit-identical to numpy's
    evaluation of reference inference.py:124.  (N,D) numpy array or GPU tensor, D <= 256."""
    t, was_numpy = _as_device_rows(embeddings, "normalize_rows")
    if t.dim() != 2:
        raise ValueError(f"embeddings must be (N,D), got {tuple(t.shape)}")
    out = torch.empty_like(t)
    with torch.cuda.device(t.device):
        _lib.check(_lib.lib().p2n_normalize_rows(t.data_ptr(), t.size(0), t.size(1), out.data_ptr(),
                                                 _rt.stream_handle(t.device)), "p2n_normalize_rows")
    return out.cpu().numpy() if was_numpy else out


def cluster_embeddings(embeddings, eps: float = 0.1, min_samples: int = 50, return_counts: bool = False):
    """Same signature and result as the reference method (inference.py:106-142).

    embeddings: (N,D) numpy array (returns a numpy int64 array, like the reference) or a
    float32 GPU tensor (returns an int64 GPU tensor).  Label 0 = noise, 1..K = clusters.
    return_counts=True also returns the DBSCAN neighbourhood sizes (int32)."""
    t, was_numpy = _as_device_rows(embeddings, "cluster_embeddings")
    if t.dim() != 2:
        raise ValueError(f"embeddings must be (N,D), got {tuple(t.shape)}")
    n, d = t.shape
    L = _lib.lib()
    labels = torch.empty(n, dtype=torch.int32, device=t.device)
    nbytes = L.p2n_cluster_workspace_bytes(n, d, int(min_samples))
    if nbytes == 0:
        raise ValueError(f"cluster_embeddings: unsupported shape/arguments N={n} D={d} min_samples={min_samples}")
    ws = _rt.workspace(nbytes, t.device)
    with torch.cuda.device(t.device):
        s = _rt.stream_handle(t.device)
        _lib.check(L.p2n_cluster_embeddings(t.data_ptr(), n, d, float(eps), int(min_samples), labels.data_ptr(), None,
                                            ws.data_ptr(), ws.numel(), s), "p2n_cluster_embeddings")
        counts = None
        if return_counts:
            counts = torch.empty(n, dtype=torch.int32, device=t.device)
            _lib.check(L.p2n_cluster_counts(ws.data_ptr(), n, d, int(min_samples), counts.data_ptr(), s), "p2n_cluster_counts")
    out = labels.to(torch.int64)
    if was_numpy:
        out = out.cpu().numpy()
        counts = counts.cpu().numpy() if counts is not None else None
    return (out, counts) if return_counts else out


def last_cluster_stats(n: int, d: int, min_samples: int, device="cuda"):
    """Slow-path counters of the last cluster_embeddings call with these sizes on this
    device/stream: dict(exact_pass1, exact_pass2, unions, border_records)."""
    device = torch.device(device)
    L = _lib.lib()
    ws = _rt.workspace(L.p2n_cluster_workspace_bytes(n, d, int(min_samples)), device)
    out = torch.zeros(4, dtype=torch.int64, device=device)
    with torch.cuda.device(device):
        _lib.check(L.p2n_cluster_stats(ws.data_ptr(), n, d, int(min_samples), out.data_ptr(), _rt.stream_handle(device)),
                   "p2n_cluster_stats")
    v = out.cpu().tolist()
    return dict(exact_pass1=v[0], exact_pass2=v[1], unions=v[2], border_records=v[3])
