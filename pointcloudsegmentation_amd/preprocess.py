"""Input normalisation on the GPU: drop-in for the reference's
`PointCloudInference.preprocess_point_cloud` (reference inference.py:76-104).
Runs in libp2n_hip.so (csrc/p2n_prep.hip); no CPU path: without a GPU the call raises.
"""
import numpy as np
import torch

from . import _lib
from . import runtime as _rt


def _f64_device(a, name):
    if a is None:
        return None
    if isinstance(a, np.ndarray):
        if not torch.cuda.is_available():
            raise RuntimeError("preprocess_point_cloud: needs a GPU (the HIP library is the only implementation)")
        a = _rt.upload(a, torch.float64, torch.device("cuda", torch.cuda.current_device()))
    if not (isinstance(a, torch.Tensor) and a.is_cuda):
        raise ValueError(f"{name} must be a numpy array or a GPU tensor")
    a = a.to(torch.float64).contiguous()
    if a.dim() != 2 or a.size(1) != 3:
        raise ValueError(f"{name} must be (N,3), got {tuple(a.shape)}")
    return a


def preprocess_point_cloud(points, colors=None, sync: bool = True):
    """points (N,3), colors (N,3) or None -> (features (1,N,6) float32 GPU tensor, centroid (3,)
    float64 numpy, scale float64): the reference's three return values, bit-identical to its
    numpy float64 evaluation.  The reference returns a CPU tensor that its caller moves to the
    device (inference.py:169); here it is already there.  sync=False returns (features, cs) with
    cs = {cx, cy, cz, scale} as a float64 GPU tensor and does not wait for the device (pipelines)."""
    p = _f64_device(points, "points")
    c = _f64_device(colors, "colors")
    n = p.size(0)
    if n < 1:
        raise ValueError("points is empty")
    if c is not None and c.size(0) != n:
        raise ValueError("points and colors differ in length")
    L = _lib.lib()
    feat = torch.empty((1, n, 6), dtype=torch.float32, device=p.device)
    cs = torch.empty(4, dtype=torch.float64, device=p.device)
    ws = _rt.workspace(L.p2n_preprocess_workspace_bytes(), p.device)
    with torch.cuda.device(p.device):
        _lib.check(L.p2n_preprocess_points(p.data_ptr(), c.data_ptr() if c is not None else None, n, feat.data_ptr(),
                                           cs.data_ptr(), ws.data_ptr(), ws.numel(), _rt.stream_handle(p.device)),
                   "p2n_preprocess_points")
    if not sync:
        return feat, cs
    h = cs.cpu().numpy()
    return feat, h[:3].copy(), np.float64(h[3])


def preprocess_stats(points, colors=None):
    """The reduction half of preprocess_point_cloud: (points, colors | None as float64 GPU tensors, cs8) with cs8 =
    {cx, cy, cz, scale, colour flag, ...} as a float64 GPU tensor -- what `embed_scene` hands to the encoder, whose
    layer-0 loader applies the affine itself (p2n_forward_eval_raw).  Does not wait for the device."""
    p = _f64_device(points, "points")
    c = _f64_device(colors, "colors")
    n = p.size(0)
    if n < 1:
        raise ValueError("points is empty")
    if c is not None and c.size(0) != n:
        raise ValueError("points and colors differ in length")
    L = _lib.lib()
    cs8 = torch.empty(8, dtype=torch.float64, device=p.device)       # its own block: the shared workspace is reused by the forward
    with torch.cuda.device(p.device):
        _lib.check(L.p2n_preprocess_points(p.data_ptr(), c.data_ptr() if c is not None else None, n, None, None,
                                           cs8.data_ptr(), 64, _rt.stream_handle(p.device)), "p2n_preprocess_points")
    return p, c, cs8


def embed_scene(model, p, c, cs8):
    """Embeddings (N,E) of one scene from its raw float64 arrays and preprocess_stats' block: bit-identical to
    model(preprocess_point_cloud(points, colors)[0])[0], one launch and one (N,6) round trip fewer.  fp32 precision,
    feature_dim 6 (anything else takes the two-step path)."""
    L = _lib.lib()
    n, E = p.size(0), model.embedding_dim
    packed = model._cache.get(model, model.feature_dim, E, precision="fp32")
    emb = torch.empty((n, E), dtype=torch.float32, device=p.device)
    ws = _rt.workspace(L.p2n_forward_eval_workspace_bytes(1, n, 6, E), p.device)
    with torch.cuda.device(p.device):
        _lib.check(L.p2n_forward_eval_raw(p.data_ptr(), c.data_ptr() if c is not None else None, n, E, packed.data_ptr(),
                                          cs8.data_ptr(), emb.data_ptr(), None, ws.data_ptr(), ws.numel(),
                                          _rt.stream_handle(p.device)), "p2n_forward_eval_raw")
    return emb


def fused_ok(model) -> bool:
    return getattr(model, "precision", "fp32") == "fp32" and model.feature_dim == 6
