"""Point-set operators on the GPU (libp2n_hip.so): FPS, ball query, kNN, grouping
gather, three-NN inverse-distance interpolation.

BASELINE.json's north_star names these; the reference has none of them except the
brute-force kNN in its smoothness loss (reference src/models/losses.py:475-479),
so their behaviour is specified by oracle/p2n_oracle.c (parity unpinned by the
reference).  Index outputs are int32.  All functions take GPU tensors only.
"""
import torch

from . import _lib
from . import runtime as _rt


def _xyz(t: torch.Tensor, name: str) -> torch.Tensor:
    t = _rt.require_cuda_f32(t, name)
    if t.dim() != 3 or t.size(2) != 3:
        raise ValueError(f"{name} must be (B,N,3), got {tuple(t.shape)}")
    return t


def _idx(t: torch.Tensor, name: str) -> torch.Tensor:
    if not t.is_cuda:
        raise ValueError(f"{name} must be a GPU tensor")
    if t.dtype != torch.int32:
        t = t.to(torch.int32)
    return t.contiguous()


class FpsExchangeTimeout(_lib.P2nError):
    """A multi-workgroup FPS call gave up waiting for a cloud's other workgroups (they were not all resident,
    e.g. because another stream's kernels held the compute units).  The indices after that point repeat the
    start index: valid to gather with, but not a farthest-point sample."""


_fps_status = {}   # device index -> int32[1], set to 1 by a timed-out exchange, private to FPS


def _fps_status_word(device: torch.device) -> torch.Tensor:
    key = device.index if device.index is not None else torch.cuda.current_device()
    w = _fps_status.get(key)
    if w is None:
        w = torch.zeros(1, dtype=torch.int32, device=device)
        _fps_status[key] = w
    return w


def farthest_point_sample(xyz: torch.Tensor, npoint: int, start_idx: int = 0) -> torch.Tensor:
    """xyz (B,N,3) -> (B,npoint) int32.  Sample 0 is start_idx; then argmax of the
    running min squared distance, lowest index on ties.  Clouds sampled by several workgroups
    (fps_needs_check) can fail when those are not co-resident: the failure is recorded on the device
    and raised by fps_check() -- call it where the indices are consumed on the host side of a
    synchronisation (DeviceBatchProducer, Point2NetSA and inference.predict do)."""
    xyz = _xyz(xyz, "xyz")
    B, N, _ = xyz.shape
    L = _lib.lib()
    idx = torch.empty((B, npoint), dtype=torch.int32, device=xyz.device)
    nbytes = L.p2n_fps_workspace_bytes(B, max(N, 1), max(npoint, 1))
    ws = _rt.workspace(nbytes, xyz.device)
    status = _fps_status_word(xyz.device)
    with torch.cuda.device(xyz.device):
        _lib.check(L.p2n_fps(xyz.data_ptr(), B, N, npoint, start_idx, idx.data_ptr(), status.data_ptr(), ws.data_ptr(),
                             ws.numel(), _rt.stream_handle(xyz.device)), "p2n_fps")
    return idx


def fps_needs_check(B: int, N: int) -> bool:
    """True when clouds of this size are sampled by a cluster of waves or workgroups (the only way FPS can fail)."""
    return _lib.lib().p2n_fps_workgroups_per_cloud(max(B, 1), max(N, 1)) > 1


def fps_needs_whole_chip(B: int, N: int) -> bool:
    """True when the clouds are sampled by clusters of fat workgroups, planned as if they owned every compute unit:
    run such a call alone (no kernels of other streams beside it).  Wave-member clusters (4,096..160,000 points) are
    light enough to become resident beside other kernels."""
    return _lib.lib().p2n_fps_cluster_kind(max(B, 1), max(N, 1)) == 2


def fps_status(device) -> int:
    """0 = every FPS call on this device since the last fps_check finished its exchanges; 1 = one timed
    out.  Synchronises the device's current stream."""
    return int(_fps_status_word(torch.device(device)).item())


def fps_check(device) -> None:
    """Raise FpsExchangeTimeout if an FPS call on this device timed out since the last check (and reset)."""
    w = _fps_status_word(torch.device(device))
    if int(w.item()) != 0:
        w.zero_()
        raise FpsExchangeTimeout("farthest_point_sample: a workgroup exchange timed out (the cloud's workgroups "
                                 "were not co-resident); the sampled indices are not valid")


def ball_query(xyz: torch.Tensor, new_xyz: torch.Tensor, radius: float, nsample: int) -> torch.Tensor:
    """xyz (B,N,3), new_xyz (B,M,3) -> (B,M,nsample) int32."""
    xyz, new_xyz = _xyz(xyz, "xyz"), _xyz(new_xyz, "new_xyz")
    B, N, _ = xyz.shape
    M = new_xyz.size(1)
    if new_xyz.size(0) != B:
        raise ValueError("batch mismatch")
    idx = torch.empty((B, M, nsample), dtype=torch.int32, device=xyz.device)
    with torch.cuda.device(xyz.device):
        _lib.check(_lib.lib().p2n_ball_query(xyz.data_ptr(), new_xyz.data_ptr(), B, N, M, float(radius), nsample,
                                             idx.data_ptr(), _rt.stream_handle(xyz.device)), "p2n_ball_query")
    return idx


def knn(query: torch.Tensor, ref: torch.Tensor, k: int, return_dist: bool = True):
    """query (B,M,3), ref (B,N,3) -> idx (B,M,k) int32 [, d2 (B,M,k)]: the k smallest
    (d2, index) pairs in ascending order."""
    query, ref = _xyz(query, "query"), _xyz(ref, "ref")
    B, M, _ = query.shape
    N = ref.size(1)
    if ref.size(0) != B:
        raise ValueError("batch mismatch")
    if query.data_ptr() == ref.data_ptr() and query.shape == ref.shape and 1 <= k <= min(N, 64):
        return knn_self(ref, k, return_dist)      # the cloud against itself: same result, Morton-ordered search
    idx = torch.empty((B, M, k), dtype=torch.int32, device=query.device)
    d2 = torch.empty((B, M, k), dtype=torch.float32, device=query.device) if return_dist else None
    with torch.cuda.device(query.device):
        _lib.check(_lib.lib().p2n_knn(query.data_ptr(), ref.data_ptr(), B, M, N, k, idx.data_ptr(), _rt.ptr(d2),
                                      _rt.stream_handle(query.device)), "p2n_knn")
    return (idx, d2) if return_dist else idx


def knn_self(xyz: torch.Tensor, k: int, return_dist: bool = True):
    """xyz (B,N,3) -> the result of knn(xyz, xyz, k), searched through a Morton-ordered copy of
    each cloud (reference losses.py:475-479 is exactly this self search)."""
    xyz = _xyz(xyz, "xyz")
    B, N, _ = xyz.shape
    L = _lib.lib()
    idx = torch.empty((B, N, k), dtype=torch.int32, device=xyz.device)
    d2 = torch.empty((B, N, k), dtype=torch.float32, device=xyz.device) if return_dist else None
    ws = _rt.workspace(L.p2n_knn_self_workspace_bytes(B, max(N, 1)), xyz.device)
    with torch.cuda.device(xyz.device):
        _lib.check(L.p2n_knn_self(xyz.data_ptr(), B, N, k, idx.data_ptr(), _rt.ptr(d2), ws.data_ptr(), ws.numel(),
                                  _rt.stream_handle(xyz.device)), "p2n_knn_self")
    return (idx, d2) if return_dist else idx


def three_nn(unknown: torch.Tensor, known: torch.Tensor):
    """unknown (B,n,3), known (B,m,3) -> dist2 (B,n,3), idx (B,n,3), weight (B,n,3)."""
    unknown, known = _xyz(unknown, "unknown"), _xyz(known, "known")
    B, n, _ = unknown.shape
    m = known.size(1)
    d2 = torch.empty((B, n, 3), dtype=torch.float32, device=unknown.device)
    idx = torch.empty((B, n, 3), dtype=torch.int32, device=unknown.device)
    w = torch.empty((B, n, 3), dtype=torch.float32, device=unknown.device)
    with torch.cuda.device(unknown.device):
        _lib.check(_lib.lib().p2n_three_nn(unknown.data_ptr(), known.data_ptr(), B, n, m, d2.data_ptr(),
                                           idx.data_ptr(), w.data_ptr(), _rt.stream_handle(unknown.device)),
                   "p2n_three_nn")
    return d2, idx, w


class _GroupGather(torch.autograd.Function):
    @staticmethod
    def forward(ctx, feat, idx):
        feat = _rt.require_cuda_f32(feat, "feat")
        idx = _idx(idx, "idx")
        B, C, N = feat.shape
        _, M, ns = idx.shape
        out = torch.empty((B, C, M, ns), dtype=torch.float32, device=feat.device)
        with torch.cuda.device(feat.device):
            _lib.check(_lib.lib().p2n_group_gather(feat.data_ptr(), idx.data_ptr(), B, C, N, M, ns, out.data_ptr(),
                                                   _rt.stream_handle(feat.device)), "p2n_group_gather")
        ctx.save_for_backward(idx)
        ctx.n = N
        return out

    @staticmethod
    def backward(ctx, gout):
        (idx,) = ctx.saved_tensors
        gout = gout.contiguous()
        B, C, M, ns = gout.shape
        gfeat = torch.empty((B, C, ctx.n), dtype=torch.float32, device=gout.device)
        with torch.cuda.device(gout.device):
            _lib.check(_lib.lib().p2n_group_gather_bwd(gout.data_ptr(), idx.data_ptr(), B, C, ctx.n, M, ns,
                                                       gfeat.data_ptr(), _rt.stream_handle(gout.device)),
                       "p2n_group_gather_bwd")
        return gfeat, None


def group_gather(feat: torch.Tensor, idx: torch.Tensor) -> torch.Tensor:
    """feat (B,C,N), idx (B,M,ns) -> (B,C,M,ns); differentiable in feat."""
    if feat.dim() != 3 or idx.dim() != 3 or feat.size(0) != idx.size(0):
        raise ValueError(f"group_gather: feat {tuple(feat.shape)} / idx {tuple(idx.shape)}")
    return _GroupGather.apply(feat, idx)


class _ThreeInterpolate(torch.autograd.Function):
    @staticmethod
    def forward(ctx, feat, idx, weight):
        feat = _rt.require_cuda_f32(feat, "feat")
        weight = _rt.require_cuda_f32(weight, "weight")
        idx = _idx(idx, "idx")
        B, C, m = feat.shape
        n = idx.size(1)
        out = torch.empty((B, C, n), dtype=torch.float32, device=feat.device)
        with torch.cuda.device(feat.device):
            _lib.check(_lib.lib().p2n_three_interpolate(feat.data_ptr(), idx.data_ptr(), weight.data_ptr(), B, C, m,
                                                        n, out.data_ptr(), _rt.stream_handle(feat.device)),
                       "p2n_three_interpolate")
        ctx.save_for_backward(idx, weight)
        ctx.m = m
        return out

    @staticmethod
    def backward(ctx, gout):
        idx, weight = ctx.saved_tensors
        gout = gout.contiguous()
        B, C, n = gout.shape
        gfeat = torch.empty((B, C, ctx.m), dtype=torch.float32, device=gout.device)
        with torch.cuda.device(gout.device):
            _lib.check(_lib.lib().p2n_three_interpolate_bwd(gout.data_ptr(), idx.data_ptr(), weight.data_ptr(), B, C,
                                                            ctx.m, n, gfeat.data_ptr(),
                                                            _rt.stream_handle(gout.device)),
                       "p2n_three_interpolate_bwd")
        return gfeat, None, None


def three_interpolate(feat: torch.Tensor, idx: torch.Tensor, weight: torch.Tensor) -> torch.Tensor:
    """feat (B,C,m), idx/weight (B,n,3) -> (B,C,n); differentiable in feat."""
    if feat.dim() != 3 or idx.dim() != 3 or idx.size(2) != 3 or idx.shape != weight.shape:
        raise ValueError("three_interpolate: bad shapes")
    return _ThreeInterpolate.apply(feat, idx, weight)
