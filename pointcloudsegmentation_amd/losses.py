"""Host-side mirror of the reference loss module (reference src/models/losses.py).

`InstanceSegLoss(feature_dim, delta_v, delta_d)` (reference :332-373, constructed at
train.py:69-73) and the classes it is built from keep their names, arguments and
returned dict keys.  GPU tensors run libp2n_hip.so (p2n_instance_seg_loss: values and
d/d embeddings in one call); CPU tensors run a stock-PyTorch composition of the same
formulas.  `FocalLoss` is kept because train.py:19 imports it; in the reference it
raises on every call (BCE target (B,N,I) vs input (B,N,N)) and
it is out of scope here, so it raises NotImplementedError with that explanation.
"""
from typing import Dict

import torch
import torch.nn as nn
import torch.nn.functional as F

from . import _lib
from . import runtime as _rt


_MAX_LABELS_PER_CLOUD = 1024


def _dense_instances_launch(labels: torch.Tensor):
    """Integer plumbing for the HIP kernels: (B,N) labels -> dense ids per (cloud,label)
    in ascending (cloud,label) order, counts, background flags, per-cloud id ranges.
    p2n_dense_instances does it in three launches (one workgroup per cloud peels off the distinct labels in
    ascending order); clouds with more than 1024 distinct labels take the torch.unique route below.
    This half queues the kernels and the copy of {number of instances, overflow} into pinned host memory on the
    current stream; _dense_instances_finish reads them once that stream has got there."""
    B, N = labels.shape
    lab = labels.to(torch.int64).contiguous()
    L = _lib.lib()
    dev = lab.device
    cap = B * _MAX_LABELS_PER_CLOUD
    inst = torch.empty(B * N, dtype=torch.int32, device=dev)
    meta = torch.empty(2 * cap + B + 2, dtype=torch.int32, device=dev)     # cnt | bg | cstart (B+1) | overflow
    cnt, bg, cstart, over = meta[:cap], meta[cap:2 * cap], meta[2 * cap:2 * cap + B + 1], meta[2 * cap + B + 1:]
    ws = _rt.workspace(L.p2n_dense_instances_workspace_bytes(B, _MAX_LABELS_PER_CLOUD), dev)
    with torch.cuda.device(dev):
        _lib.check(L.p2n_dense_instances(lab.data_ptr(), B, N, _MAX_LABELS_PER_CLOUD, inst.data_ptr(), cnt.data_ptr(),
                                         bg.data_ptr(), cstart.data_ptr(), over.data_ptr(), ws.data_ptr(), ws.numel(),
                                         _rt.stream_handle(dev)), "p2n_dense_instances")
    tail = torch.empty(2, dtype=torch.int32, pin_memory=True)
    tail.copy_(meta[2 * cap + B:], non_blocking=True)   # {cstart[B] = number of instances, overflow}: the one host read
    return lab, inst, cnt, bg, cstart, tail


def _dense_instances_finish(state):
    """after the launching stream has been waited for (stream or event synchronize)"""
    lab, inst, cnt, bg, cstart, tail = state
    if int(tail[1]) == 0:
        T = int(tail[0])
        return inst, cnt[:T], bg[:T], cstart, T
    return _dense_instances_torch(lab)


def _dense_instances(labels: torch.Tensor):
    state = _dense_instances_launch(labels)
    torch.cuda.current_stream(labels.device).synchronize()
    return _dense_instances_finish(state)


def _dense_instances_torch(lab: torch.Tensor):
    """the same through torch.unique per cloud, as reference losses.py:389-395 enumerates the instances (any label
    values, any number of them); only clouds with more than 1024 distinct labels come here"""
    B, N = lab.shape
    inst, cnt, bg, cstart, base = [], [], [], [0], 0
    for b in range(B):
        uniq, inv, c = torch.unique(lab[b], return_inverse=True, return_counts=True)
        inst.append(inv + base)
        cnt.append(c)
        bg.append(uniq == 0)
        base += int(uniq.numel())
        cstart.append(base)
    return (torch.cat(inst).to(torch.int32).contiguous(), torch.cat(cnt).to(torch.int32).contiguous(),
            torch.cat(bg).to(torch.int32).contiguous(), torch.tensor(cstart, dtype=torch.int32, device=lab.device), base)


class _Geometry:
    """What the loss needs from the coordinates and the labels alone (nothing from the embeddings): dense instance
    ids (`dense`: the launched state of _dense_instances_launch) and the k+1 nearest neighbours of every point.
    `InstanceSegLoss.prepare` queues both on a second stream while the model's forward runs; `ready` is recorded
    on that stream."""
    __slots__ = ("key", "dense", "knn", "ready")

    @staticmethod
    def key_of(points, labels, k):
        return (points.data_ptr(), points._version, tuple(points.shape), labels.data_ptr(), labels._version,
                tuple(labels.shape), int(k))


_side_streams = {}


def _side_stream(dev: torch.device) -> torch.cuda.Stream:
    idx = dev.index if dev.index is not None else torch.cuda.current_device()
    st = _side_streams.get(idx)
    if st is None:
        st = _side_streams[idx] = torch.cuda.Stream(device=dev)
    return st


def _check_inputs(points, emb_shape, labels, emb_device):
    B, N = emb_shape[0], emb_shape[1]
    if points.shape != (B, N, 3) or labels.shape != (B, N):
        raise ValueError(f"points {tuple(points.shape)} / embeddings {tuple(emb_shape)} / labels "
                         f"{tuple(labels.shape)} do not agree")
    if not labels.is_cuda or labels.device != emb_device or points.device != emb_device:
        # raw pointers go to the kernels: a host or foreign-device tensor would fault the GPU where
        # stock PyTorch raises a device-mismatch error
        raise ValueError(f"points ({points.device}), embeddings ({emb_device}) and instance_labels "
                         f"({labels.device}) must be on the same GPU")
    if labels.dtype.is_floating_point or labels.dtype == torch.bool:
        raise TypeError(f"instance_labels must be an integer tensor, got {labels.dtype}")


def _knn_self_idx(points: torch.Tensor, k1: int) -> torch.Tensor:
    """(B,N,k1) int32 neighbour table of p2n_knn_self on the current stream (reference losses.py:475-479)."""
    L = _lib.lib()
    B, N, _ = points.shape
    dev = points.device
    idx = torch.empty((B, N, k1), dtype=torch.int32, device=dev)
    ws = _rt.workspace(L.p2n_knn_self_workspace_bytes(B, N), dev)
    with torch.cuda.device(dev):
        _lib.check(L.p2n_knn_self(points.data_ptr(), B, N, k1, idx.data_ptr(), None, ws.data_ptr(), ws.numel(),
                                  _rt.stream_handle(dev)), "p2n_knn_self")
    return idx


class _InstanceSegLossFn(torch.autograd.Function):
    @staticmethod
    def forward(ctx, points, emb, labels, delta_v, delta_d, k, geo):
        L = _lib.lib()
        points = _rt.require_cuda_f32(points, "points")
        emb_c = _rt.require_cuda_f32(emb, "embeddings")
        B, N, E = emb_c.shape
        _check_inputs(points, emb_c.shape, labels, emb_c.device)
        dev = emb_c.device
        knn = None
        if geo is not None:          # prepared on the side stream: this stream waits for it, the tensors change hands
            geo.ready.synchronize()                       # host: the instance count is needed for the launch below
            torch.cuda.current_stream(dev).wait_event(geo.ready)
            inst, cnt, bg, cstart, T = _dense_instances_finish(geo.dense)
            knn = geo.knn
            for t in (inst, cnt, bg, cstart, knn):
                t.record_stream(torch.cuda.current_stream(dev))
        else:
            inst, cnt, bg, cstart, T = _dense_instances(labels)
        want = ctx.needs_input_grad[1]
        losses = torch.empty(3, dtype=torch.float32, device=dev)
        grads = [torch.empty_like(emb_c) for _ in range(3)] if want else [None, None, None]
        nbytes = L.p2n_instance_seg_loss_workspace_bytes(B, N, E, T, k)
        ws = _rt.workspace(nbytes, dev)
        with torch.cuda.device(dev):
            if knn is None:
                _lib.check(L.p2n_instance_seg_loss(points.data_ptr(), emb_c.data_ptr(), inst.data_ptr(), cnt.data_ptr(),
                                                   bg.data_ptr(), cstart.data_ptr(), B, N, E, T, float(delta_v),
                                                   float(delta_d), int(k), losses.data_ptr(), _rt.ptr(grads[0]),
                                                   _rt.ptr(grads[1]), _rt.ptr(grads[2]), ws.data_ptr(), ws.numel(),
                                                   _rt.stream_handle(dev)), "p2n_instance_seg_loss")
            else:
                _lib.check(L.p2n_instance_seg_loss_knn(points.data_ptr(), emb_c.data_ptr(), inst.data_ptr(),
                                                       cnt.data_ptr(), bg.data_ptr(), cstart.data_ptr(), knn.data_ptr(),
                                                       B, N, E, T, float(delta_v), float(delta_d), int(k),
                                                       losses.data_ptr(), _rt.ptr(grads[0]), _rt.ptr(grads[1]),
                                                       _rt.ptr(grads[2]), ws.data_ptr(), ws.numel(),
                                                       _rt.stream_handle(dev)), "p2n_instance_seg_loss_knn")
        if want:
            ctx.save_for_backward(*grads)
        return losses

    @staticmethod
    def backward(ctx, g):
        dp, dq, ds = ctx.saved_tensors
        g = g.to(torch.float32).contiguous()
        out = torch.empty_like(dp)
        n = out.numel()
        if n % 4 or not g.is_cuda:
            return None, g[0] * dp + g[1] * dq + g[2] * ds, None, None, None, None, None
        with torch.cuda.device(out.device):          # (g0*dp + g1*dq) + g2*ds in one pass, weights read on the device
            _lib.check(_lib.lib().p2n_combine3(g.data_ptr(), dp.data_ptr(), dq.data_ptr(), ds.data_ptr(), n, out.data_ptr(),
                                               _rt.stream_handle(out.device)), "p2n_combine3")
        return None, out, None, None, None, None, None


class DiscriminativeLoss(nn.Module):
    """reference losses.py:375-452 (CPU tensors: stock PyTorch)."""

    def __init__(self, delta_v=0.1, delta_d=0.5):
        super().__init__()
        self.delta_v = delta_v
        self.delta_d = delta_d

    def forward(self, embeddings: torch.Tensor, labels: torch.Tensor) -> Dict[str, torch.Tensor]:
        if embeddings.is_cuda:
            raise RuntimeError("on GPU tensors use InstanceSegLoss (one fused HIP call); "
                               "DiscriminativeLoss alone is only provided for CPU tensors")
        B = embeddings.size(0)
        pull = embeddings.new_zeros(())
        push = embeddings.new_zeros(())
        for b in range(B):
            e, lab = embeddings[b], labels[b]
            uniq = torch.unique(lab)
            uniq = uniq[uniq != 0]
            if len(uniq) == 0:
                continue
            means = torch.stack([e[lab == u].mean(0) for u in uniq])
            inst_pull = embeddings.new_zeros(())
            for i, u in enumerate(uniq):
                dist = torch.norm(e[lab == u] - means[i].unsqueeze(0), p=2, dim=1)
                inst_pull = inst_pull + torch.clamp(dist - self.delta_v, min=0).mean()
            pull = pull + inst_pull / (means.size(0) + 1e-6)
            if means.size(0) > 1:
                nm = F.normalize(means, p=2, dim=1)
                d = torch.cdist(nm, nm)
                mask = torch.triu(torch.ones_like(d), diagonal=1)
                push = push + (torch.clamp(2 * self.delta_d - d, min=0) * mask).sum() / (mask.sum() + 1e-6)
        pull, push = pull / B, push / B
        return {"total_loss": pull + push, "pull_loss": pull, "push_loss": push}


class SmoothnessLoss(nn.Module):
    """reference losses.py:454-489 (CPU tensors: stock PyTorch)."""

    def __init__(self, k: int = 20):
        super().__init__()
        self.k = k

    def forward(self, points: torch.Tensor, embeddings: torch.Tensor) -> torch.Tensor:
        if embeddings.is_cuda:
            raise RuntimeError("on GPU tensors use InstanceSegLoss (one fused HIP call)")
        B = points.size(0)
        loss = points.new_zeros(())
        for b in range(B):
            d = torch.cdist(points[b], points[b])
            _, idx = d.topk(k=self.k + 1, dim=1, largest=False)
            idx = idx[:, 1:]
            e = embeddings[b]
            loss = loss + torch.norm(e[idx] - e.unsqueeze(1), p=2, dim=2).mean()
        return loss / B


class InstanceSegLoss(nn.Module):
    """reference losses.py:332-373: criterion(points, embeddings, labels) -> dict."""

    def __init__(self, feature_dim: int = 64, delta_v: float = 0.1, delta_d: float = 0.5):
        super().__init__()
        self.feature_dim = feature_dim
        self.delta_v = delta_v
        self.delta_d = delta_d
        self.instance_loss = DiscriminativeLoss(delta_v=delta_v, delta_d=delta_d)
        self.smoothness_loss = SmoothnessLoss()
        self._geo = None

    def prepare(self, points: torch.Tensor, instance_labels: torch.Tensor) -> None:
        """Optional, GPU tensors only: start the part of the loss that depends on the coordinates and the labels alone
        -- label densification (with its one host read) and the k+1-nearest-neighbour search -- on a second stream,
        to run beside the model's forward instead of between forward and backward.  The next forward() with the SAME
        two tensors (same storage, unmodified since) picks the result up; any other call ignores it.  No counterpart
        in the reference (train.py:124-127 runs model, then criterion); results are identical with or without it."""
        self._geo = None
        if not (points.is_cuda and instance_labels.is_cuda):
            return
        k = self.smoothness_loss.k
        points = _rt.require_cuda_f32(points, "points")
        if points.dim() != 3 or points.size(1) < k + 1:
            return                     # forward() raises the proper error
        _check_inputs(points, (points.size(0), points.size(1)), instance_labels, points.device)
        dev = points.device
        cur, side = torch.cuda.current_stream(dev), _side_stream(dev)
        geo = _Geometry()
        geo.key = _Geometry.key_of(points, instance_labels, k)
        side.wait_stream(cur)          # the inputs are ready where the current stream stands
        with torch.cuda.stream(side):
            geo.knn = _knn_self_idx(points, k + 1)
            geo.dense = _dense_instances_launch(instance_labels)   # read back in forward(): nothing blocks here
            geo.ready = torch.cuda.Event()
            geo.ready.record(side)
        points.record_stream(side)
        instance_labels.record_stream(side)
        self._geo = geo

    def forward(self, points: torch.Tensor, embeddings: torch.Tensor,
                instance_labels: torch.Tensor) -> Dict[str, torch.Tensor]:
        if embeddings.is_cuda:
            k = self.smoothness_loss.k
            if points.size(1) < k + 1:
                raise RuntimeError(f"selected index k out of range: {k + 1} neighbours from {points.size(1)} points")
            geo, self._geo = self._geo, None
            if geo is not None and (not points.is_contiguous() or points.dtype != torch.float32 or
                                    geo.key != _Geometry.key_of(points, instance_labels, k)):
                geo = None             # prepared for other tensors (or they were modified since): search again
            L = _InstanceSegLossFn.apply(points, embeddings, instance_labels, self.delta_v, self.delta_d, k, geo)
            pull, push, smooth = L[0], L[1], L[2]
            inst = pull + push
            return {"total_loss": inst + 0.1 * smooth, "instance_loss": inst, "smoothness_loss": smooth,
                    "pull_loss": pull, "push_loss": push}
        en = F.normalize(embeddings, p=2, dim=2)
        inst = self.instance_loss(en, instance_labels)
        smooth = self.smoothness_loss(points, en)
        return {"total_loss": inst["total_loss"] + 0.1 * smooth, "instance_loss": inst["total_loss"],
                "smoothness_loss": smooth, "pull_loss": inst["pull_loss"], "push_loss": inst["push_loss"]}


class FocalLoss(nn.Module):
    """Name kept for `from src.models.losses import InstanceSegLoss, FocalLoss`
    (reference train.py:19).  The reference's implementation (:492-543) raises on every
    call and train.py calls it with the wrong arity (train.py:127); it is out of scope."""

    def __init__(self, alpha: float = 0.25, gamma: float = 2.0):
        super().__init__()
        self.alpha = alpha
        self.gamma = gamma

    def forward(self, *args, **kwargs):
        raise NotImplementedError("FocalLoss is broken in the reference (shape mismatch in BCE) and is not ported; "
                                  "use loss.type: instance_seg")
