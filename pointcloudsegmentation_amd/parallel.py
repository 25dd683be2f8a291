"""Data-parallel training over the GPUs of one node.

One process per GPU; the batch is sharded over ranks (B=32 -> 4 clouds per GPU at 8
GPUs) and the only collective per step is ONE flat-bucket all-reduce of the 303,296
fp32 gradients (1.21 MB) over torch.distributed -- backend "nccl" is RCCL over xGMI on
ROCm, "gloo" on CPU.  At this size the collective is latency-bound (about 2.1 MB on
the wire per GPU for a ring of 8), so a single bucket after the backward beats any
overlap scheme; there is nothing to bucket.

BatchNorm uses LOCAL batch statistics (like torch DDP without SyncBatchNorm): with equal
per-rank batch sizes the averaged loss gradient is the gradient of the mean of the
per-rank losses, which differs from the single-process B=32 reference only through the
BN statistics.  Running statistics are averaged across ranks by `sync_bn_buffers`.
"""
from typing import Iterable, List

import torch
import torch.distributed as dist


def _all_reduce_mean(flat: torch.Tensor, world: int) -> None:
    """In-place mean over ranks.  Under RCCL ("nccl") the bucket is reduced where it is and any error of the
    collective propagates (a retry on one rank only would desynchronise the job).  Under gloo a device bucket goes
    through a host copy -- chosen by the backend, never by catching a failed collective: gloo's own device-tensor
    path hung after a few steps with two ranks sharing one GPU (bench.py's one-GPU rehearsal), and gloo is only ever
    the CPU / rehearsal backend here."""
    if flat.is_cuda and dist.get_backend() == "gloo":
        host = flat.cpu()
        dist.all_reduce(host, op=dist.ReduceOp.SUM)
        flat.copy_(host)
    else:
        dist.all_reduce(flat, op=dist.ReduceOp.SUM)
    flat.div_(world)


def _shared_bucket(grads: List[torch.Tensor]):
    """The flat buffer the gradients are views of (train.py allocates one per backward), or None."""
    st = grads[0].untyped_storage()
    if any(g.untyped_storage().data_ptr() != st.data_ptr() or not g.is_contiguous() for g in grads):
        return None
    return torch.empty(0, dtype=grads[0].dtype, device=grads[0].device).set_(st)


def _reduce_through_bucket(tensors: List[torch.Tensor], world: int) -> None:
    """Tensors that do not share a buffer (CPU modules, foreign gradients): one bucket, filled and emptied
    with fused multi-tensor copies."""
    flat = torch.empty(sum(t.numel() for t in tensors), dtype=tensors[0].dtype, device=tensors[0].device)
    views = [v.view_as(t) for v, t in zip(flat.split([t.numel() for t in tensors]), tensors)]
    torch._foreach_copy_(views, tensors)
    _all_reduce_mean(flat, world)
    torch._foreach_copy_(tensors, views)


def allreduce_gradients(params: Iterable[torch.nn.Parameter], world_size: int = None) -> None:
    """Average .grad over all ranks with ONE all-reduce of one flat fp32 bucket.  After a backward through
    the HIP training path the gradients already ARE views of one flat buffer: it is reduced in place."""
    params = [p for p in params if p.requires_grad]
    if not dist.is_available() or not dist.is_initialized():
        return
    world = world_size or dist.get_world_size()
    if world == 1:
        return
    for p in params:
        if p.grad is None:
            p.grad = torch.zeros_like(p)
    grads = [p.grad for p in params]
    bucket = _shared_bucket(grads)
    if bucket is not None:
        _all_reduce_mean(bucket, world)
    else:
        _reduce_through_bucket(grads, world)


def sync_bn_buffers(model: torch.nn.Module) -> None:
    """Average BatchNorm running statistics across ranks (one flat all-reduce)."""
    if not dist.is_available() or not dist.is_initialized() or dist.get_world_size() == 1:
        return
    bufs = [b for n, b in model.named_buffers() if "running" in n]
    _reduce_through_bucket(bufs, dist.get_world_size())
    cache = getattr(model, "_cache", None)
    if cache is not None:                 # in-place foreach copies: make sure the eval image is rebuilt
        cache.key = None
        enc = getattr(model, "encoder", None)
        if enc is not None and getattr(enc, "_cache", None) is not None:
            enc._cache.key = None


def train_step(model, criterion, optimizer, features, labels, max_norm: float = 1.0):
    """One step shaped like reference train.py:118-136: zero_grad, forward, criterion,
    backward, (all-reduce), clip_grad_norm_(1.0), optimizer.step().  Returns the loss dict."""
    points = features[:, :, :3].contiguous()
    optimizer.zero_grad()
    if hasattr(criterion, "prepare"):
        criterion.prepare(points, labels)      # neighbour search + label ids on a second stream, beside the forward
    emb = model(features)
    out = criterion(points, emb, labels)
    out["total_loss"].backward()
    allreduce_gradients(model.parameters())
    torch.nn.utils.clip_grad_norm_(model.parameters(), max_norm=max_norm)
    optimizer.step()
    return out
