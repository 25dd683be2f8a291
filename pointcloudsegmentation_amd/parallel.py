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


def _flat_grads(params: List[torch.nn.Parameter]) -> torch.Tensor:
    return torch.cat([(p.grad if p.grad is not None else torch.zeros_like(p)).reshape(-1) for p in params])


def allreduce_gradients(params: Iterable[torch.nn.Parameter], world_size: int = None) -> None:
    """Average .grad over all ranks with one all-reduce of one flat fp32 bucket."""
    params = [p for p in params if p.requires_grad]
    if not dist.is_available() or not dist.is_initialized():
        return
    world = world_size or dist.get_world_size()
    if world == 1:
        return
    bucket = _flat_grads(params)
    dist.all_reduce(bucket, op=dist.ReduceOp.SUM)
    bucket.div_(world)
    off = 0
    for p in params:
        n = p.numel()
        if p.grad is None:
            p.grad = torch.empty_like(p)
        p.grad.copy_(bucket[off:off + n].view_as(p))
        off += n


def sync_bn_buffers(model: torch.nn.Module) -> None:
    """Average BatchNorm running statistics across ranks (one flat all-reduce)."""
    if not dist.is_available() or not dist.is_initialized() or dist.get_world_size() == 1:
        return
    bufs = [b for n, b in model.named_buffers() if "running" in n]
    flat = torch.cat([b.reshape(-1) for b in bufs])
    dist.all_reduce(flat, op=dist.ReduceOp.SUM)
    flat.div_(dist.get_world_size())
    off = 0
    for b in bufs:
        b.copy_(flat[off:off + b.numel()].view_as(b))
        off += b.numel()


def train_step(model, criterion, optimizer, features, labels, max_norm: float = 1.0):
    """One step shaped like reference train.py:118-136: zero_grad, forward, criterion,
    backward, (all-reduce), clip_grad_norm_(1.0), optimizer.step().  Returns the loss dict."""
    points = features[:, :, :3].contiguous()
    optimizer.zero_grad()
    emb = model(features)
    out = criterion(points, emb, labels)
    out["total_loss"].backward()
    allreduce_gradients(model.parameters())
    torch.nn.utils.clip_grad_norm_(model.parameters(), max_norm=max_norm)
    optimizer.step()
    return out
