"""Scene sharding across the GPUs of a node.

Eval-mode clouds are independent (reference point2net.py:67-87 has no cross-cloud
term once BatchNorm uses running statistics), so the multi-GPU inference path is a
partition of whole scenes over ranks with replicated weights (1.2 MB) and NO
data-path collective.  The only communication is the optional gather of results on
rank 0, which callers that keep results rank-local (bench.py) never run.
"""
from typing import List, Sequence

import torch


def shard_indices(n_items: int, rank: int, world: int) -> List[int]:
    """Round-robin partition: item i belongs to rank i % world (balanced to within one
    item for any n_items; stable as the list grows)."""
    if not (0 <= rank < world):
        raise ValueError(f"rank {rank} outside world of {world}")
    return list(range(rank, n_items, world))


@torch.no_grad()
def infer_sharded(model: torch.nn.Module, scenes: Sequence[torch.Tensor], rank: int, world: int,
                  gather: bool = False):
    """Run `model` on this rank's share of `scenes` (each (N_i, C), ragged sizes allowed).
    Returns {scene_index: (N_i, E) embeddings}; with gather=True rank 0 receives every
    rank's results through torch.distributed (gloo or nccl) and the others get {}."""
    dev = next(model.parameters()).device
    mine = {}
    for i in shard_indices(len(scenes), rank, world):
        x = scenes[i].to(dev, non_blocking=True).unsqueeze(0)
        mine[i] = model(x)[0]
    if not gather or world == 1:
        return mine
    import torch.distributed as dist
    payload = {i: v.cpu() for i, v in mine.items()}
    bucket = [None] * world if rank == 0 else None
    dist.gather_object(payload, bucket, dst=0)
    if rank != 0:
        return {}
    out = {}
    for part in bucket:
        out.update(part)
    return out
