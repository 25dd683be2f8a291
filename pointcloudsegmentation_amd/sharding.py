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
    ids = shard_indices(len(scenes), rank, world)
    if dev.type != "cuda":
        for i in ids:
            mine[i] = model(scenes[i].to(dev).unsqueeze(0))[0]
    else:
        # scene i+1 is uploaded from a pinned staging buffer on a side stream while scene i is computed
        main, side = torch.cuda.current_stream(dev), torch.cuda.Stream(device=dev)

        def stage(i):
            with torch.cuda.stream(side):
                src = scenes[i]
                if not src.is_cuda:
                    pinned = torch.empty(src.shape, dtype=src.dtype, pin_memory=True)
                    pinned.copy_(src)
                    src = pinned
                x = src.to(dev, non_blocking=True)
                ev = torch.cuda.Event()
                ev.record(side)
            return x, ev

        nxt = stage(ids[0]) if ids else None
        for k, i in enumerate(ids):
            x, ev = nxt
            main.wait_event(ev)
            x.record_stream(main)
            if k + 1 < len(ids):
                nxt = stage(ids[k + 1])
            mine[i] = model(x.unsqueeze(0))[0]
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
