#!/usr/bin/env python3
"""Secondary benchmark (BASELINE.json configs[3]): one training step -- train-mode
forward, InstanceSegLoss, backward, flat-bucket gradient all-reduce (RCCL), grad clip,
Adam -- at global batch 32 x 8192 points, sharded over the ranks (strong scaling).

  python tools/bench_train.py --steps 10
  python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 \
         tools/bench_train.py --steps 10
Prints one JSON line on rank 0 (not the driver's headline line; see bench.py for that)."""
import argparse
import json
import os
import sys
import time

import numpy as np
import torch

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, REPO)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--steps", type=int, default=10)
    ap.add_argument("--warmup", type=int, default=2)
    ap.add_argument("--global-batch", type=int, default=32)
    ap.add_argument("--points", type=int, default=8192)
    ap.add_argument("--bf16", action="store_true", help="model.precision = 'bf16': GEMM operands in bf16 (opt-in mode)")
    ap.add_argument("--x3", action="store_true", help="model.precision = 'fp32x3': three-way bf16 split products, fp32 accuracy")
    ap.add_argument("--no-prepare", action="store_true", help="do not start the loss's neighbour search beside the forward")
    ap.add_argument("--no-loss", action="store_true", help="loss = emb.sum() (model fwd+bwd only)")
    args = ap.parse_args()
    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local = int(os.environ.get("LOCAL_RANK", "0"))
    torch.cuda.set_device(local)
    dev = torch.device("cuda", local)
    if world > 1:
        import torch.distributed as dist
        dist.init_process_group("nccl", device_id=dev)
    from pointcloudsegmentation_amd.parallel import allreduce_gradients
    from pointcloudsegmentation_amd.synth import make_batch, make_labels
    from src.models.losses import InstanceSegLoss
    from src.models.point2net import Point2Net
    assert args.global_batch % world == 0
    b = args.global_batch // world
    torch.manual_seed(0)
    model = Point2Net(6, 64).to(dev).train()
    if args.bf16:
        model.precision = "bf16"
    if args.x3:
        model.precision = "fp32x3"
    crit = InstanceSegLoss(64, 0.1, 0.5)
    # reference train.py:84-88 (Adam, lr 1e-4, weight_decay 1e-5); fused=True is torch's single-kernel implementation of it
    opt = torch.optim.Adam(model.parameters(), lr=1e-4, weight_decay=1e-5, fused=True)
    x = torch.from_numpy(make_batch(b, args.points, seed=200 + rank)).to(dev)
    y = torch.from_numpy(make_labels(b, args.points, seed=rank)).to(dev)
    pts = x[:, :, :3].contiguous()

    def step():
        opt.zero_grad()
        if not args.no_loss and not args.no_prepare:
            crit.prepare(pts, y)
        emb = model(x)
        loss = emb.sum() if args.no_loss else crit(pts, emb, y)["total_loss"]
        loss.backward()
        allreduce_gradients(model.parameters())
        torch.nn.utils.clip_grad_norm_(model.parameters(), max_norm=1.0)
        opt.step()
        return loss

    for _ in range(args.warmup):
        step()
    torch.cuda.synchronize(dev)
    if world > 1:
        dist.barrier()
    t0 = time.perf_counter()
    for _ in range(args.steps):
        loss = step()
    torch.cuda.synchronize(dev)
    if world > 1:
        dist.barrier()
    dt = time.perf_counter() - t0
    if world > 1:
        t = torch.tensor([dt], dtype=torch.float64, device=dev)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        dt = float(t.item())
    if rank == 0:
        pts_step = args.global_batch * args.points
        print(json.dumps({"metric": "training step points/sec (global B=32 x 8192)", "value": pts_step * args.steps / dt,
                          "unit": "points/s", "n_gpus": world, "steps": args.steps, "ms_per_step": 1e3 * dt / args.steps,
                          "scaling": "strong", "dtype": "bf16 GEMM operands, f32 accumulate/store" if args.bf16 else "f32", "data": "synthetic", "loss": float(loss),
                          "config": {"workload": "train fwd + InstanceSegLoss + bwd + grad all-reduce + clip + Adam"
                                     if not args.no_loss else "train fwd + bwd (loss = emb.sum()) + all-reduce + Adam",
                                     "global_batch": args.global_batch, "points_per_cloud": args.points}}), flush=True)
    if world > 1:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
