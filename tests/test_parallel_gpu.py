"""GPU, world_size 2 over torch.distributed: the N>1 paths (data-parallel training step, scene-sharded
inference) with DEVICE tensors, i.e. through the HIP kernels -- what tests/test_parallel_gloo.py and
tests/test_sharding_gloo.py check on CPU tensors.  Both ranks share the one GPU of the test box (cuda:0), each
child initialises it itself; the backend is gloo (RCCL needs one device per rank), so the collective moves the
same flat gradient bucket the RCCL path moves.  No scaling curve is claimed from this: it proves the code path.
"""
import datetime
import os
import sys

import numpy as np
import pytest
import torch
import torch.multiprocessing as mp

from conftest import REPO, free_port

pytestmark = pytest.mark.gpu


def _make(seed=0):
    from src.models.losses import InstanceSegLoss
    from src.models.point2net import Point2Net
    torch.manual_seed(seed)
    m = Point2Net(6, 64).train()
    return m, InstanceSegLoss(64, 0.1, 0.5)


def _data(rank):
    from pointcloudsegmentation_amd.synth import make_batch, make_labels
    return (torch.from_numpy(make_batch(2, 512, seed=50 + rank)),
            torch.from_numpy(make_labels(2, 512, seed=rank, num_instances=4)))


def _train_worker(rank, world, port, out_dir):
    sys.path.insert(0, REPO)
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    import torch.distributed as dist
    from pointcloudsegmentation_amd import parallel
    dist.init_process_group("gloo", rank=rank, world_size=world, timeout=datetime.timedelta(seconds=90))
    dev = torch.device("cuda", 0)
    m, crit = _make()
    m = m.to(dev)
    opt = torch.optim.Adam(m.parameters(), lr=1e-4, weight_decay=1e-5)
    x, y = _data(rank)
    out = parallel.train_step(m, crit, opt, x.to(dev), y.to(dev))
    grads = [p.grad for p in m.parameters()]
    zero_copy = parallel._shared_bucket(grads) is not None      # the gradients are views of ONE flat buffer
    parallel.sync_bn_buffers(m)
    torch.save({"sd": {k: v.cpu() for k, v in m.state_dict().items()},
                "grads": {k: p.grad.cpu() for k, p in m.named_parameters()},
                "loss": float(out["total_loss"]), "zero_copy": zero_copy}, os.path.join(out_dir, f"r{rank}.pt"))
    dist.barrier()
    dist.destroy_process_group()


def test_two_rank_training_step_on_device_tensors(tmp_path):
    port = free_port()
    mp.spawn(_train_worker, args=(2, port, str(tmp_path)), nprocs=2, join=True)
    r0 = torch.load(tmp_path / "r0.pt")
    r1 = torch.load(tmp_path / "r1.pt")
    assert r0["zero_copy"] and r1["zero_copy"], "gradients were copied out of the flat bucket"
    for k in r0["sd"]:
        assert torch.equal(r0["sd"][k], r1["sd"][k]), k          # replicas stay in lock-step
    for k in r0["grads"]:
        assert torch.equal(r0["grads"][k], r1["grads"][k]), k    # both ranks hold the same averaged gradient
    # expected averaged (pre-clip) gradient = mean of the two single-process gradients, here from the HIP path
    # in this process; train_step clips after averaging
    dev = torch.device("cuda", 0)
    singles = []
    for rank in range(2):
        m, crit = _make()
        m = m.to(dev)
        x, y = _data(rank)
        x = x.to(dev)
        crit(x[:, :, :3].contiguous(), m(x), y.to(dev))["total_loss"].backward()
        singles.append({k: p.grad.cpu().clone() for k, p in m.named_parameters()})
    mean = {k: (singles[0][k] + singles[1][k]) / 2 for k in singles[0]}
    total = torch.sqrt(sum((v.double() ** 2).sum() for v in mean.values()))
    scale = min(1.0, 1.0 / (float(total) + 1e-6))
    for k, v in mean.items():
        ref = (v * scale).numpy()
        tol = 2e-4 * max(float(np.abs(ref).max()), 1e-6)         # fp32 atomics: summation order differs run to run
        assert np.abs(r0["grads"][k].numpy() - ref).max() <= tol, k


def _infer_worker(rank, world, port, out_path):
    sys.path.insert(0, REPO)
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    import torch.distributed as dist
    from pointcloudsegmentation_amd.sharding import infer_sharded
    from pointcloudsegmentation_amd.synth import make_batch
    from src.models.point2net import Point2Net
    dist.init_process_group("gloo", rank=rank, world_size=world, timeout=datetime.timedelta(seconds=90))
    torch.manual_seed(0)
    model = Point2Net(6, 64).eval().to("cuda:0")
    scenes = [torch.from_numpy(make_batch(1, n, seed=n)[0]) for n in (5000, 1310, 640, 20000, 333)]
    res = infer_sharded(model, scenes, rank, world, gather=True)
    if rank == 0:
        torch.save({k: v for k, v in res.items()}, out_path)
    dist.barrier()
    dist.destroy_process_group()


def test_two_rank_sharded_inference_on_device_tensors(tmp_path):
    from pointcloudsegmentation_amd.sharding import infer_sharded
    from pointcloudsegmentation_amd.synth import make_batch
    from src.models.point2net import Point2Net
    out = str(tmp_path / "res.pt")
    port = free_port()
    mp.spawn(_infer_worker, args=(2, port, out), nprocs=2, join=True)
    got = torch.load(out)
    torch.manual_seed(0)
    model = Point2Net(6, 64).eval().to("cuda:0")
    scenes = [torch.from_numpy(make_batch(1, n, seed=n)[0]) for n in (5000, 1310, 640, 20000, 333)]
    want = infer_sharded(model, scenes, 0, 1)
    assert sorted(got) == [0, 1, 2, 3, 4]
    for i in range(5):
        assert torch.equal(got[i], want[i].cpu()), i            # eval clouds are independent: bit-identical
