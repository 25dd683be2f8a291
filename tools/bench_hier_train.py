#!/usr/bin/env python3
"""Point2NetSA training step on the device (forward + backward through the train-mode shared MLPs and the
operators' adjoints) at B x N points; prints ONE JSON line with wall ms per step and the library's per-kernel ms."""
import argparse
import ctypes
import json
import os
import sys
import time

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, REPO)
import torch  # noqa: E402

from pointcloudsegmentation_amd import _lib  # noqa: E402
from pointcloudsegmentation_amd.pointnet2 import Point2NetSA  # noqa: E402
from pointcloudsegmentation_amd.synth import make_batch  # noqa: E402


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--B", type=int, default=16)
    ap.add_argument("--N", type=int, default=8192)
    ap.add_argument("--iters", type=int, default=5)
    a = ap.parse_args()
    dev = torch.device("cuda", 0)
    torch.manual_seed(0)
    model = Point2NetSA(6, 64).train().to(dev)
    opt = torch.optim.Adam(model.parameters(), lr=1e-4)
    x = torch.from_numpy(make_batch(a.B, a.N, seed=5)).to(dev)
    r = torch.randn(a.B, a.N, 64, device=dev)

    def step():
        opt.zero_grad(set_to_none=True)
        loss = (model(x) * r).mean()
        loss.backward()
        opt.step()
        return loss

    for _ in range(2):
        step()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(a.iters):
        loss = step()
    torch.cuda.synchronize()
    wall = (time.perf_counter() - t0) / a.iters
    L = _lib.lib()
    L.p2n_profile_select(-1)
    L.p2n_profile_enable(1)
    for _ in range(a.iters):
        step()
    ms = (ctypes.c_float * 32)()
    cnt = (ctypes.c_int * 32)()
    L.p2n_profile_collect(32, ms, cnt)
    L.p2n_profile_enable(0)
    assert bool(torch.isfinite(loss))
    print(json.dumps({"metric": "Point2NetSA training step (fwd + bwd + Adam), B=%d x N=%d" % (a.B, a.N),
                      "ms_per_step_wall": 1e3 * wall, "points_per_s": a.B * a.N / wall, "loss": float(loss),
                      "peak_mem_GiB": torch.cuda.max_memory_allocated() / 2**30,
                      "library_kernel_ms_by_profile_id": {str(i): ms[i] / a.iters for i in range(32) if cnt[i]}}))


if __name__ == "__main__":
    main()
