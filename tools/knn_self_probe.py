"""Timing of the self kNN (Morton-ordered) against the streaming kNN on the train-step shape."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from pointcloudsegmentation_amd import ops

def timed(f, iters=20):
    for _ in range(3): f()
    torch.cuda.synchronize()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(iters): f()
    b.record(); torch.cuda.synchronize()
    return a.elapsed_time(b) / iters

for (B, N, k) in ((32, 8192, 17), (8, 8192, 17), (32, 2048, 17), (4, 16384, 33), (32, 8192, 33)):
    g = torch.Generator(device="cuda").manual_seed(1)
    x = torch.rand(B, N, 3, device="cuda", generator=g)
    xq = x.clone()   # separate query tensor: the streaming kernel
    i0 = ops.knn(xq, x, k, return_dist=False)
    i1 = ops.knn_self(x, k, return_dist=False)
    same = bool((i0 == i1).all())
    t0 = timed(lambda: ops.knn(xq, x, k, return_dist=False))
    t1 = timed(lambda: ops.knn_self(x, k, return_dist=False))
    print(f"B={B} N={N} k={k}: streaming {t0:.3f} ms, self {t1:.3f} ms, equal={same}", flush=True)
