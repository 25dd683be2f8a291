"""Experiment: does the loss's neighbour search (VALU-bound, depends on coordinates only) hide behind the train-mode
forward (MFMA-bound) when it runs on a second stream?  Prints ms for forward alone, kNN alone, both serial, both
overlapped."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from pointcloudsegmentation_amd import ops
from pointcloudsegmentation_amd.synth import make_batch
from src.models.point2net import Point2Net

torch.manual_seed(0)
m = Point2Net(6, 64).cuda().train()
x = torch.from_numpy(make_batch(32, 8192, seed=200)).cuda()
pts = x[:, :, :3].contiguous()
side = torch.cuda.Stream()


def fwd():
    with torch.no_grad():
        return m(x)


def knn():
    return ops.knn_self(pts, 21, return_dist=False)


def both_serial():
    fwd(); knn()


def both_overlap():
    cur = torch.cuda.current_stream()
    side.wait_stream(cur)
    with torch.cuda.stream(side):
        knn()
    fwd()
    cur.wait_stream(side)


def timeit(f, n=20):
    for _ in range(3):
        f()
    torch.cuda.synchronize()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(n):
        f()
    b.record()
    torch.cuda.synchronize()
    return a.elapsed_time(b) / n


for name, f in (("forward", fwd), ("knn", knn), ("serial", both_serial), ("overlap", both_overlap)):
    print("%-8s %.3f ms" % (name, timeit(f)), flush=True)
