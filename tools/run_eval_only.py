#!/usr/bin/env python3
"""N eval forwards of Point2Net at the headline shape in one precision, nothing else: the program rocprofv3 wraps for
kernel traces and PMC passes of one path (usage: run_eval_only.py <fp32|fp32x3|bf16> [steps])."""
import os, sys
REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, REPO)
import torch
import bench
from pointcloudsegmentation_amd.synth import make_batch

prec = sys.argv[1] if len(sys.argv) > 1 else "fp32"
steps = int(sys.argv[2]) if len(sys.argv) > 2 else 50
dev = torch.device("cuda", 0)
m = bench.load_model(dev)
m.precision = prec
x = torch.from_numpy(make_batch(16, 8192, 1000)).to(dev)
with torch.no_grad():
    for _ in range(steps):
        y = m(x)
torch.cuda.synchronize()
print(prec, steps, float(y.abs().max()))
