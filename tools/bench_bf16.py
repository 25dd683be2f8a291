#!/usr/bin/env python3
"""bf16 eval path alone (for rocprofv3 runs): B=16 x N=8192, prints kernel times from HIP events."""
import sys, ctypes, os, json
REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, REPO)
import numpy as np, torch
from pointcloudsegmentation_amd import _lib
from pointcloudsegmentation_amd.point2net import Point2Net
from pointcloudsegmentation_amd.synth import make_batch
L = _lib.lib()
w = dict(np.load(os.path.join(REPO, "tests", "golden", "weights.npz")))
m = Point2Net(); m.load_state_dict({k: torch.from_numpy(np.asarray(v)) for k, v in w.items()}); m = m.cuda().eval(); m.precision = "bf16"
x = torch.from_numpy(make_batch(16, 8192, 1)).cuda()
with torch.no_grad():
    for _ in range(3): m(x)
    torch.cuda.synchronize(); L.p2n_profile_enable(1)
    for _ in range(20): m(x)
    ms = (ctypes.c_float * 32)(); cnt = (ctypes.c_int * 32)(); L.p2n_profile_collect(32, ms, cnt)
print(json.dumps({"encoder_ms": ms[0] / cnt[0], "mid_ms": ms[1] / cnt[1], "head_ms": ms[2] / cnt[2]}))
