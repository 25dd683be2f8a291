#!/usr/bin/env python3
"""Phase clocks of the wave-cluster FPS kernel: needs tools/micro/libp2n_timing.so (the library built with
-DP2N_FPS_TIMING; see the FR_T macros in csrc/p2n_ops.hip).  usage: fps_timing.py B N M"""
import os, sys
REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, REPO)
os.environ.setdefault("P2N_FPS_ROUNDS", "2")
import numpy as np, torch
from pointcloudsegmentation_amd import _lib
_lib.LIB_PATH = os.path.join(REPO, "tools", "micro", "libp2n_timing.so")
from pointcloudsegmentation_amd import ops
from pointcloudsegmentation_amd.synth import make_batch
B, N, M = (int(a) for a in sys.argv[1:4])
xyz = torch.from_numpy(np.ascontiguousarray(make_batch(B, N, 7)[:, :, :3])).cuda()
for _ in range(2):
    ops.farthest_point_sample(xyz, M); torch.cuda.synchronize()
