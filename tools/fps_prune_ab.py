#!/usr/bin/env python3
"""FPS, dense vs bucket-pruned single-workgroup kernel (P2N_FPS_PRUNE=1 switches the pruned one off): us per sample."""
import os, sys, subprocess, json
REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if len(sys.argv) > 1 and sys.argv[1] == "child":
    sys.path.insert(0, REPO)
    import numpy as np, torch
    from pointcloudsegmentation_amd import ops
    from pointcloudsegmentation_amd.synth import make_batch
    B, N, M = int(sys.argv[2]), int(sys.argv[3]), int(sys.argv[4])
    xyz = torch.from_numpy(np.ascontiguousarray(make_batch(B, N, 7)[:, :, :3])).cuda()
    a = ops.farthest_point_sample(xyz, M); torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(5): ops.farthest_point_sample(xyz, M)
    e1.record(); torch.cuda.synchronize()
    ms = e0.elapsed_time(e1) / 5
    print(json.dumps({"B": B, "N": N, "M": M, "prune": os.environ.get("P2N_FPS_PRUNE", "auto"), "pw": os.environ.get("P2N_FPS_PW", "0"), "ms": round(ms, 4),
                      "us_per_sample": round(1e3 * ms / M, 4), "checksum": int(a.long().sum())}))
else:
    for (B, N, M) in [(16, 8192, 2048), (16, 2048, 512), (16, 4096, 1024), (16, 6144, 1536), (8, 12288, 3072), (8, 10000, 2500)]:
        for pr, pw in (("1", "0"), ("2", "0"), ("2", "8"), ("2", "4")):
            env = dict(os.environ, P2N_FPS_PRUNE=pr, P2N_FPS_PW=pw)
            r = subprocess.run([sys.executable, __file__, "child", str(B), str(N), str(M)], env=env, capture_output=True, text=True, timeout=120)
            print(r.stdout.strip() or r.stderr[-300:], flush=True)
