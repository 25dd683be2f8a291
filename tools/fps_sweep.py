#!/usr/bin/env python3
"""FPS timing sweep: workgroups per cloud (P2N_FPS_G) x cloud size.  Prints us per sample."""
import os, sys, subprocess, json
REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if len(sys.argv) > 1 and sys.argv[1] == "child":
    sys.path.insert(0, REPO)
    import numpy as np, torch
    from pointcloudsegmentation_amd import ops
    from pointcloudsegmentation_amd.synth import make_batch
    B, N, M = int(sys.argv[2]), int(sys.argv[3]), int(sys.argv[4])
    xyz = torch.from_numpy(np.ascontiguousarray(make_batch(B, N, 7)[:, :, :3])).cuda()
    ops.farthest_point_sample(xyz, M); torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(3): ops.farthest_point_sample(xyz, M)
    e1.record(); torch.cuda.synchronize()
    ms = e0.elapsed_time(e1) / 3
    print(json.dumps({"B": B, "N": N, "M": M, "G": os.environ.get("P2N_FPS_G", "auto"), "ms": ms, "us_per_sample": 1e3 * ms / M,
                      "status": ops.fps_status("cuda")}))
else:
    for (B, N, M, gs) in [(16, 8192, 2048, ["auto", 2]), (1, 8192, 2048, ["auto", 2, 4]), (1, 2048, 512, ["auto"]),
                          (1, 40000, 10000, ["auto", 4, 5, 8, 10, 16, 32]),
                          (8, 40000, 2000, ["auto", 4, 8, 16]), (1, 150000, 4000, ["auto", 16, 32, 64]), (1, 1000000, 2000, ["auto", 64])]:
        for g in gs:
            env = dict(os.environ)
            if g != "auto": env["P2N_FPS_G"] = str(g)
            else: env.pop("P2N_FPS_G", None)
            r = subprocess.run([sys.executable, __file__, "child", str(B), str(N), str(M)], env=env, capture_output=True, text=True, timeout=300)
            print(r.stdout.strip() or r.stderr[-300:], flush=True)
