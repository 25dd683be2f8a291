#!/usr/bin/env python3
"""FPS: several samples per exchange round (default for clusters) vs one (P2N_FPS_ROUNDS=-1), over workgroups per cloud.
Prints ms, us per sample and a checksum of the indices (must be equal across variants of one shape)."""
import os, sys, subprocess, json
REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if len(sys.argv) > 1 and sys.argv[1] == "child":
    sys.path.insert(0, REPO)
    import numpy as np, torch
    from pointcloudsegmentation_amd import ops
    from pointcloudsegmentation_amd.synth import make_batch
    B, N, M = int(sys.argv[2]), int(sys.argv[3]), int(sys.argv[4])
    xyz = torch.from_numpy(np.ascontiguousarray(make_batch(B, N, 7)[:, :, :3])).cuda()
    idx = ops.farthest_point_sample(xyz, M); torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(3): ops.farthest_point_sample(xyz, M)
    e1.record(); torch.cuda.synchronize()
    ms = e0.elapsed_time(e1) / 3
    w = torch.arange(1, M + 1, device="cuda", dtype=torch.int64)
    chk = int(((idx.long() + 1) * w).sum().item() % 1000000007)
    print(json.dumps({"B": B, "N": N, "M": M, "G": os.environ.get("P2N_FPS_G", "auto"), "rounds": os.environ.get("P2N_FPS_ROUNDS", "on"),
                      "prune": os.environ.get("P2N_FPS_PRUNE", "auto"), "ms": round(ms, 4), "us_per_sample": round(1e3 * ms / M, 4),
                      "checksum": chk, "status": ops.fps_status("cuda")}))
else:
    shapes = [(1, 40000, 10000, ["auto", 16, 32], "0"), (8, 40000, 2000, ["auto", 32], "0"),
              (1, 150000, 4000, ["auto", 32], "0"),
              (16, 8192, 2048, ["auto", 1, 16, 32], "1"), (1, 8192, 2048, ["auto", 1, 32], "1"), (16, 2048, 512, ["auto", 1, 16], "1"),
              (32, 8192, 2048, ["auto"], "1"), (1, 4096, 1024, ["auto", 1], "1")]
    for (B, N, M, gs, noprune) in shapes:
        for g in gs:
            for rounds in ("on", "-1", "2"):
                if rounds == "-1" and g not in ("auto", 1, 32): continue
                if rounds == "2" and g == 1: continue
                env = dict(os.environ)
                for k in ("P2N_FPS_G", "P2N_FPS_ROUNDS", "P2N_FPS_PRUNE"): env.pop(k, None)
                if g != "auto": env["P2N_FPS_G"] = str(g)
                if rounds != "on": env["P2N_FPS_ROUNDS"] = rounds
                if noprune == "1": env["P2N_FPS_PRUNE"] = "1"
                r = subprocess.run([sys.executable, __file__, "child", str(B), str(N), str(M)], env=env, capture_output=True, text=True, timeout=300)
                print(r.stdout.strip() or r.stderr[-300:], flush=True)
