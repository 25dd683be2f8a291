#!/usr/bin/env python3
"""Why is the hierarchical sibling's lone forward slow inside bench.py?  Runs bench_hier.run after pieces of the bench."""
import os, sys, json
REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, REPO); sys.path.insert(0, os.path.join(REPO, "tools"))
import torch
import bench, bench_hier
dev = torch.device("cuda", 0)
def hier(tag):
    r = bench_hier.run(16, 8192, iters=20)
    print(tag, round(r["ms_per_forward_wall"], 3), round(r["ms_per_forward_in_a_stream_of_8_batches"], 3), flush=True)
what = sys.argv[1]
hier("fresh")
if what == "fps":
    print(bench.fps_leg(dev)["ms_1m_to_16384"]); hier("after fps leg")
elif what == "train":
    print({k: v.get("ms_per_step") for k, v in bench.train_leg(dev, steps=4).items() if isinstance(v, dict)}); hier("after train leg")
    torch.cuda.empty_cache(); hier("after empty_cache")
elif what == "streams":
    ss = [torch.cuda.Stream() for _ in range(12)]
    for s in ss:
        with torch.cuda.stream(s): torch.zeros(10, device=dev).add_(1)
    torch.cuda.synchronize(); hier("after 12 more streams")
