#!/bin/bash
# Quick A/B of the clustering passes: parity tests, the two synthetic orders, the scene (run through gpurun).
set -e
cd $GRAFT_REPO_ROOT
timeout -k 10 300 python -u -m pytest tests/test_cluster_gpu.py -x -q -m gpu --timeout 200 --timeout-method thread 2>&1 | tail -1
timeout -k 10 100 python3 tools/bench_cluster.py --order runs --cpu-n 0 > gpurun_out/x_cluster_runs.json
timeout -k 10 100 python3 tools/bench_cluster.py --order random --cpu-n 0 > gpurun_out/x_cluster_random.json
timeout -k 10 200 python3 tools/bench_inference.py > gpurun_out/x_inference.json 2> gpurun_out/x_inference.err
python3 - <<'P'
import json
for f in ("runs", "random"):
    d = json.load(open(f"gpurun_out/x_cluster_{f}.json"))
    print(f, round(d["ms"], 2), {k: round(v, 2) for k, v in d["kernels_ms"].items()}, d["stats"]["unions"])
d = json.load(open("gpurun_out/x_inference.json"))
print("scene", round(d["end_to_end_ms_host_in_host_out"], 2), "model-emb cluster", round(d["cluster_ms_model_embeddings"], 2),
      "33-cluster", round(d["cluster_ms_33_cluster_embeddings"], 2))
P
