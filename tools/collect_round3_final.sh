#!/bin/bash
# Final evidence set of round 3 (run through gpurun):  tools/collect_round3_final.sh <tag>
# GPU suite, default bench.py, the clustering legs (both orders, kernel trace, repeat runs, big sizes), scene bench.
set -e
tag=$1
cd $GRAFT_REPO_ROOT
out=gpurun_out
timeout -k 10 800 python -u -m pytest tests -x -q -m gpu --timeout 300 --timeout-method thread > $out/${tag}_gpu_tests.log 2>&1
tail -1 $out/${tag}_gpu_tests.log
timeout -k 10 600 python3 bench.py > $out/${tag}_bench.json 2> $out/${tag}_bench.err
timeout -k 10 100 python3 tools/bench_cluster.py --order runs --cpu-n 0 > $out/${tag}_cluster_runs.json
timeout -k 10 100 python3 tools/bench_cluster.py --order random --cpu-n 0 > $out/${tag}_cluster_random.json
timeout -k 10 200 python3 tools/bench_inference.py > $out/${tag}_inference.json 2> $out/${tag}_inference.err
timeout -k 10 200 python3 tools/cluster_repeat.py > $out/${tag}_cluster_repeat.txt 2>&1
timeout -k 10 200 python3 tools/cluster_big.py > $out/${tag}_cluster_big.txt 2>&1
bash tools/cluster_trace.sh /root/repo/tools/bench_cluster.py --order random --cpu-n 0 > $out/${tag}_cluster_kernels.txt
echo done
