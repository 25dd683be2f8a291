#!/bin/bash
# Everything profiles/ cites for one round state: headline bench + kernel trace + counters + training step
# (collect_profiles.sh), operator table (collect_ops_profile.sh), clustering and scene inference.
#   tools/collect_all.sh <tag>    (run on the GPU box through gpurun)
set -e
tag=${1:-rXX}
here=$(cd "$(dirname "$0")" && pwd)
bash $here/collect_profiles.sh $tag
bash $here/collect_ops_profile.sh $tag
out=$GRAFT_REPO_ROOT/gpurun_out
cd /tmp
timeout -k 10 200 python3 /root/repo/tools/bench_cluster.py --order random > $out/${tag}_cluster_random.json 2> $out/${tag}_cluster.err
timeout -k 10 200 python3 /root/repo/tools/bench_cluster.py --order runs --cpu-n 0 > $out/${tag}_cluster_runs.json 2>> $out/${tag}_cluster.err
timeout -k 10 200 python3 /root/repo/tools/bench_inference.py > $out/${tag}_inference.json 2> $out/${tag}_inference.err
timeout -k 10 200 python3 /root/repo/tools/fps_sweep.py > $out/${tag}_fps_sweep.txt 2>&1
echo all done
