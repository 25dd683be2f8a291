#!/bin/bash
# Operator roofline evidence: event-timed table + rocprofv3 kernel trace of the same command, and
# the HBM counters of the gather / interpolation kernels.  Run on the GPU box through gpurun.
#   tools/collect_ops_profile.sh <tag>     -> gpurun_out/<tag>_ops_*
set -e
tag=${1:-rXX}
out=$GRAFT_REPO_ROOT/gpurun_out
mkdir -p $out
cd /tmp && export TMPDIR=/tmp
timeout -k 10 300 python3 /root/repo/tools/bench_ops.py > $out/${tag}_ops_roofline.json 2> $out/${tag}_ops.err
timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/prof_ops -o ops -- python3 /root/repo/tools/bench_ops.py --no-lidar > $out/${tag}_ops_ks.log 2>&1
cp $(find /tmp/prof_ops -name "*kernel_stats.csv") $out/${tag}_kernel_stats_ops.csv
timeout -s KILL 120 rocprofv3 --pmc FETCH_SIZE --output-format csv -d /tmp/prof_of -o f -- python3 /root/repo/tools/bench_ops.py --no-lidar --only-gather --iters 3 > $out/${tag}_ops_pmc_f.log 2>&1
timeout -s KILL 120 rocprofv3 --pmc WRITE_SIZE --output-format csv -d /tmp/prof_ow -o w -- python3 /root/repo/tools/bench_ops.py --no-lidar --only-gather --iters 3 > $out/${tag}_ops_pmc_w.log 2>&1
for p in of ow; do
  d=$(dirname $(find /tmp/prof_$p -name "*counter_collection.csv"))
  cp $d/*counter_collection.csv /tmp/prof_$p/pmc_counter_collection.csv
  python3 /root/repo/profiles/summarize_pmc.py /tmp/prof_$p
done > $out/${tag}_pmc_ops.txt
echo done
