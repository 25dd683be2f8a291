#!/bin/bash
# Hierarchical forward (Point2NetSA, B=16 x N=8192): bench line, kernel trace, MFMA/HBM counters.
#   tools/collect_hier.sh <tag> [--unfused]     (run on the GPU box through gpurun)
set -e
tag=${1:-rXX}; shift || true
out=$GRAFT_REPO_ROOT/gpurun_out
mkdir -p $out
cd /tmp && export TMPDIR=/tmp
timeout -k 10 200 python3 /root/repo/tools/bench_hier.py "$@" > $out/${tag}_hier.json 2> $out/${tag}_hier.err
rm -rf /tmp/prof_h*
timeout -k 10 200 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/prof_hk -o hk -- python3 /root/repo/tools/bench_hier.py --iters 10 "$@" > $out/${tag}_hier_ks.log 2>&1
cp $(find /tmp/prof_hk -name "*kernel_stats.csv") $out/${tag}_kernel_stats_hier.csv
timeout -s KILL 120 rocprofv3 --pmc FETCH_SIZE --output-format csv -d /tmp/prof_hf -o f -- python3 /root/repo/tools/bench_hier.py --iters 3 "$@" > $out/${tag}_hier_pmc_f.log 2>&1
timeout -s KILL 120 rocprofv3 --pmc WRITE_SIZE --output-format csv -d /tmp/prof_hw -o w -- python3 /root/repo/tools/bench_hier.py --iters 3 "$@" > $out/${tag}_hier_pmc_w.log 2>&1
timeout -s KILL 120 rocprofv3 --pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES GRBM_GUI_ACTIVE --output-format csv -d /tmp/prof_hm -o m -- python3 /root/repo/tools/bench_hier.py --iters 3 "$@" > $out/${tag}_hier_pmc_m.log 2>&1
for p in hf hw hm; do
  d=$(dirname $(find /tmp/prof_$p -name "*counter_collection.csv"))
  cp $d/*counter_collection.csv /tmp/prof_$p/pmc_counter_collection.csv
  python3 /root/repo/profiles/summarize_pmc.py /tmp/prof_$p
done > $out/${tag}_pmc_hier.txt
echo hier done
