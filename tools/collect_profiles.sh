#!/bin/bash
# Collect the evidence files for profiles/: run on the GPU box through gpurun.
#   tools/collect_profiles.sh <tag>     -> gpurun_out/<tag>_*
set -e
tag=${1:-rXX}
out=$GRAFT_REPO_ROOT/gpurun_out
mkdir -p $out
cd /tmp && export TMPDIR=/tmp
timeout -k 10 300 python3 /root/repo/bench.py > $out/${tag}_bench.json 2> $out/${tag}_bench.err
timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/prof_ks -o ks -- python3 /root/repo/bench.py --no-cpu --no-train > $out/${tag}_ks.log 2>&1
cp $(find /tmp/prof_ks -name "*kernel_stats.csv") $out/${tag}_kernel_stats_bench.csv
timeout -s KILL 120 rocprofv3 --pmc FETCH_SIZE --output-format csv -d /tmp/prof_f -o f -- python3 /root/repo/bench.py --no-cpu --no-fps --no-hier --no-train --no-scenes --steps 5 --warmup 1 > $out/${tag}_pmc_f.log 2>&1
timeout -s KILL 120 rocprofv3 --pmc WRITE_SIZE --output-format csv -d /tmp/prof_w -o w -- python3 /root/repo/bench.py --no-cpu --no-fps --no-hier --no-train --no-scenes --steps 5 --warmup 1 > $out/${tag}_pmc_w.log 2>&1
timeout -s KILL 120 rocprofv3 --pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES GRBM_GUI_ACTIVE --output-format csv -d /tmp/prof_m -o m -- python3 /root/repo/bench.py --no-cpu --no-fps --no-hier --no-train --no-scenes --steps 5 --warmup 1 > $out/${tag}_pmc_m.log 2>&1
for p in f w m; do
  d=$(dirname $(find /tmp/prof_$p -name "*counter_collection.csv"))
  cp $d/*counter_collection.csv /tmp/prof_$p/pmc_counter_collection.csv
  python3 /root/repo/profiles/summarize_pmc.py /tmp/prof_$p
done > $out/${tag}_pmc_bench.txt
timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/prof_tr -o tr -- python3 /root/repo/tools/bench_train.py --steps 5 > $out/${tag}_train_step.json 2> $out/${tag}_tr.err
cp $(find /tmp/prof_tr -name "*kernel_stats.csv") $out/${tag}_kernel_stats_train_step.csv
echo done
