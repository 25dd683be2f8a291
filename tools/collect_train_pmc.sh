#!/bin/bash
# Counters for the training-step kernels (tools/bench_train.py): MFMA busy, HBM bytes, LDS bank conflicts.
#   tools/collect_train_pmc.sh <tag>     (run on the GPU box through gpurun)
set -e
tag=${1:-rXX}
out=$GRAFT_REPO_ROOT/gpurun_out
mkdir -p $out
cd /tmp && export TMPDIR=/tmp
rm -rf /tmp/prof_t*
timeout -s KILL 150 rocprofv3 --pmc FETCH_SIZE --output-format csv -d /tmp/prof_tf -o f -- python3 /root/repo/tools/bench_train.py --steps 2 > $out/${tag}_train_pmc_f.log 2>&1
timeout -s KILL 150 rocprofv3 --pmc WRITE_SIZE --output-format csv -d /tmp/prof_tw -o w -- python3 /root/repo/tools/bench_train.py --steps 2 > $out/${tag}_train_pmc_w.log 2>&1
timeout -s KILL 150 rocprofv3 --pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES GRBM_GUI_ACTIVE --output-format csv -d /tmp/prof_tm -o m -- python3 /root/repo/tools/bench_train.py --steps 2 > $out/${tag}_train_pmc_m.log 2>&1
timeout -s KILL 150 rocprofv3 --pmc SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY --output-format csv -d /tmp/prof_tl -o l -- python3 /root/repo/tools/bench_train.py --steps 2 > $out/${tag}_train_pmc_l.log 2>&1
for p in tf tw tm tl; do
  d=$(dirname $(find /tmp/prof_$p -name "*counter_collection.csv"))
  cp $d/*counter_collection.csv /tmp/prof_$p/pmc_counter_collection.csv
  python3 /root/repo/profiles/summarize_pmc.py /tmp/prof_$p
done > $out/${tag}_pmc_train.txt
echo train pmc done
