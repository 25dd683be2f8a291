#!/bin/bash
# Counter passes over the fp32x3 eval path alone (tools/run_eval_only.py): where the cycles of its two MFMA kernels go.
#   tools/pmc_x3.sh <tag> [precision]   -> gpurun_out/<tag>_pmc_x3.txt, <tag>_kernel_stats_x3.csv
tag=${1:-rXX}
prec=${2:-fp32x3}
out=$GRAFT_REPO_ROOT/gpurun_out
mkdir -p $out
cd /tmp && export TMPDIR=/tmp
timeout -k 10 200 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/px_ks -o ks -- python3 /root/repo/tools/run_eval_only.py $prec 200 > $out/${tag}_x3_ks.log 2>&1 || exit 1
cp $(find /tmp/px_ks -name "*kernel_stats.csv") $out/${tag}_kernel_stats_${prec}.csv
timeout -s KILL 120 rocprofv3 --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_VALU_MFMA_BUSY_CYCLES SQ_WAIT_INST_LDS GRBM_GUI_ACTIVE --output-format csv -d /tmp/px_a -o a -- python3 /root/repo/tools/run_eval_only.py $prec 5 > $out/${tag}_x3_pmc_a.log 2>&1 || exit 1
timeout -s KILL 120 rocprofv3 --pmc SQ_INSTS_VALU SQ_INSTS_MFMA SQ_INSTS_LDS SQ_INSTS_SALU SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_LDS_BANK_CONFLICT SQ_ACTIVE_INST_LDS --output-format csv -d /tmp/px_b -o b -- python3 /root/repo/tools/run_eval_only.py $prec 5 > $out/${tag}_x3_pmc_b.log 2>&1 || exit 1
for p in a b; do
  d=$(dirname $(find /tmp/px_$p -name "*counter_collection.csv"))
  cp $d/*counter_collection.csv /tmp/px_$p/pmc_counter_collection.csv
  python3 /root/repo/profiles/summarize_pmc.py /tmp/px_$p
done > $out/${tag}_pmc_${prec}.txt
echo done
