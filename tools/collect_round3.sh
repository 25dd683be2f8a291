#!/bin/bash
# Round-3 evidence beyond collect_all.sh: the fp32x3 eval path (kernel trace + counters), the training step in fp32 and
# fp32x3 (kernel traces), counters of the training step's kernels incl. the loss kernels, the hierarchical sibling, the
# hand-off micro-benchmark and the two-rank rehearsal of bench.py.   tools/collect_round3.sh <tag>   (on the GPU box)
tag=${1:-rXX}
here=$(cd "$(dirname "$0")" && pwd)
out=$GRAFT_REPO_ROOT/gpurun_out
mkdir -p $out
bash $here/pmc_x3.sh $tag fp32x3 || exit 1
cd /tmp && export TMPDIR=/tmp
for mode in "" "--x3"; do
  name=fp32; [ -n "$mode" ] && name=fp32x3
  rm -rf /tmp/ptr_$name
  timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/ptr_$name -o tr -- python3 /root/repo/tools/bench_train.py --steps 7 $mode > $out/${tag}_train_step_${name}_profiled.json 2> /dev/null || exit 1
  cp $(find /tmp/ptr_$name -name "*kernel_stats.csv") $out/${tag}_kernel_stats_train_${name}.csv
  timeout -k 10 200 python3 /root/repo/tools/bench_train.py --steps 20 $mode 2> /dev/null | tail -1 > $out/${tag}_train_step_${name}.json || exit 1
done
P2N_X3_MASK=7 timeout -k 10 200 python3 /root/repo/tools/bench_train.py --steps 20 --x3 2> /dev/null | tail -1 > $out/${tag}_train_step_fp32x3_all_gemms.json || exit 1
bash $here/collect_train_pmc.sh $tag || exit 1
bash $here/collect_hier.sh $tag || exit 1
timeout -k 10 60 /root/repo/tools/micro/fps_handoff 20000 > $out/${tag}_micro_fps_handoff.txt 2>&1 || exit 1
P2N_BENCH_REHEARSE=1 timeout -k 10 300 python3 -m torch.distributed.run --nnodes=1 --nproc-per-node 2 --master-addr 127.0.0.1 --master-port 29513 /root/repo/bench.py --gpus 2 --steps 20 --warmup 5 > $out/${tag}_bench_rehearse_2ranks_one_gpu.json 2> $out/${tag}_rehearse.err || exit 1
echo round3 done
