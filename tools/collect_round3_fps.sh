#!/bin/bash
# Evidence for the FPS round kernels (wave clusters, round clusters): A/B sweep over kernel families and members per
# cloud, the large-cloud shapes, phase clocks (needs tools/micro/libp2n_timing.so), the hierarchical sibling and the
# two-rank rehearsal of bench.py.   tools/collect_round3_fps.sh <tag>   (on the GPU box)
tag=${1:-rXX}
here=$(cd "$(dirname "$0")" && pwd)
out=$GRAFT_REPO_ROOT/gpurun_out
mkdir -p $out
cd /tmp && export TMPDIR=/tmp
timeout -k 10 500 python3 -u $here/fps_rounds_ab.py > $out/${tag}_fps_rounds_ab.txt 2>&1 || exit 1
: > $out/${tag}_fps_large.txt
for shape in "1 1000000 16384" "1 300000 8192" "4 150000 4096"; do
  timeout -k 10 120 python3 -u $here/fps_big.py $shape 2> /dev/null >> $out/${tag}_fps_large.txt || exit 1
  P2N_FPS_ROUNDS=-1 timeout -k 10 120 python3 -u $here/fps_big.py $shape 2> /dev/null >> $out/${tag}_fps_large.txt || exit 1
done
if [ -f $here/micro/libp2n_timing.so ]; then
  : > $out/${tag}_fps_phase_clocks.txt
  for shape in "1 40000 10000" "16 8192 2048" "8 40000 2000"; do
    echo "B N M = $shape" >> $out/${tag}_fps_phase_clocks.txt
    timeout -k 10 120 python3 -u $here/fps_timing.py $shape 2> /dev/null | tail -1 >> $out/${tag}_fps_phase_clocks.txt || exit 1
  done
fi
bash $here/collect_hier.sh $tag || exit 1
cd /tmp
P2N_BENCH_REHEARSE=1 timeout -k 10 300 python3 -m torch.distributed.run --nnodes=1 --nproc-per-node 2 --master-addr 127.0.0.1 --master-port 29513 /root/repo/bench.py --gpus 2 --steps 20 --warmup 5 2> $out/${tag}_rehearse.err | grep '^{"metric"' > $out/${tag}_bench_rehearse_2ranks_one_gpu.json || exit 1
echo fps evidence done
