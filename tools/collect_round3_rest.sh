tag=r03_g
here=/root/repo/tools
out=$GRAFT_REPO_ROOT/gpurun_out
bash $here/collect_train_pmc.sh $tag || exit 1
bash $here/collect_hier.sh $tag || exit 1
cd /tmp
timeout -k 10 60 /root/repo/tools/micro/fps_handoff 20000 > $out/${tag}_micro_fps_handoff.txt 2>&1 || exit 1
P2N_BENCH_REHEARSE=1 timeout -k 10 300 python3 -m torch.distributed.run --nnodes=1 --nproc-per-node 2 --master-addr 127.0.0.1 --master-port 29513 /root/repo/bench.py --gpus 2 --steps 20 --warmup 5 > $out/${tag}_bench_rehearse_2ranks_one_gpu.json 2> $out/${tag}_rehearse.err || exit 1
echo rest done
