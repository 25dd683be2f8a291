#!/bin/bash
# Sweep of the LDS-staged gather / interpolation launch geometry (P2N_GL_* overrides) on the GPU box.
set -e
cd "$(dirname "$0")/.."
mkdir -p gpurun_out
out=gpurun_out/gather_sweep.log
: > $out
echo "off" >> $out
P2N_GL_OFF=1 timeout -k 10 120 python tools/bench_ops.py --no-lidar --only-gather >> $out 2>&1
echo "auto" >> $out
timeout -k 10 120 python tools/bench_ops.py --no-lidar --only-gather >> $out 2>&1
for ch in ${CHS:-1 2 4 8}; do for th in ${THS:-256 512 1024}; do for sp in ${SPS:-1 2}; do
  set -- $ch $th $sp
  echo "ch=$1 th=$2 sp=$3" >> $out
  P2N_GL_CH=$1 P2N_GL_THREADS=$2 P2N_GL_SPLIT=$3 timeout -k 10 120 python tools/bench_ops.py --no-lidar --only-gather >> $out 2>&1
done; done; done
