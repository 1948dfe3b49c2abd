#!/bin/bash
# fused kernel at fixed batch (2048 mono streams, B=256): time vs partitions -> per-launch fixed cost and per-partition slope
set -e
o=gpurun_out/p_sweep.txt
: > $o
for P in 47 94 188 375 750; do
  python tools/perf_probe.py few 2048 1 256 $((P*256)) 200 >> $o
done
cat $o
