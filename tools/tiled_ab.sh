#!/bin/bash
# A/B of the time-tiled MAC kernels on the GPU box (IRB_TILED_KERNEL: 0 = round-2 k_cmac_tiled<N,8>, 1 = k_cmac_tiled_pk) over the
# grid target of the partition splits (IRB_TILED_WGS).  Output: gpurun_out/tiled_ab.txt
out=gpurun_out/tiled_ab.txt
: > $out
for shape in "256 96000 3750 1" "128 480000 2048 2" "256 96000 64 64" "256 96000 8 2048" "256 96000 16 2048"; do
  echo "== offline $shape" >> $out
  for k in 0 1; do
    for w in 256 512 768 1024; do
      echo -n "kernel=$k wgs=$w  " >> $out
      IRB_TILED_KERNEL=$k IRB_TILED_WGS=$w python tools/perf_probe.py offline $shape >> $out 2>&1 || exit 1
    done
  done
done
