#!/bin/bash
# A/B of the matrix-core tiled MAC (k_cmac_mfma; IRB_DEBUG=mfma_min_t=0 switches it off) against the packed-FMA kernel, over the
# workgroup target of its partition splits (IRB_DEBUG mfma_wgs), whole calls timed with events.  Output: gpurun_out/mfma_ab.txt
out=gpurun_out/mfma_ab.txt
: > $out
for shape in "256 96000 3750 1 200" "128 480000 2048 2 50" "256 96000 1024 1 200" "256 96000 512 4 100" "256 96000 1024 16 20" "512 96000 2048 2 50" "256 96000 256 8 100" "256 96000 128 16 100" "64 24000 4096 1 100" "1024 96000 1024 2 50"; do
  echo "== offline $shape" >> $out
  echo -n "packed-FMA kernel    " >> $out
  IRB_DEBUG=mfma_min_t=0 python tools/perf_probe.py offline $shape >> $out 2>&1 || exit 1
  for w in 128 256 512; do
    echo -n "mfma wgs=$w " >> $out
    IRB_DEBUG=mfma_min_t=2,mfma_wgs=$w python tools/perf_probe.py offline $shape >> $out 2>&1 || exit 1
  done
done
