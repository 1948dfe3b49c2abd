#!/bin/bash
# round-4 working probe: matrix-core MAC stamps + offline calls + the one-shot function over its knobs.  Output: gpurun_out/r4_probe.txt
out=gpurun_out/r4_probe.txt
: > $out
for a in "3750 1 375" "3750 2 375" "4096 2 375" "1024 4 375" "1024 8 375" "2048 4 3750"; do tools/mfma_stamps $a >> $out 2>&1 || exit 1; done
for shape in "256 96000 3750 1 200" "256 96000 4125 1 200" "256 96000 1024 1 200" "128 480000 2048 2 50" "256 96000 512 4 100" "512 96000 2048 2 50"; do
  for res in 2 1; do
    echo -n "res=$res " >> $out
    IRB_TILED_MFMA_RES=$res python tools/perf_probe.py offline $shape >> $out 2>&1 || exit 1
  done
done
for cfg in cfg1 cfg0; do
  for helper in 1 0; do
    for chunk in 512 1024 2048; do
      IRB_OFFLINE_HELPER=$helper IRB_OFFLINE_CHUNK=$chunk python tools/oneshot_probe.py $cfg 21 >> $out 2>&1 || exit 1
    done
  done
done
python tools/oneshot_probe.py cfg2 11 >> $out 2>&1 || exit 1
