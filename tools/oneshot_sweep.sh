#!/bin/bash
# irb_convolve_periodic from pageable host buffers over the pipeline's knobs: chunk size x {staged copies, zero copy} x
# {unpacking thread, one thread}.  Output: gpurun_out/oneshot_sweep.txt
out=gpurun_out/oneshot_sweep.txt
: > $out
for cfg in cfg1 cfg0; do
  for mode in 0 1; do
    for helper in 1 0; do
      for chunk in 256 512 1024 2048 16384; do
        IRB_OFFLINE_MODE=$mode IRB_OFFLINE_HELPER=$helper IRB_OFFLINE_CHUNK=$chunk python tools/oneshot_probe.py $cfg 21 >> $out 2>&1 || exit 1
      done
    done
  done
done
IRB_OFFLINE_CHUNK=1024 python tools/oneshot_probe.py cfg2 11 >> $out 2>&1 || exit 1
