#!/bin/bash
# A/B of the one-block latency path on the GPU box: configs[1] single stream and configs[2] stereo, round-1 split
# kernel vs the column-sliced kernel with its knobs.  Output: gpurun_out/latency_sweep.txt
set -e
out=gpurun_out/latency_sweep.txt
: > $out
run() { env "$@" python tools/perf_probe.py few $ARGS >> $out 2>&1; }
for ARGS in "1 1 256 96000 3000" "1 2 128 480000 3000" "1 2 256 2048 3000" "8 1 256 96000 2000" "32 1 256 96000 1000"; do
  echo "== $ARGS" >> $out
  run IRB_COLS=0
  run IRB_COLS=1
  for w in 4 16 64; do for ps in 1 2 4 8 16; do run IRB_COLW=$w IRB_PSPLITS=$ps; done; done
  run IRB_COLS_SC1=0
done
echo "== batch profile (configs[1] geometry, one block per call)" >> $out
for I in 64 128 192 256 384 512 1024 2048; do
  ARGS="$I 1 256 96000 300"
  run IRB_COLS=0
  run IRB_COLS=1
  run IRB_FUSED_FROM=100000 IRB_PSPLITS=2
  run IRB_FUSED_FROM=100000 IRB_PSPLITS=4
  run IRB_FUSED_FROM=100000 IRB_PSPLITS=8
done
