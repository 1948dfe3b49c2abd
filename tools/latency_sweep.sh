#!/bin/bash
# A/B of the one-block step on the GPU box: round-1 kernels (IRB_COLS=0: fused from 128 streams, else k_step_split)
# vs the column-sliced kernel and its knobs; then the batch-size profile.  Output: gpurun_out/latency_sweep.txt,
# gpurun_out/batch_sweep.txt
set -e
out=gpurun_out/latency_sweep.txt
: > $out
run() { env "$@" python tools/perf_probe.py few $ARGS >> $out 2>&1; }
for ARGS in "1 1 256 96000 3000" "1 2 128 480000 3000" "1 2 256 2048 3000" "8 1 256 96000 2000" "32 1 256 96000 1000"; do
  echo "== $ARGS" >> $out
  run IRB_COLS=0
  run IRB_COLS=1
  for w in 4 16 64; do for ps in 1 2 4 8 16; do run IRB_COLW=$w IRB_PSPLITS=$ps IRB_FUSED_ROWS=0; done; done
  run IRB_COLS_HANDOFF=0 IRB_FUSED_ROWS=0
  run IRB_COLS_HANDOFF=1 IRB_FUSED_ROWS=0
  run IRB_COLS_HANDOFF=2 IRB_FUSED_ROWS=0
done
out=gpurun_out/batch_sweep.txt
: > $out
echo "== batch profile (configs[1] geometry, one block per call): round-1 heuristic, default, forced cols, forced fused" >> $out
for I in 16 32 64 128 192 256 384 512 1024 2048; do
  ARGS="$I 1 256 96000 300"
  run IRB_COLS=0
  run IRB_COLS=1
  run IRB_FUSED_FROM=100000
  run IRB_FUSED_FROM=1
done
