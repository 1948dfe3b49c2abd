#!/bin/bash
# Batch-size profile of the one-block step (configs[1] geometry, device-resident): per stream count, the default path and
# the A/B arms given as extra arguments ("VAR=val VAR=val" per arm).  Output: gpurun_out/batch_sweep.txt
out=gpurun_out/batch_sweep.txt
: > $out
for I in ${SWEEP_I:-16 32 64 128 192 256 320 384 448 512 768 1024 2048}; do
  n=$(( I >= 1024 ? 150 : 400 ))
  python tools/perf_probe.py few $I 1 256 96000 $n >> $out 2>&1 || exit 1
  for arm in "$@"; do env $arm python tools/perf_probe.py few $I 1 256 96000 $n >> $out 2>&1 || exit 1; done
done
