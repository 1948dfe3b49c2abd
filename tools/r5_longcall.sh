#!/bin/bash
# one gpurun call: the one-grid long call (k_long_call) -- its tests, then whole-call times against the three-launch form
set -e
mkdir -p gpurun_out
timeout -k 10 600 python -m pytest tests/test_gpu_parity.py -q -x -k "matrix_core or long_call or one_grid or non_finite or random_shapes or packed_fma_tiled" > gpurun_out/r5_longcall_tests.txt 2>&1 || { tail -60 gpurun_out/r5_longcall_tests.txt; exit 1; }
tail -3 gpurun_out/r5_longcall_tests.txt
(for a in "256 96000 3750 1 400" "256 96000 1024 1 400" "128 480000 2048 2 100" "256 96000 4125 1 400" "64 6400 600 8 400"; do
   echo "== one grid: $a"; timeout -k 5 120 python tools/perf_probe.py offline $a
   echo "== three launches: $a"; IRB_DEBUG=long_call=0 timeout -k 5 120 python tools/perf_probe.py offline $a
 done) > gpurun_out/r5_longcall_times.txt 2>&1
cat gpurun_out/r5_longcall_times.txt
