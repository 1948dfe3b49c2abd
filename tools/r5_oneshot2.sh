#!/bin/bash
set -e
mkdir -p gpurun_out
timeout -k 10 300 python -m pytest tests/test_gpu_parity.py -x -q -m gpu -k "convolve or one_shot or staged or non_finite" > gpurun_out/r5b_tests.txt 2>&1 || { tail -40 gpurun_out/r5b_tests.txt; exit 1; }
tail -2 gpurun_out/r5b_tests.txt
{
timeout -k 10 120 python tools/oneshot_probe.py all 21
IRB_OFFLINE_THREADS=1 timeout -k 10 60 python tools/oneshot_probe.py cfg1 21
} > gpurun_out/r5b_probe.txt 2>&1
cat gpurun_out/r5b_probe.txt
timeout -k 10 200 bash tools/oneshot_timeline.sh cfg1
