#!/bin/bash
set -e
mkdir -p gpurun_out
timeout -k 10 600 python -m pytest tests -x -q -m gpu > gpurun_out/r5f_tests.txt 2>&1 || { tail -60 gpurun_out/r5f_tests.txt; exit 1; }
tail -2 gpurun_out/r5f_tests.txt
{
timeout -k 10 120 python tools/oneshot_probe.py all 21
timeout -k 10 120 python tools/oneshot_probe.py all 21 pinned
IRB_OFFLINE_DIRECT=1 timeout -k 10 120 python tools/oneshot_probe.py all 21
} > gpurun_out/r5f_probe.txt 2>&1
cat gpurun_out/r5f_probe.txt
