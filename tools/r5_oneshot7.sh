#!/bin/bash
set -e
mkdir -p gpurun_out
timeout -k 10 300 python -m pytest tests/test_gpu_parity.py tests/test_gpu_cfg5.py -x -q -m gpu -k "convolve or one_shot or staged or non_finite" > gpurun_out/r5g_tests.txt 2>&1 || { tail -60 gpurun_out/r5g_tests.txt; exit 1; }
tail -2 gpurun_out/r5g_tests.txt
{
timeout -k 10 120 python tools/oneshot_probe.py all 21
IRB_DEBUG=offline_pieces=1 timeout -k 10 120 python tools/oneshot_probe.py cfg1 21
IRB_DEBUG=offline_pieces=3 timeout -k 10 120 python tools/oneshot_probe.py cfg1 21
IRB_OFFLINE_THREADS=4 timeout -k 10 120 python tools/oneshot_probe.py cfg1 21
IRB_OFFLINE_THREADS=2 timeout -k 10 120 python tools/oneshot_probe.py cfg1 21
timeout -k 10 120 python tools/oneshot_probe.py cfg1 21 pinned
} > gpurun_out/r5g_probe.txt 2>&1
grep -v "^    " gpurun_out/r5g_probe.txt
timeout -k 10 200 bash tools/oneshot_timeline.sh cfg1
