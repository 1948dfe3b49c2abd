#!/bin/bash
# one gpurun call: the one-shot tests, then the one-shot probe in every form (default staged, thread counts, pinned buffers, opt-in registration)
set -e
mkdir -p gpurun_out
timeout -k 10 500 python -m pytest tests/test_gpu_parity.py tests/test_gpu_cfg5.py -x -q -m gpu -k "convolve or one_shot or staged or non_finite or delta or raw" > gpurun_out/r5a_tests.txt 2>&1 || { tail -40 gpurun_out/r5a_tests.txt; exit 1; }
tail -3 gpurun_out/r5a_tests.txt
{
timeout -k 10 120 python tools/oneshot_probe.py all 21
IRB_OFFLINE_THREADS=1 timeout -k 10 60 python tools/oneshot_probe.py cfg1 21
IRB_OFFLINE_THREADS=2 timeout -k 10 60 python tools/oneshot_probe.py cfg1 21
IRB_OFFLINE_THREADS=4 timeout -k 10 60 python tools/oneshot_probe.py cfg1 21
timeout -k 10 120 python tools/oneshot_probe.py all 21 pinned
IRB_OFFLINE_DIRECT=1 timeout -k 10 60 python tools/oneshot_probe.py cfg1 21
} > gpurun_out/r5a_probe.txt 2>&1
cat gpurun_out/r5a_probe.txt
