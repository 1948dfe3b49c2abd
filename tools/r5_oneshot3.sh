#!/bin/bash
set -e
mkdir -p gpurun_out
timeout -k 10 300 python -m pytest tests/test_gpu_parity.py tests/test_gpu_cfg5.py -x -q -m gpu -k "convolve or one_shot or staged or non_finite or resident" > gpurun_out/r5c_tests.txt 2>&1 || { tail -60 gpurun_out/r5c_tests.txt; exit 1; }
tail -2 gpurun_out/r5c_tests.txt
{
for np in 1 2 3 4; do
IRB_DEBUG=offline_pieces=$np timeout -k 10 120 python tools/oneshot_probe.py cfg1 21
IRB_DEBUG=offline_pieces=$np timeout -k 10 120 python tools/oneshot_probe.py cfg1 21 pinned
done
for np in 1 2 3 4 6; do
IRB_DEBUG=offline_pieces=$np timeout -k 10 120 python tools/oneshot_probe.py cfg2 15
IRB_DEBUG=offline_pieces=$np timeout -k 10 120 python tools/oneshot_probe.py cfg2 15 pinned
done
timeout -k 10 120 python tools/oneshot_probe.py cfg0 21
timeout -k 10 120 python tools/oneshot_probe.py cfg0 21 pinned
} > gpurun_out/r5c_probe.txt 2>&1
grep -v "^    " gpurun_out/r5c_probe.txt
