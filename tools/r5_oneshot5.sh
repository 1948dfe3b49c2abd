#!/bin/bash
set -e
mkdir -p gpurun_out
timeout -k 10 300 python -m pytest tests/test_gpu_parity.py tests/test_gpu_cfg5.py -x -q -m gpu -k "convolve or one_shot or staged or non_finite or resident" > gpurun_out/r5e_tests.txt 2>&1 || { tail -60 gpurun_out/r5e_tests.txt; exit 1; }
tail -2 gpurun_out/r5e_tests.txt
IRB_DEBUG=offline_overlap_mac=1,offline_pieces=4 timeout -k 10 200 bash tools/oneshot_timeline.sh cfg2 pinned
{
for np in 1 2; do
IRB_DEBUG=offline_overlap_mac=1,offline_pieces=$np timeout -k 10 120 python tools/oneshot_probe.py cfg2 15 pinned
done
IRB_DEBUG=offline_overlap_mac=1,offline_pieces=4,offline_cu_masks=0 timeout -k 10 120 python tools/oneshot_probe.py cfg2 15 pinned
IRB_DEBUG=offline_overlap_mac=0,offline_pieces=1 timeout -k 10 120 python tools/oneshot_probe.py cfg2 15 pinned
for np in 3 4 6; do
IRB_DEBUG=offline_overlap_mac=1,offline_pieces=$np timeout -k 10 120 python tools/oneshot_probe.py cfg2 15
done
IRB_DEBUG=offline_overlap_mac=0,offline_pieces=1 timeout -k 10 120 python tools/oneshot_probe.py cfg2 15
} > gpurun_out/r5e_probe.txt 2>&1
grep -v "^    " gpurun_out/r5e_probe.txt
