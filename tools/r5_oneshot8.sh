#!/bin/bash
set -e
mkdir -p gpurun_out
{
timeout -k 10 120 python tools/oneshot_probe.py cfg1 21
IRB_DEBUG=offline_in_one_launch=0 timeout -k 10 120 python tools/oneshot_probe.py cfg1 21
IRB_DEBUG=offline_pieces=1 timeout -k 10 120 python tools/oneshot_probe.py cfg1 21
IRB_DEBUG=offline_pieces=1,offline_in_one_launch=0 timeout -k 10 120 python tools/oneshot_probe.py cfg1 21
timeout -k 10 120 python tools/oneshot_probe.py cfg1 21 pinned
} > gpurun_out/r5h_probe.txt 2>&1
cat gpurun_out/r5h_probe.txt
IRB_DEBUG=offline_pieces=1 timeout -k 10 200 bash tools/oneshot_timeline.sh cfg1
