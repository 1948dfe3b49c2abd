#!/bin/bash
set -e
mkdir -p gpurun_out
IRB_DEBUG=offline_pieces=2 timeout -k 10 200 bash tools/oneshot_timeline.sh cfg1 pinned
cp gpurun_out/oneshot_timeline_cfg1.txt gpurun_out/r5_tl_cfg1_pinned_2p.txt
IRB_DEBUG=offline_pieces=2 timeout -k 10 200 bash tools/oneshot_timeline.sh cfg1
cp gpurun_out/oneshot_timeline_cfg1.txt gpurun_out/r5_tl_cfg1_staged_2p.txt
IRB_DEBUG=offline_pieces=3 timeout -k 10 200 bash tools/oneshot_timeline.sh cfg2 pinned
cp gpurun_out/oneshot_timeline_cfg2.txt gpurun_out/r5_tl_cfg2_pinned_3p.txt
