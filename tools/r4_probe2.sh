#!/bin/bash
out=gpurun_out/r4_probe2.txt
: > $out
for cfg in cfg1 cfg0 cfg2; do
  python tools/oneshot_probe.py $cfg 21 >> $out 2>&1 || exit 1
  IRB_OFFLINE_DIRECT=0 python tools/oneshot_probe.py $cfg 21 >> $out 2>&1 || exit 1
done
