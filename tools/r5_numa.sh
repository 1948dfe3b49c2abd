#!/bin/bash
mkdir -p gpurun_out
{
for n in /sys/devices/system/node/node*; do echo "$n: $(cat $n/cpulist)"; done
cat /sys/class/drm/card*/device/numa_node 2>/dev/null | head -3
which taskset numactl
for n in /sys/devices/system/node/node*; do
  l=$(cat $n/cpulist)
  echo "--- whole process on $n"
  taskset -c $l timeout -k 10 120 python tools/oneshot_probe.py cfg1 21 | head -1
  taskset -c $l timeout -k 10 120 python tools/oneshot_probe.py cfg1 21 | head -1
done
echo "--- unpinned x3"
for i in 1 2 3; do timeout -k 10 120 python tools/oneshot_probe.py cfg1 21 | head -1; done
} > gpurun_out/r5_numa.txt 2>&1
cat gpurun_out/r5_numa.txt
