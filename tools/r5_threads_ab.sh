#!/bin/bash
# staged one-shot at configs[1] / configs[0] / configs[2]: host threads 2..6, five alternating passes on one box (C ABI median of 21 calls each)
for pass in 1 2 3 4 5; do
  for t in 3 4 5 2 6; do
    IRB_OFFLINE_THREADS=$t python3 tools/oneshot_probe.py cfg1 21 | grep -v "^  " | sed "s/^/threads=$t /"
  done
done
for t in 3 4 5; do
  IRB_OFFLINE_THREADS=$t python3 tools/oneshot_probe.py cfg0 21 | grep -v "^  " | sed "s/^/threads=$t /"
  IRB_OFFLINE_THREADS=$t python3 tools/oneshot_probe.py cfg2 11 | grep -v "^  " | sed "s/^/threads=$t /"
done
nproc
