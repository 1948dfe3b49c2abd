#!/bin/bash
for shape in "256 4096 1891 1 200" "256 4096 1024 1 200" "256 16384 1024 1 200" "256 8192 1024 4 100" "128 2048 2048 2 100" "256 32768 1024 1 200"; do
  echo "== offline $shape"
  echo -n "packed-FMA  "; IRB_DEBUG=mfma_min_t=0 python tools/perf_probe.py offline $shape
  echo -n "mfma        "; python tools/perf_probe.py offline $shape
done
