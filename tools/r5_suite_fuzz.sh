#!/bin/bash
# one gpurun call: the whole GPU suite, then a short randomised campaign
set -e
mkdir -p gpurun_out
timeout -k 10 600 python -m pytest tests -q -x -m gpu > gpurun_out/r5_suite.txt 2>&1 || { tail -60 gpurun_out/r5_suite.txt; exit 1; }
tail -3 gpurun_out/r5_suite.txt
timeout -k 10 200 python tools/fuzz_paths.py ${1:-120} ${2:-7} > gpurun_out/r5_fuzz.txt 2>&1 || { tail -30 gpurun_out/r5_fuzz.txt; exit 1; }
tail -3 gpurun_out/r5_fuzz.txt
