#!/bin/bash
# kernel-trace timelines of the latency path (kernel duration vs launch gap) + batch profile variants
set -e
cd /tmp && export TMPDIR=/tmp
R=/root/repo
O=$R/gpurun_out/latprof
rm -rf $O; mkdir -p $O
prof() { name=$1; shift; env "$@" rocprofv3 --kernel-trace --stats --output-format csv -d $O/$name -- python3 $R/tools/perf_probe.py few $ARGS > $O/$name.log 2>&1; }
ARGS="1 2 256 2048 2000";  prof fusedP8 IRB_COLS=0
ARGS="1 2 256 2048 2000";  prof colsP8 IRB_COLS=1
ARGS="1 1 256 96000 2000"; prof cols_cfg2 IRB_COLS=1
ARGS="1 1 256 96000 2000"; prof split_cfg2 IRB_COLS=0
ARGS="1 2 128 480000 2000"; prof cols_cfg3 IRB_COLW=4 IRB_PSPLITS=4
ARGS="1 2 128 480000 2000"; prof split_cfg3 IRB_COLS=0
cd $R
out=gpurun_out/batch_sweep.txt
: > $out
run() { env "$@" python tools/perf_probe.py few $ARGS >> $out 2>&1; }
for I in 32 64 128 192 256 384 512; do
  ARGS="$I 1 256 96000 300"
  run IRB_COLS=0
  run IRB_COLS=0 IRB_FUSED_DEEP=1 IRB_SPLITS=1
  run IRB_FUSED_FROM=100000 IRB_COLS_SC1=1 IRB_PSPLITS=1
  run IRB_FUSED_FROM=100000 IRB_COLS_SC1=1 IRB_PSPLITS=2
  run IRB_FUSED_FROM=100000 IRB_COLS_SC1=1 IRB_PSPLITS=4
done
