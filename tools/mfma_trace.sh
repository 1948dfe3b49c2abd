#!/bin/bash
# Per-kernel durations (rocprofv3 kernel trace) of the time-tiled offline call with the matrix-core MAC.  Output: gpurun_out/mfma_trace/
# usage: tools/mfma_trace.sh "<perf_probe offline args>" ...
cd /tmp && export TMPDIR=/tmp
R=/root/repo
O=$R/gpurun_out/mfma_trace
mkdir -p $O
for ARGS in "$@"; do
  tag=$(echo $ARGS | tr ' ' '_')
  rocprofv3 --kernel-trace --stats --output-format csv -d $O/kt_$tag -- python3 $R/tools/perf_probe.py offline $ARGS > $O/kt_$tag.log 2>&1
  cp $O/kt_$tag/*/*kernel_stats.csv $O/kt_$tag.csv 2>/dev/null
  echo "== offline $ARGS"
  python3 - $O/kt_$tag.csv <<'PY'
import csv, sys
for r in csv.DictReader(open(sys.argv[1])):
    if "irb::" in r["Name"]: print(f"  {r['AverageNs']:>12} ns x {r['Calls']:>4}  {r['Name'][:100]}")
PY
done
