#!/bin/bash
# per-kernel durations (rocprofv3 kernel trace) of the 2^20-point real FFT for the plan knobs given as "VAR=val ..." arms
cd /tmp && export TMPDIR=/tmp
R=/root/repo
O=$R/gpurun_out/lfft_ab
rm -rf $O; mkdir -p $O
i=0
for arm in "$@"; do
  i=$((i+1))
  ( for kv in $arm; do export "$kv"; done
    rocprofv3 --kernel-trace --stats --output-format csv -d $O/arm$i -- python3 $R/tools/perf_probe.py lfft ${LOGN:-20} 1 200 > $O/arm$i.log 2>&1 )
  echo "== $arm: $(grep '^lfft' $O/arm$i.log)"
  python3 - $O/arm$i <<'PY'
import csv, glob, sys
for f in glob.glob(sys.argv[1] + "/*/*kernel_stats.csv"):
    for r in csv.DictReader(open(f)):
        if "lfft" in r["Name"] or "rsplit" in r["Name"]:
            print(f"   {float(r['AverageNs'])/1000:8.2f} us x {r['Calls']:>4}  {r['Name'][:100]}")
PY
done
