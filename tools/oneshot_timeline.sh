#!/bin/bash
# device timeline of the last irb_convolve_periodic call of `oneshot_probe.py <cfg> 5` (rocprofv3 kernel trace): kernels in start order
# usage: tools/oneshot_timeline.sh [cfg1]      output: gpurun_out/oneshot_timeline_<cfg>.txt
cfg=${1:-cfg1}
R=/root/repo
O=$R/gpurun_out/oneshot_tl_$cfg
rm -rf $O; mkdir -p $O
(cd /tmp && export TMPDIR=/tmp && rocprofv3 --kernel-trace --output-format csv -d $O -- python3 $R/tools/oneshot_probe.py $cfg 5 $2 > $O/run.log 2>&1)
python3 - $O $cfg <<'PY'
import csv, glob, sys
f = sorted(glob.glob(sys.argv[1] + "/*/*_kernel_trace.csv"))[0]
rows = sorted((r for r in csv.DictReader(open(f)) if r["Start_Timestamp"].isdigit()), key=lambda r: int(r["Start_Timestamp"]))
last_reset = max(i for i, r in enumerate(rows) if "k_reset" in r["Kernel_Name"])
t0 = int(rows[last_reset]["Start_Timestamp"])
out = [f"device timeline of ONE irb_convolve_periodic call ({sys.argv[2]}), rocprofv3 --kernel-trace; us from the call's first device operation", "   start      duration   kernel"]
for r in rows[last_reset:]:
    s, e = int(r["Start_Timestamp"]), int(r["End_Timestamp"])
    out.append(f"{(s - t0) / 1e3:8.1f}  +{(e - s) / 1e3:8.1f}   {r['Kernel_Name'][:90]}")
open(f"/root/repo/gpurun_out/oneshot_timeline_{sys.argv[2]}.txt", "w").write("\n".join(out) + "\n")
print("\n".join(out))
PY
rm -rf $O
