#!/bin/bash
# Per-kernel durations (rocprofv3 kernel trace) and SQ / cache counters of the time-tiled offline call, per MAC kernel variant.
# Output: gpurun_out/tiled_prof/*.  usage: tools/tiled_profile.sh "<perf_probe offline args>" <kernel ids...>
cd /tmp && export TMPDIR=/tmp
R=/root/repo
O=$R/gpurun_out/tiled_prof
mkdir -p $O
ARGS="$1"; shift
tag=$(echo $ARGS | tr ' ' '_')
for k in "$@"; do
  export IRB_TILED_KERNEL=$k
  rocprofv3 --kernel-trace --stats --output-format csv -d $O/kt_${tag}_k$k -- python3 $R/tools/perf_probe.py offline $ARGS > $O/kt_${tag}_k$k.log 2>&1
  cp $O/kt_${tag}_k$k/*/*kernel_stats.csv $O/kt_${tag}_k$k.csv 2>/dev/null
  i=0
  for set in "SQ_WAVES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY" "SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_INSTS_VMEM SQ_ACTIVE_INST_VMEM SQ_INSTS_SALU SQ_WAIT_INST_LDS" "TCC_HIT_sum TCC_MISS_sum" "TCP_TOTAL_CACHE_ACCESSES_sum TCP_TCC_READ_REQ_sum" "GRBM_GUI_ACTIVE"; do
    i=$((i+1))
    rocprofv3 --pmc $set --output-format csv -d $O/pmc_${tag}_k${k}_$i -- python3 $R/tools/perf_probe.py offline $ARGS > /dev/null 2> $O/pmc_${tag}_k${k}_$i.err || true
  done
done
python3 - "$O" "$tag" "$@" <<'PY'
import csv, glob, sys, collections
O, tag = sys.argv[1], sys.argv[2]
for k in sys.argv[3:]:
    print(f"== {tag} kernel variant {k}")
    try:
        for r in csv.DictReader(open(f"{O}/kt_{tag}_k{k}.csv")):
            if "irb::" in r["Name"]: print(f"  {r['AverageNs']:>12} ns x {r['Calls']:>4}  {r['Name'][:90]}")
    except Exception as e: print("  no kernel stats", e)
    acc = collections.defaultdict(list)
    for f in glob.glob(f"{O}/pmc_{tag}_k{k}_*/**/*counter_collection.csv", recursive=True):
        for r in csv.DictReader(open(f)):
            if "k_cmac_tiled" in r["Kernel_Name"]: acc[r["Counter_Name"]].append(float(r["Counter_Value"]))
    for c, v in sorted(acc.items()): print(f"  {c:32s} {sum(v)/len(v):16.1f}  ({len(v)} dispatches)")
PY
