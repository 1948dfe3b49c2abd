#!/bin/bash
# SQ / TCC / GRBM counters of the headline kernel k_step_fused<512,2,true> (separate --pmc passes over bench.py --traffic-child)
set -e
cd /tmp && export TMPDIR=/tmp
R=/root/repo
O=$R/gpurun_out/headline_pmc
rm -rf $O; mkdir -p $O
for set in "SQ_WAVES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY" "TCC_HIT_sum TCC_MISS_sum" "GRBM_GUI_ACTIVE" "SQ_INSTS_VALU SQ_INSTS_VMEM_RD SQ_INSTS_LDS SQ_INSTS_SALU"; do
  tag=$(echo $set | tr ' ' '_' | cut -c1-40)
  rocprofv3 --pmc $set --output-format csv -d $O/$tag -- python3 $R/bench.py --traffic-child --instances 2048 > /dev/null 2> $O/$tag.err || true
done
python3 - <<PY
import csv,glob,collections,statistics,json
acc=collections.defaultdict(list)
for f in glob.glob("$O/*/*/*counter_collection.csv"):
    for r in csv.DictReader(open(f)):
        if "k_step_fused" in r["Kernel_Name"]: acc[r["Counter_Name"]].append(float(r["Counter_Value"]))
m={k:statistics.mean(v) for k,v in acc.items()}
d={}
if "SQ_WAVE_CYCLES" in m:
    d={"share_waiting_on_memory_or_barrier": m["SQ_WAIT_ANY"]/m["SQ_WAVE_CYCLES"], "share_issuing": m["SQ_ACTIVE_INST_ANY"]/m["SQ_WAVE_CYCLES"],
       "share_issue_stalled": m["SQ_WAIT_INST_ANY"]/m["SQ_WAVE_CYCLES"], "waves": m["SQ_WAVES"]}
if "TCC_HIT_sum" in m: d["l2_hit_rate"]=m["TCC_HIT_sum"]/(m["TCC_HIT_sum"]+m["TCC_MISS_sum"])
json.dump({"kernel":"k_step_fused<512,2,true>, 2048 instances","counters_mean_per_launch":m,"derived":d}, open("$R/gpurun_out/headline_pmc.json","w"), indent=1)
print(json.dumps(d, indent=1)); print({k: round(v) for k,v in m.items()})
PY
