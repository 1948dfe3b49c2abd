#!/bin/bash
# SQ counters of the matrix-core tiled MAC (tools/mfma_stamps runs k_cmac_mfma<512> six times on synthetic rows).
# usage: tools/mfma_pmc.sh "<T splits P>" ...   Output: gpurun_out/mfma_pmc/
cd /tmp && export TMPDIR=/tmp
R=/root/repo
O=$R/gpurun_out/mfma_pmc
mkdir -p $O
for ARGS in "$@"; do
  tag=$(echo $ARGS | tr ' ' '_')
  i=0
  for set in "SQ_WAVES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY" "SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_INSTS_MFMA SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_VALU_MFMA_MOPS_F32" "SQ_INSTS_LDS SQ_ACTIVE_INST_LDS SQ_WAIT_INST_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_LDS_ADDR_CONFLICT" "SQ_INSTS_VMEM_RD SQ_ACTIVE_INST_VMEM SQ_INSTS_SALU SQ_ACTIVE_INST_SCA SQ_INST_CYCLES_SALU" "GRBM_GUI_ACTIVE SQ_CYCLES SQ_ACTIVE_INST_MISC SQ_INSTS_BRANCH"; do
    i=$((i+1))
    rocprofv3 --pmc $set --output-format csv -d $O/pmc_${tag}_$i -- $R/tools/mfma_stamps $ARGS > /dev/null 2> $O/pmc_${tag}_$i.err || true
  done
  echo "== mfma_stamps $ARGS"
  python3 - "$O" "$tag" <<'PY'
import csv, glob, sys, collections
O, tag = sys.argv[1], sys.argv[2]
acc = collections.defaultdict(list)
for f in glob.glob(f"{O}/pmc_{tag}_*/**/*counter_collection.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        if "k_cmac_mfma" in r["Kernel_Name"]:
            acc[r["Counter_Name"]].append(float(r["Counter_Value"]))
for k in sorted(acc):
    v = acc[k]
    print(f"  {k:34s} {sum(v)/len(v):16.0f}   (per launch, {len(v)} launches)")
PY
done
