#!/bin/bash
# Why does k_tile8_fused (2048 instances x 8 blocks per call, configs[1] geometry) stream at 0.72 of the HBM peak when k_step_fused (the same
# rows, one block per call) streams at 0.82?  Both kernels, counter set by counter set (separate rocprofv3 --pmc passes), plus the kernel
# trace: HBM traffic (FETCH_SIZE x2 on gfx950, KiB -> bytes), the SQ wait / issue split, scalar against vector instructions, and the
# memory side -- L1 (TCP) pending / return stalls and read latency, texture addresser stalls, L2 (TCC) read requests in flight towards
# HBM and credit stalls.
# usage: tools/tiled8_pmc.sh [tag]        output: gpurun_out/tiled8_pmc_<tag>.txt      (IRB_DEBUG=tiled_fused8=0: the three-launch form)
cd /tmp && export TMPDIR=/tmp
R=/root/repo
TAG=${1:-base}
O=$R/gpurun_out/tiled8_pmc_$TAG
rm -rf $O; mkdir -p $O
# (a block's counter registers are few: TCP and TCC take four of these per pass, TA two -- more and the profiler aborts the process)
SETS=("FETCH_SIZE" "WRITE_SIZE"
      "SQ_WAVES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY"
      "SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_SMEM SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_INSTS_LDS"
      "SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_LDS SQ_INST_CYCLES_SALU SQ_INST_LEVEL_VMEM"
      "TCP_PENDING_STALL_CYCLES_sum TCP_TCC_READ_REQ_sum TCP_TCC_READ_REQ_LATENCY_sum TCP_TCP_LATENCY_sum"
      "TCP_TOTAL_READ_sum TCP_TCR_TCP_STALL_CYCLES_sum TCP_TD_TCP_STALL_CYCLES_sum TCP_TCP_TA_ADDR_STALL_CYCLES_sum"
      "TA_ADDR_STALLED_BY_TC_CYCLES_sum TA_TA_BUSY_sum"
      "TA_DATA_STALLED_BY_TC_CYCLES_sum TA_BUFFER_READ_WAVEFRONTS_sum"
      "TCC_EA0_RDREQ_sum TCC_EA0_RDREQ_LEVEL_sum TCC_BUSY_sum TCC_CYCLE_sum"
      "TCC_EA0_RDREQ_DRAM_CREDIT_STALL_sum TCC_TAG_STALL_sum TCC_HIT_sum TCC_MISS_sum"
      "GRBM_GUI_ACTIVE")
for w in tile8 step; do
  if [ $w = tile8 ]; then ARGS="256 96000 8 2048 10 1"; else ARGS="256 96000 1 2048 40 1"; fi
  timeout -k 5 200 rocprofv3 --kernel-trace --stats --output-format csv -d $O/$w/kt -- python3 $R/tools/perf_probe.py offline $ARGS > $O/$w.kt.log 2>&1 || exit 1
  echo "$w kernel trace done" >> $O/progress.txt
  i=0
  for set in "${SETS[@]}"; do
    i=$((i+1))
    timeout -k 5 200 rocprofv3 --pmc $set --output-format csv -d $O/$w/pmc_$i -- python3 $R/tools/perf_probe.py offline $ARGS > /dev/null 2> $O/$w.pmc_$i.err || echo "$w set $i ($set) FAILED" >> $O/progress.txt
    echo "$w set $i done" >> $O/progress.txt
  done
done
python3 - "$O" "$TAG" <<'PY' > $R/gpurun_out/tiled8_pmc_$TAG.txt
import csv, glob, sys, collections
O, tag = sys.argv[1], sys.argv[2]
S, P, N, B = 2048, 375, 512, 256
res = {}
for w, kern, T in (("tile8", "k_tile8_fused", 8), ("step", "k_step_fused", 1)):
    print(f"== {w}: perf_probe offline 256 96000 {T} 2048 ({tag}): 2048 mono instances x {T} block(s) per call, P = 375, N = 512")
    print("  " + open(f"{O}/{w}.kt.log").read().strip().splitlines()[-1])
    us = None
    for f in glob.glob(f"{O}/{w}/kt/**/*kernel_stats.csv", recursive=True):
        for r in csv.DictReader(open(f)):
            if "irb::" in r["Name"]:
                print(f"  {float(r['AverageNs'])/1e3:10.1f} us x {r['Calls']:>4}  {r['Name'][:100]}")
                if kern in r["Name"]: us = float(r["AverageNs"]) / 1e3
    acc, dur = collections.defaultdict(list), collections.defaultdict(list)
    for f in glob.glob(f"{O}/{w}/pmc_*/**/*counter_collection.csv", recursive=True):
        for r in csv.DictReader(open(f)):
            if kern in r["Kernel_Name"]:
                acc[r["Counter_Name"]].append(float(r["Counter_Value"]))
                dur[r["Counter_Name"]].append(float(r["End_Timestamp"]) - float(r["Start_Timestamp"]))     # ns, of THIS dispatch under the counters
    m = {k: sum(v) / len(v) for k, v in acc.items()}
    if "GRBM_GUI_ACTIVE" in acc:
        # the shader clock while the kernel ran: busy cycles (summed over the 8 XCDs) / 8 / the same dispatch's duration
        ghz = [v / 8.0 / d for v, d in zip(acc["GRBM_GUI_ACTIVE"], dur["GRBM_GUI_ACTIVE"]) if d > 0]
        m["_ghz"] = sum(ghz) / len(ghz)
        m["_us_pmc"] = sum(dur["GRBM_GUI_ACTIVE"]) / len(dur["GRBM_GUI_ACTIVE"]) / 1e3
    m["_us"] = us
    res[w] = m
    print(f"-- {kern} (means per launch)")
    for k in sorted(m):
        if not k.startswith("_"): print(f"  {k:36s} {m[k]:18.1f}  ({len(acc[k])} dispatches)")
    g = lambda k: m.get(k, float("nan"))
    rd, wr = g("FETCH_SIZE") * 1024 * 2, g("WRITE_SIZE") * 1024
    print(f"  HBM read {rd/1e6:.1f} MB (FETCH_SIZE x2, KiB->B)  write {wr/1e6:.1f} MB  total {(rd+wr)/1e6:.1f} MB per launch" + (f" = {(rd+wr)/us/1e6:.3f} TB/s over the traced {us:.1f} us" if us else ""))
    print(f"  wave cycles: issuing {g('SQ_ACTIVE_INST_ANY')/g('SQ_WAVE_CYCLES'):.3f}  waiting (memory/barrier) {g('SQ_WAIT_ANY')/g('SQ_WAVE_CYCLES'):.3f}  issue-stalled {g('SQ_WAIT_INST_ANY')/g('SQ_WAVE_CYCLES'):.3f}  waves {g('SQ_WAVES'):.0f}")
    print(f"  instructions per wave: VALU {g('SQ_INSTS_VALU')/g('SQ_WAVES'):.0f}  SALU {g('SQ_INSTS_SALU')/g('SQ_WAVES'):.0f}  SMEM {g('SQ_INSTS_SMEM')/g('SQ_WAVES'):.0f}  VMEM read {g('SQ_INSTS_VMEM_RD')/g('SQ_WAVES'):.0f}  LDS {g('SQ_INSTS_LDS')/g('SQ_WAVES'):.0f}")
    print(f"  vector-memory instructions in flight per wave while it lives (SQ_INST_LEVEL_VMEM / SQ_WAVE_CYCLES): {g('SQ_INST_LEVEL_VMEM')/g('SQ_WAVE_CYCLES'):.2f}")
    print(f"  L1: read requests to L2 {g('TCP_TCC_READ_REQ_sum'):.0f}, mean L2 round trip {g('TCP_TCC_READ_REQ_LATENCY_sum')/g('TCP_TCC_READ_REQ_sum'):.0f} cycles; pending-stall cycles {g('TCP_PENDING_STALL_CYCLES_sum'):.3g}, "
          f"return-path (TCR) stall {g('TCP_TCR_TCP_STALL_CYCLES_sum'):.3g}, data-return (TD) stall {g('TCP_TD_TCP_STALL_CYCLES_sum'):.3g}, address-in stall {g('TCP_TCP_TA_ADDR_STALL_CYCLES_sum'):.3g}")
    print(f"  texture addresser: busy {g('TA_TA_BUSY_sum'):.3g}, address stalled by L1 {g('TA_ADDR_STALLED_BY_TC_CYCLES_sum'):.3g}, data stalled by L1 {g('TA_DATA_STALLED_BY_TC_CYCLES_sum'):.3g}")
    print(f"  L2: read requests to HBM {g('TCC_EA0_RDREQ_sum'):.0f}, in flight on average (LEVEL / busy cycles of the 128 channels) {g('TCC_EA0_RDREQ_LEVEL_sum')/g('TCC_BUSY_sum'):.1f} per channel, "
          f"DRAM credit stalls {g('TCC_EA0_RDREQ_DRAM_CREDIT_STALL_sum'):.3g}, tag stalls {g('TCC_TAG_STALL_sum'):.3g}, hit rate {g('TCC_HIT_sum')/(g('TCC_HIT_sum')+g('TCC_MISS_sum')):.3f}")
    print(f"  mean HBM read latency seen by the L2 (LEVEL / requests): {g('TCC_EA0_RDREQ_LEVEL_sum')/g('TCC_EA0_RDREQ_sum'):.0f} L2 cycles")
    print(f"  clock while the kernel ran: GRBM_GUI_ACTIVE {g('GRBM_GUI_ACTIVE'):.4g} cycles over 8 XCDs in {g('_us_pmc'):.1f} us (the same dispatches, under the counter) = {g('_ghz'):.3f} GHz")
print(f"minimum traffic of the 8-block call: H {S*P*N*4/1e6:.1f} MB + FDL window (P+T rows) {S*(P+8)*N*4/1e6:.1f} MB + audio {2*S*8*B*4/1e6:.1f} MB; of the one-block step: {S*(8*P*N+16*B)/1e6:.1f} MB")
PY
cat $R/gpurun_out/tiled8_pmc_$TAG.txt
rm -rf $O          # (the raw passes are hundreds of MB; the summary is what is kept)
