#!/bin/bash
# HBM traffic (FETCH_SIZE / WRITE_SIZE in their own passes; FETCH_SIZE x2 on gfx950, KiB -> bytes), the SQ wait / issue split and
# the L2 hit rate of the 8-blocks-per-call tile of the packed-FMA MAC (k_cmac_tiled_pk<512,8,4,4,true>) at 2048 instances -- the
# "configs[1] x 2048, 8 blocks per call" entry of the bench line -- plus the kernel trace of the same command.
# usage: tools/tiled8_pmc.sh [tag]        output: gpurun_out/tiled8_pmc_<tag>.txt
cd /tmp && export TMPDIR=/tmp
R=/root/repo
TAG=${1:-base}
O=$R/gpurun_out/tiled8_pmc_$TAG
rm -rf $O; mkdir -p $O
ARGS="256 96000 8 2048 10"   # (IRB_DEBUG=tiled_fused8=0 in the environment: the three-launch form)
rocprofv3 --kernel-trace --stats --output-format csv -d $O/kt -- python3 $R/tools/perf_probe.py offline $ARGS > $O/kt.log 2>&1 || exit 1
i=0
for set in "FETCH_SIZE" "WRITE_SIZE" "SQ_WAVES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY" "SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_INSTS_LDS SQ_INSTS_SALU" "TCC_HIT_sum TCC_MISS_sum TCC_EA0_RDREQ_sum TCC_EA0_WRREQ_sum" "GRBM_GUI_ACTIVE"; do
  i=$((i+1))
  rocprofv3 --pmc $set --output-format csv -d $O/pmc_$i -- python3 $R/tools/perf_probe.py offline $ARGS > /dev/null 2> $O/pmc_$i.err || true
done
python3 - "$O" "$TAG" <<'PY' > $R/gpurun_out/tiled8_pmc_$TAG.txt
import csv, glob, sys, collections
O, tag = sys.argv[1], sys.argv[2]
print(f"== perf_probe offline 256 96000 8 2048 10 ({tag}): 2048 mono instances x 8 blocks per call, P = 375, N = 512")
print(open(f"{O}/kt.log").read().strip().splitlines()[-1])
for f in glob.glob(f"{O}/kt/**/*kernel_stats.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        if "irb::" in r["Name"]: print(f"  {float(r['AverageNs'])/1e3:10.1f} us x {r['Calls']:>4}  {r['Name'][:100]}")
for kern in ("k_tile8_fused", "k_cmac_tiled", "k_push_fft_batch", "k_ifft_ola_run"):
    acc = collections.defaultdict(list)
    for f in glob.glob(f"{O}/pmc_*/**/*counter_collection.csv", recursive=True):
        for r in csv.DictReader(open(f)):
            if kern in r["Kernel_Name"]: acc[r["Counter_Name"]].append(float(r["Counter_Value"]))
    if not acc: continue
    m = {k: sum(v) / len(v) for k, v in acc.items()}
    print(f"-- {kern} (means per launch)")
    for k in sorted(m): print(f"  {k:28s} {m[k]:18.1f}  ({len(acc[k])} dispatches)")
    if "FETCH_SIZE" in m:
        rd, wr = m["FETCH_SIZE"] * 1024 * 2, m.get("WRITE_SIZE", 0) * 1024
        print(f"  HBM read {rd/1e6:.1f} MB (FETCH_SIZE x2, KiB->B)  write {wr/1e6:.1f} MB  total {(rd+wr)/1e6:.1f} MB per launch")
    if "SQ_WAVE_CYCLES" in m:
        print(f"  wave cycles: issuing {m['SQ_ACTIVE_INST_ANY']/m['SQ_WAVE_CYCLES']:.3f}  waiting (memory/barrier) {m['SQ_WAIT_ANY']/m['SQ_WAVE_CYCLES']:.3f}  issue-stalled {m['SQ_WAIT_INST_ANY']/m['SQ_WAVE_CYCLES']:.3f}  waves {m['SQ_WAVES']:.0f}")
    if "TCC_HIT_sum" in m: print(f"  L2 hit rate {m['TCC_HIT_sum']/(m['TCC_HIT_sum']+m['TCC_MISS_sum']):.3f}")
S, T, P, N = 2048, 8, 375, 512
print(f"minimum traffic of the call: H {S*P*N*4/1e6:.1f} MB + FDL window (P+T rows) {S*(P+T)*N*4/1e6:.1f} MB + partial spectra out/in {2*S*T*N*4/1e6:.1f} MB + audio {2*S*T*256*4/1e6:.1f} MB")
PY
cat $R/gpurun_out/tiled8_pmc_$TAG.txt
