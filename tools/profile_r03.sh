#!/bin/bash
# Round-3 evidence in one gpurun call: the bench line and the rocprofv3 kernel trace of the same command on the same device,
# the kernel trace of the GPU test suite, FETCH/WRITE passes of the headline launch, kernel traces of the real-time callback
# and of the 2^20 FFT chain, SQ/TCC counters + per-workgroup stamps of the time-tiled MAC, the hand-off soak under load.
# Outputs under gpurun_out/prof_r03/ (profiles/collect_r03.py turns them into profiles/r03_*).
R=/root/repo
O=$R/gpurun_out/prof_r03
rm -rf $O; mkdir -p $O
cd $R
echo "== bench (plain), then the same command under kernel-trace"
python3 bench.py > $O/bench.json 2> $O/bench.err || { tail -5 $O/bench.err; exit 1; }
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats --output-format csv -d $O/bench_kt -- python3 $R/bench.py --no-pmc > $O/bench_under_trace.json 2> $O/bench_under_trace.err
echo "== GPU test suite under kernel-trace"
cd $R && rocprofv3 --kernel-trace --stats --output-format csv -d $O/tests_kt -- python3 -m pytest tests -m gpu -x -q -p no:cacheprovider --deselect tests/test_gpu_parity.py::test_bench_multi_gpu_code_path_rehearsal > $O/tests_kt.log 2>&1; tail -2 $O/tests_kt.log; cd /tmp
echo "== PMC: headline kernel"
for c in FETCH_SIZE WRITE_SIZE; do
  rocprofv3 --pmc $c --output-format csv -d $O/pmc_$c -- python3 $R/bench.py --traffic-child --instances 2048 > /dev/null 2> $O/pmc_$c.err
done
echo "== kernel-trace: real-time callback, 2^20 FFT chain"
rocprofv3 --kernel-trace --stats --output-format csv -d $O/rt256_kt -- python3 $R/tools/perf_probe.py rtc 256 3000 > $O/rt256_kt.log 2>&1
rocprofv3 --kernel-trace --stats --output-format csv -d $O/rt512_kt -- python3 $R/tools/perf_probe.py rtc 512 3000 > $O/rt512_kt.log 2>&1
rocprofv3 --kernel-trace --stats --output-format csv -d $O/lfft_kt -- python3 $R/tools/perf_probe.py lfft 20 1 100 > $O/lfft_kt.log 2>&1
cd $R
echo "== time-tiled MAC: counters, stamps"
IRB_TILED_MFMA_MIN_T=0 ./tools/tiled_profile.sh "256 96000 3750 1" 1 0 > $O/tiled_pmc_cfg2.txt 2>&1       # the vector kernels (calls of < 512 blocks; A/B arm of longer ones)
IRB_TILED_MFMA_MIN_T=0 ./tools/tiled_profile.sh "128 480000 2048 2" 1 > $O/tiled_pmc_cfg3.txt 2>&1
for a in "3750 2" "3750 1" "3750 4" "940 8" "7500 1"; do timeout -k 5 60 ./tools/tiled_stamps $a; done > $O/tiled_stamps.txt 2>&1
./tools/valu_peak > $O/valu_peak.txt 2>&1
echo "== matrix-core tiled MAC: kernel trace of the offline calls, stamps, counters, instruction peak, A/B against the vector kernel"
./tools/mfma_trace.sh "256 96000 3750 1 100" "128 480000 2048 2 50" "256 96000 1024 16 20" > $O/mfma_trace.txt 2>&1
for a in "3750 1 375" "1024 4 375" "2048 2 3750"; do timeout -k 5 60 ./tools/mfma_stamps $a; done > $O/mfma_stamps.txt 2>&1
./tools/mfma_pmc.sh "3750 1 375" > $O/mfma_pmc.txt 2>&1
./tools/mfma_peak > $O/mfma_peak.txt 2>&1
./tools/mfma_ab.sh; cp gpurun_out/mfma_ab.txt $O/mfma_ab.txt
echo "== real-time callback A/B, hand-off soak"
for hb in 64 256 512 1024; do python3 tools/perf_probe.py rtc $hb 3000; IRB_RT_RESIDENT=0 python3 tools/perf_probe.py rtc $hb 3000; IRB_RT_RESIDENT=0 IRB_RT_BAR=0 python3 tools/perf_probe.py rtc $hb 3000; IRB_RT_RESIDENT=0 IRB_RT_ONE_LAUNCH=0 python3 tools/perf_probe.py rtc $hb 3000; done > $O/rt_ab.txt 2>&1
timeout -k 10 300 python3 tools/handoff_under_load.py > $O/handoff_under_load.txt 2>&1; tail -2 $O/handoff_under_load.txt
echo done
