#!/bin/bash
# Round-4 evidence in one gpurun call: the bench line and the rocprofv3 kernel trace of the same command on the same device, the
# kernel trace of the GPU test suite, FETCH/WRITE passes of the headline launch, the one-shot function (direct / staged forms,
# kernel + copy trace, registration cost), counters of the 8-blocks-per-call kernel in both forms, stamps / trace / counters / A-B
# of the matrix-core MAC, what a live resident server does to other handles with and without quiet sections, the accuracy table.
# Outputs under gpurun_out/prof_r04/ (profiles/collect_r04.py turns them into profiles/r04_*).
# (gpurun copies at most 64 MiB back: the raw traces stay in /tmp on the box, only stats / summaries / the counters go to gpurun_out)
R=/root/repo
O=/tmp/prof_r04_raw
K=$R/gpurun_out/prof_r04
rm -rf $O $K $R/gpurun_out/mfma_trace $R/gpurun_out/mfma_pmc $R/gpurun_out/tiled8_pmc_*; mkdir -p $O $K
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
cd $R
echo "== one-shot function: forms, trace, registration cost"
(for cfg in cfg1 cfg0 cfg2; do
   python3 tools/oneshot_probe.py $cfg 21
   IRB_OFFLINE_DIRECT=0 python3 tools/oneshot_probe.py $cfg 21
   IRB_OFFLINE_DIRECT=0 IRB_OFFLINE_HELPER=0 python3 tools/oneshot_probe.py $cfg 21
 done) > $O/oneshot.txt 2>&1
./tools/pin_probe > $O/pin_probe.txt 2>&1
(cd /tmp && rocprofv3 --kernel-trace --memory-copy-trace --stats --output-format csv -d $O/oneshot_kt -- python3 $R/tools/oneshot_probe.py cfg1 5 > $O/oneshot_kt.log 2>&1)
echo "== 8 blocks per call at 2048 instances: counters, one launch vs three"
./tools/tiled8_pmc.sh fused > /dev/null 2>&1; IRB_TILED_FUSED8=0 ./tools/tiled8_pmc.sh three_launches > /dev/null 2>&1
cat gpurun_out/tiled8_pmc_fused.txt gpurun_out/tiled8_pmc_three_launches.txt > $O/tiled8_pmc.txt
timeout -k 5 120 ./tools/clock_under_load 2048 > $O/clock_under_load.txt 2>&1
echo "== matrix-core MAC: stamps, kernel trace of the offline calls, counters, A/B against the vector kernel"
(for a in "3750 1 375" "4096 1 375" "1024 4 375" "2048 2 3750"; do timeout -k 5 60 ./tools/mfma_stamps $a; done; echo "== the same kernel with no loads in its loop (timing only)"; timeout -k 5 60 ./tools/mfma_stamps_exp2 3750 1 375) > $O/mfma_stamps.txt 2>&1
./tools/mfma_trace.sh "256 96000 3750 1 100" "256 96000 4125 1 100" "256 96000 1024 1 100" "128 480000 2048 2 50" "256 96000 1024 16 20" > $O/mfma_trace.txt 2>&1
./tools/mfma_pmc.sh "3750 1 375" > $O/mfma_pmc.txt 2>&1
./tools/mfma_ab.sh; cp gpurun_out/mfma_ab.txt $O/mfma_ab.txt
echo "== a live resident server and the rest of the process: with quiet sections (the library), without (diagnostic)"
python3 -m pytest tests/test_gpu_parity.py -q -x -k resident_server_does_not_stall -p no:cacheprovider > $O/resident_test.log 2>&1; tail -1 $O/resident_test.log
IRB_DEVICE_QUIET=0 timeout -k 10 200 python3 -m pytest tests/test_gpu_parity.py -q -x -k resident_server_does_not_stall -p no:cacheprovider > $O/resident_test_unprotected.log 2>&1; tail -1 $O/resident_test_unprotected.log
(echo "== quiet sections (default)"; cat gpurun_out/resident_vs_process.txt; echo "== IRB_DEVICE_QUIET=0 (diagnostic: the library's device-wide waits not bracketed)"; cat gpurun_out/resident_vs_process_unprotected.txt) > $O/resident_vs_process.txt 2>&1
echo "== real-time callback, accuracy"
(for hb in 64 256 512 1024; do python3 tools/perf_probe.py rtc $hb 3000; IRB_RT_RESIDENT=0 python3 tools/perf_probe.py rtc $hb 3000; done) > $O/rt_ab.txt 2>&1
python3 tools/accuracy_report.py > /dev/null 2>&1; cp gpurun_out/accuracy.txt $O/accuracy.txt
echo "== keep: line, stats, counters, summaries"
cp $O/*.json $O/*.txt $O/*.err $O/*.log $K/ 2>/dev/null
for d in bench_kt tests_kt oneshot_kt; do mkdir -p $K/$d/x; f=$(ls -S $O/$d/*/*_kernel_stats.csv | head -1); cp $f $K/$d/x/; done
f=$(ls -S $O/oneshot_kt/*/*_kernel_trace.csv | head -1); tail -n +2 $f | tail -n 400 > /tmp/tail.csv; (head -1 $f; cat /tmp/tail.csv) > $K/oneshot_kt/x/oneshot_kernel_trace.csv
for c in FETCH_SIZE WRITE_SIZE; do mkdir -p $K/pmc_$c/x; for f in $O/pmc_$c/*/*_counter_collection.csv; do (head -1 $f; grep k_step_fused $f) > $K/pmc_$c/x/$(basename $f); done; done
rm -rf $R/gpurun_out/tiled8_pmc_fused $R/gpurun_out/tiled8_pmc_three_launches $R/gpurun_out/mfma_trace/kt_*/ $R/gpurun_out/mfma_pmc
du -sh $R/gpurun_out | tail -1
echo done
