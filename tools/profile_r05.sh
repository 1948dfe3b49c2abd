#!/bin/bash
# Round-5 evidence in ONE gpurun call (run once, on the final code): the bench line and the rocprofv3 kernel trace of the same command on the
# same device, the kernel trace of the GPU test suite (kernel coverage), FETCH / WRITE passes of the headline launch, the one-shot function in
# its three forms with the device timeline of the default (staged) one, per-kernel durations of the long (matrix-core) calls, the real-time
# callback's forms, the accuracy table.  Outputs under gpurun_out/prof_r05/ (profiles/collect_r05.py turns them into profiles/r05_*).
# (gpurun copies at most 64 MiB back: the raw traces stay in /tmp on the box, only stats / summaries / the counters go to gpurun_out)
R=/root/repo
O=/tmp/prof_r05_raw
K=$R/gpurun_out/prof_r05
rm -rf $O $K $R/gpurun_out/mfma_trace; mkdir -p $O $K
cd $R
echo "== bench (plain), then the same command under kernel-trace"
python3 bench.py > $O/bench.json 2> $O/bench.err || { tail -5 $O/bench.err; exit 1; }
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats --output-format csv -d $O/bench_kt -- python3 $R/bench.py --no-pmc > $O/bench_under_trace.json 2> $O/bench_under_trace.err
echo "== GPU test suite under kernel-trace"
cd $R && rocprofv3 --kernel-trace --stats --output-format csv -d $O/tests_kt -- python3 -m pytest tests -m gpu -x -q -p no:cacheprovider --deselect tests/test_gpu_parity.py::test_bench_multi_gpu_code_path_rehearsal > $O/tests_kt.log 2>&1; tail -2 $O/tests_kt.log; cd /tmp
cp $R/gpurun_out/resident_vs_process.txt $O/resident_vs_process.txt 2>/dev/null
echo "== PMC: headline kernel"
for c in FETCH_SIZE WRITE_SIZE; do
  rocprofv3 --pmc $c --output-format csv -d $O/pmc_$c -- python3 $R/bench.py --traffic-child --instances 2048 > /dev/null 2> $O/pmc_$c.err
done
cd $R
echo "== one-shot function: default (staged), irb_host_alloc buffers (direct), opt-in registration"
(for cfg in cfg1 cfg0 cfg2; do
   python3 tools/oneshot_probe.py $cfg 21
   python3 tools/oneshot_probe.py $cfg 21 pinned
   IRB_OFFLINE_DIRECT=1 python3 tools/oneshot_probe.py $cfg 21
 done) > $O/oneshot.txt 2>&1
./tools/oneshot_timeline.sh cfg1 > /dev/null 2>&1; cp gpurun_out/oneshot_timeline_cfg1.txt $O/oneshot_timeline.txt
echo "== long calls on the matrix cores: per-kernel durations"
./tools/mfma_trace.sh "256 96000 3750 1 100" "256 96000 4125 1 100" "256 96000 1024 1 100" "128 480000 2048 2 50" "256 96000 1024 16 20" > $O/mfma_trace.txt 2>&1
echo "== real-time callback, accuracy"
(for hb in 64 256 512 1024; do python3 tools/perf_probe.py rtc $hb 3000; IRB_RT_RESIDENT=0 python3 tools/perf_probe.py rtc $hb 3000; done) > $O/rt_ab.txt 2>&1
python3 tools/accuracy_report.py > /dev/null 2>&1; cp gpurun_out/accuracy.txt $O/accuracy.txt
echo "== keep: line, stats, counters, summaries"
cp $O/*.json $O/*.txt $O/*.err $O/*.log $K/ 2>/dev/null
for d in bench_kt tests_kt; do mkdir -p $K/$d/x; f=$(ls -S $O/$d/*/*_kernel_stats.csv | head -1); cp $f $K/$d/x/; done
for c in FETCH_SIZE WRITE_SIZE; do mkdir -p $K/pmc_$c/x; for f in $O/pmc_$c/*/*_counter_collection.csv; do (head -1 $f; grep k_step_fused $f) > $K/pmc_$c/x/$(basename $f); done; done
rm -rf $R/gpurun_out/mfma_trace/kt_*/
du -sh $R/gpurun_out | tail -1
echo done
