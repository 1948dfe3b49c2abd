#!/bin/bash
# Round-2 evidence: kernel-trace of the default bench and of the GPU test suite, PMC passes for the headline kernel
# and the latency kernels.  Outputs under gpurun_out/prof_r02/ (copied into profiles/ by profiles/collect_r02.py).
set -e
cd /tmp && export TMPDIR=/tmp
R=/root/repo
O=$R/gpurun_out/prof_r02
rm -rf $O; mkdir -p $O
echo "== bench (default command) under kernel-trace"
rocprofv3 --kernel-trace --stats --output-format csv -d $O/bench_kt -- python3 $R/bench.py --no-pmc > $O/bench_kt.json 2> $O/bench_kt.err
echo "== GPU test suite under kernel-trace"
cd $R && rocprofv3 --kernel-trace --stats --output-format csv -d $O/tests_kt -- python3 -m pytest tests -m gpu -x -q -p no:cacheprovider --deselect tests/test_gpu_parity.py::test_bench_multi_gpu_code_path_rehearsal > $O/tests_kt.log 2>&1; cd /tmp
echo "== PMC: headline kernel"
for c in FETCH_SIZE WRITE_SIZE; do
  rocprofv3 --pmc $c --output-format csv -d $O/pmc_$c -- python3 $R/bench.py --traffic-child --instances 2048 > /dev/null 2> $O/pmc_$c.err
done
echo "== PMC: latency kernels (configs[1] single stream, configs[2] stereo)"
for set in "SQ_WAVES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY" "TCC_HIT_sum TCC_MISS_sum" "GRBM_GUI_ACTIVE" "FETCH_SIZE" ; do
  tag=$(echo $set | tr ' ' '_' | cut -c1-40)
  rocprofv3 --pmc $set --output-format csv -d $O/lat1_$tag -- python3 $R/tools/perf_probe.py few 1 1 256 96000 200 > /dev/null 2> $O/lat1_$tag.err || true
  rocprofv3 --pmc $set --output-format csv -d $O/lat2_$tag -- python3 $R/tools/perf_probe.py few 1 2 128 480000 200 > /dev/null 2> $O/lat2_$tag.err || true
done
echo "== kernel-trace: latency kernels"
rocprofv3 --kernel-trace --stats --output-format csv -d $O/lat1_kt -- python3 $R/tools/perf_probe.py few 1 1 256 96000 2000 > $O/lat1_kt.log 2>&1
rocprofv3 --kernel-trace --stats --output-format csv -d $O/lat2_kt -- python3 $R/tools/perf_probe.py few 1 2 128 480000 2000 > $O/lat2_kt.log 2>&1
rocprofv3 --kernel-trace --stats --output-format csv -d $O/lfft_kt -- python3 $R/tools/perf_probe.py lfft 20 1 100 > $O/lfft_kt.log 2>&1
echo done
