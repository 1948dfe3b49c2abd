#!/bin/bash
# The bench line and the rocprofv3 kernel trace of the same command on the SAME device, back to back (their
# k_step_fused<512,2,true> durations must agree): gpurun_out/bench_pair/{bench.json, kt/}
set -e
R=/root/repo
O=$R/gpurun_out/bench_pair
rm -rf $O; mkdir -p $O
cd $R && python3 bench.py > $O/bench.json 2> $O/bench.err
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats --output-format csv -d $O/kt -- python3 $R/bench.py --no-pmc > $O/bench_under_trace.json 2> $O/bench_under_trace.err
echo done
