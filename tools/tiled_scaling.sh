#!/bin/bash
# kernel durations of the time-tiled call vs the number of blocks (one stream, configs[1] geometry): slope = steady-state rate,
# intercept = per-launch fixed cost.  usage: tools/tiled_scaling.sh <kernel variant> [IRB_* assignments...]
cd /tmp && export TMPDIR=/tmp
R=/root/repo
O=$R/gpurun_out/tiled_scaling
mkdir -p $O
k=$1; shift
for kv in "$@"; do export "$kv"; done
export IRB_TILED_KERNEL=$k
for T in 235 470 940 1880 3750 7500 15000; do
  rocprofv3 --kernel-trace --stats --output-format csv -d $O/k${k}_T$T -- python3 $R/tools/perf_probe.py offline 256 96000 $T 1 > $O/k${k}_T$T.log 2>&1
  f=$(ls $O/k${k}_T$T/*/*kernel_stats.csv | head -1)
  echo "variant $k $* T=$T: $(grep -o 'splits=[0-9]*' $O/k${k}_T$T.log) $(python3 -c "
import csv,sys
for r in csv.DictReader(open('$f')):
    n=r['Name']
    if 'irb::' in n and 'ir_fft' not in n: print(n.split('irb::')[1].split('<')[0], round(float(r['AverageNs'])/1000,1), 'us;', end=' ')
")"
done
