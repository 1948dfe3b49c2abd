#!/bin/bash
cd /root/repo
O=gpurun_out/robust.txt
: > $O
for i in 1 2 3; do timeout -k 10 300 python -m pytest tests -m gpu -q -x 2>&1 | tail -1 >> $O || exit 1; done
timeout -k 10 300 python tools/fuzz_paths.py 240 13 >> $O 2>&1 || exit 1
timeout -k 10 200 python tools/rt_server_soak.py 60 23 2>&1 | tail -3 >> $O || exit 1
timeout -k 10 400 python tools/soak.py 2>&1 | tail -4 >> $O
