#!/bin/bash
# once, on the final code: the GPU suite, the randomised path campaign (engines + the one-shot function), the resident-server soak,
# the 20 000-step soak of every step path.  Output: gpurun_out/robust.txt
cd /root/repo
O=gpurun_out/robust.txt
: > $O
timeout -k 10 300 python -m pytest tests -m gpu -q -x > gpurun_out/robust_gpu_suite.log 2>&1; tail -1 gpurun_out/robust_gpu_suite.log >> $O; grep -q ' failed' gpurun_out/robust_gpu_suite.log && exit 1
timeout -k 10 400 python tools/fuzz_paths.py 300 41 >> $O 2>&1 || exit 1
timeout -k 10 200 python tools/rt_server_soak.py 60 23 2>&1 | tail -3 >> $O || exit 1
timeout -k 10 400 python tools/soak.py 2>&1 | tail -4 >> $O
cat $O
