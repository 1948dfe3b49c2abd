#!/bin/bash
set -e
mkdir -p gpurun_out
{
echo "--- bench.one_shot alone in a fresh process"
timeout -k 10 120 python - <<'PY'
import sys, numpy as np
sys.path.insert(0, ".")
import bench
from irbaboon_amd.synth import decaying_ir, white_noise
e = bench.one_shot(0, np, decaying_ir, white_noise, 1, 1, 256, 96000, 960000, "x")
print(e["wall_ms_c_abi"], e["wall_ms_python_mirror"], e["direct_pinned"]["wall_ms_c_abi"], e["breakdown_ms"]["fft_in_ms"], e["breakdown_ms"]["pack_helper_ms"])
PY
echo "--- the same after import torch + torch.cuda.synchronize"
timeout -k 10 120 python - <<'PY'
import sys, numpy as np, torch
sys.path.insert(0, ".")
torch.cuda.synchronize(0)
import bench
from irbaboon_amd.synth import decaying_ir, white_noise
e = bench.one_shot(0, np, decaying_ir, white_noise, 1, 1, 256, 96000, 960000, "x")
print(e["wall_ms_c_abi"], e["wall_ms_python_mirror"], e["direct_pinned"]["wall_ms_c_abi"], e["breakdown_ms"]["fft_in_ms"], e["breakdown_ms"]["pack_helper_ms"])
print("threads:", len(__import__("os").listdir("/proc/self/task")))
PY
echo "--- probe"
timeout -k 10 120 python tools/oneshot_probe.py cfg1 21
nproc; cat /proc/cpuinfo | grep "model name" | head -1; python -c "import os; print(len(os.sched_getaffinity(0)))"
} > gpurun_out/r5_benchctx.txt 2>&1
cat gpurun_out/r5_benchctx.txt
