#!/bin/bash
# one gpurun call: the whole GPU suite, then the default bench line
set -e
mkdir -p gpurun_out
timeout -k 10 600 python -m pytest tests -q -x -m gpu > gpurun_out/r5_suite.txt 2>&1 || { tail -60 gpurun_out/r5_suite.txt; exit 1; }
tail -3 gpurun_out/r5_suite.txt
cat gpurun_out/resident_vs_process.txt | tail -1
timeout -k 10 400 python bench.py > gpurun_out/r5_bench.json 2> gpurun_out/r5_bench.err || { tail -30 gpurun_out/r5_bench.err; exit 1; }
python - <<'PY'
import json
d = json.loads(open("gpurun_out/r5_bench.json").read().strip().splitlines()[-1])
print({k: d[k] for k in ("metric", "value", "ms_per_step")}, d["roofline"])
oc = d.get("other_configs", {})
for k, v in oc.items():
    if "one_shot" in k:
        print(k, v.get("wall_ms_c_abi"), v.get("wall_ms_python_mirror"), (v.get("direct_pinned") or {}).get("wall_ms_c_abi"))
PY
