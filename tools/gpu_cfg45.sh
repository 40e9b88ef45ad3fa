#!/bin/bash
cd $GRAFT_REPO_ROOT; export TMPDIR=/tmp
timeout -k 10 300 python bench.py --config 4 --steps 5 --warmup 2 --no-cpu-baseline 2>gpurun_out/cfg4.err > gpurun_out/cfg4.json || exit 1
timeout -k 10 300 python bench.py --config 5 --steps 2 --warmup 1 --no-cpu-baseline 2>gpurun_out/cfg5.err > gpurun_out/cfg5.json || exit 1
python - <<'PY'
import json
for f in ("cfg4","cfg5"):
    d=json.loads(open(f"gpurun_out/{f}.json").read().strip().splitlines()[-1])
    print(f, d["value"], d["ms_per_step"], d.get("kernels_ms_per_step"), d.get("node_cache_gib"), d.get("roofline",{}).get("kernel"))
PY
