#!/bin/bash
# full GPU check of a build: every -m gpu test, then the driver's bench command (and configs 4 / 5 with "all")
cd $GRAFT_REPO_ROOT; export TMPDIR=/tmp
timeout -k 10 900 python -m pytest tests -m gpu -x -q --timeout 600 2>&1 | tail -3 || exit 1
timeout -k 10 500 python bench.py --gpus 1 --steps 20 --warmup 5 2> gpurun_out/bench_full.err | tee gpurun_out/bench_full.json | cut -c1-330
if [ "$1" = "all" ]; then
  timeout -k 10 300 python bench.py --config 4 --steps 5 --warmup 2 --no-cpu-baseline 2>gpurun_out/cfg4.err > gpurun_out/cfg4.json || exit 1
  timeout -k 10 300 python bench.py --config 5 --steps 2 --warmup 1 --no-cpu-baseline 2>gpurun_out/cfg5.err > gpurun_out/cfg5.json || exit 1
  python - <<'PY'
import json
for f in ("bench_full","cfg4","cfg5"):
    d=json.loads(open(f"gpurun_out/{f}.json").read().strip().splitlines()[-1])
    print(f, round(d["value"],1), round(d["ms_per_step"],2), d.get("kernels_ms_per_step"), d.get("node_cache_gib"))
PY
fi
