#!/bin/bash
# round 3, first GPU pass: new tests first, then the whole -m gpu suite, then the bench line
cd $GRAFT_REPO_ROOT; export TMPDIR=/tmp
timeout -k 10 900 python -m pytest tests/test_gpu_round3.py -m gpu -x -q -s --timeout 600 > gpurun_out/r3a_new.log 2>&1; rc=$?
tail -25 gpurun_out/r3a_new.log
[ $rc -ne 0 ] && exit $rc
timeout -k 10 900 python -m pytest tests -m gpu -x -q --timeout 600 --deselect tests/test_gpu_round3.py > gpurun_out/r3a_all.log 2>&1; rc=$?
tail -5 gpurun_out/r3a_all.log
[ $rc -ne 0 ] && exit $rc
timeout -k 10 300 python bench.py --gpus 1 --steps 10 --warmup 3 --no-cpu-baseline 2> gpurun_out/r3a_bench.err > gpurun_out/r3a_bench.json; rc=$?
python - <<'PY'
import json
d=json.loads(open("gpurun_out/r3a_bench.json").read().strip().splitlines()[-1])
print(round(d["value"],1), round(d["ms_per_step"],2), d.get("kernels_ms_per_step"), d.get("failed_chains"), d.get("parity_golden"))
PY
exit $rc
