#!/bin/bash
# quick GPU check after a change to the Newton loop: the chain-level parity tests, then bench lines of configs 3 and 4
cd $GRAFT_REPO_ROOT; export TMPDIR=/tmp
timeout -k 10 600 python -m pytest tests -m gpu -x -q --timeout 600 -k "chain or golden or newton or roots or k8 or cfg4 or cfg5 or lost" > gpurun_out/quick_tests.log 2>&1; rc=$?
tail -4 gpurun_out/quick_tests.log
[ $rc -ne 0 ] && exit $rc
for cfg in 3 4; do
timeout -k 10 200 python bench.py --config $cfg --steps 5 --warmup 2 --no-cpu-baseline --no-cold 2>/dev/null | python -c "
import sys,json
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); k=d['kernels_ms_per_step']
print('cfg$cfg', round(d['value'],1), round(d['ms_per_step'],2), {a:round(b,2) for a,b in k.items()}, d.get('parity_golden'))" || exit 1
done
