#!/bin/bash
# same-box A/B of library variants on bench --config 4 only: tools/gpu_ab4.sh <variant> ...
cd $GRAFT_REPO_ROOT; export TMPDIR=/tmp
for rep in 1 2; do
for v in "$@"; do
if [ $v = default ]; then unset EMME_LIB; else export EMME_LIB=$PWD/build/variants/$v.so; fi
timeout -k 10 200 python bench.py --config 4 --steps 5 --warmup 2 --no-cpu-baseline --no-cold 2>/dev/null | python -c "
import sys,json
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); k=d['kernels_ms_per_step']
print('$v cfg4', round(d['value'],1), round(d['ms_per_step'],2), 'fill', round(k['fill_main'],2), 'lu', round(k['linstep_lu_trace'],2), 'other', round(k['other'],2), d.get('parity_golden'))"
done; done
