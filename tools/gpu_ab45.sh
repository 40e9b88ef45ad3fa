#!/bin/bash
# same-box A/B on configs 4 (EM) and the no-cache path: tools/gpu_ab45.sh <variant> ..
cd $GRAFT_REPO_ROOT; export TMPDIR=/tmp
for v in "$@"; do
if [ $v = default ]; then unset EMME_LIB; else export EMME_LIB=$PWD/build/variants/$v.so; fi
timeout -k 10 200 python bench.py --config 4 --steps 5 --warmup 2 --no-cpu-baseline 2>/dev/null | python -c "
import sys,json
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); k=d['kernels_ms_per_step']
print('$v cfg4', round(d['ms_per_step'],2), 'fill', round(k['fill_main'],2), 'lu', round(k['linstep_lu_trace'],2))"
EMME_NODE_CACHE_GB=0 timeout -k 10 200 python bench.py --steps 3 --warmup 1 --no-cpu-baseline --no-cold 2>/dev/null | python -c "
import sys,json
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); k=d['kernels_ms_per_step']
print('$v nocache', round(d['ms_per_step'],2), 'fill', round(k['fill_main'],2))"
EMME_DENSE=0 timeout -k 10 200 python bench.py --steps 5 --warmup 2 --no-cpu-baseline --no-cold 2>/dev/null | python -c "
import sys,json
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); k=d['kernels_ms_per_step']
print('$v union', round(d['ms_per_step'],2), 'fill', round(k['fill_main'],2))"
done
