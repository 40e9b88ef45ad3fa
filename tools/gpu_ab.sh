#!/bin/bash
# same-box A/B of library variants on the bench's steady state: tools/gpu_ab.sh <variant> [<variant> ..]
# ("default" = the in-tree library; others = build/variants/<name>.so); two rounds, ms per step and fill per step
cd $GRAFT_REPO_ROOT; export TMPDIR=/tmp
for rep in 1 2; do
for v in "$@"; do
if [ $v = default ]; then unset EMME_LIB; else export EMME_LIB=$PWD/build/variants/$v.so; fi
timeout -k 10 200 python bench.py --steps 10 --warmup 3 --no-cpu-baseline --no-cold 2>/dev/null | python -c "
import sys,json
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); k=d['kernels_ms_per_step']
print('$v', round(d['ms_per_step'],2), 'fill', round(k['fill_main'],2), 'deferred', round(k['fill_deferred'],2), 'lu', round(k['linstep_lu_trace'],2))"
done; done
