#!/bin/bash
# same-box A/B of an environment variable on the headline bench: tools/gpu_env_ab.sh VAR v1 v2 ...  ("-" = unset)
cd $GRAFT_REPO_ROOT; export TMPDIR=/tmp
var=$1; shift
for rep in 1 2; do
for v in "$@"; do
if [ "$v" = "-" ]; then unset $var; else export $var=$v; fi
timeout -k 10 200 python bench.py --steps 8 --warmup 2 --no-cpu-baseline --no-cold 2>/dev/null | python -c "
import sys,json
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); k=d['kernels_ms_per_step']
print('$var=$v', round(d['value'],1), round(d['ms_per_step'],2), 'fill', round(k['fill_main'],2), 'lu', round(k['linstep_lu_trace'],2), d['parity_golden']['iteration_count_mismatches'], d['parity_golden']['max_rel_err'])"
done; done
