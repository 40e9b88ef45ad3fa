#!/bin/bash
# same-box sweep of the dense fill's chunk policy (environment switches only)
cd $GRAFT_REPO_ROOT; export TMPDIR=/tmp
run() {
  timeout -k 10 200 python bench.py --steps 10 --warmup 3 --no-cpu-baseline --no-cold 2>/dev/null | python -c "
import sys,json
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); k=d['kernels_ms_per_step']
print('$1', round(d['ms_per_step'],2), 'fill', round(k['fill_main'],2), 'deferred', round(k['fill_deferred'],2))"
}
run "default"
for c in 2 4 5 6; do EMME_DENSE_MIN_COLS=$c run "mincols=$c"; done
for r in 3 6; do EMME_DENSE_COST_RATIO=$r run "ratio=$r"; done
for t in 0 4000; do EMME_DENSE_MIN_TASKS=$t run "mintasks=$t"; done
run "default"
