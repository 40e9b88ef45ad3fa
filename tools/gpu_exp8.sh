#!/bin/bash
cd $GRAFT_REPO_ROOT; export TMPDIR=/tmp; O=gpurun_out
export EMME_DENSE=1 EMME_DENSE_COST_RATIO=100
echo "== timing"; timeout -k 10 100 python tools/iter_profile.py 2 2>&1 | grep "wall\|asm ms\|dense rounds" | tail -3
bash tools/pmc_collect.sh r02d
python3 - <<'PY'
import json
d=json.load(open('gpurun_out/r02d_pmc_summary.json'))['kernels']
for k,v in d.items():
    if 'dense' in k or 'coop' in k or 'btab' in k:
        print(k, {a:(round(b,4) if isinstance(b,float) else b) for a,b in v.items() if not a.startswith(('SQ_','TCC','TCP','GRBM','FETCH','WRITE')) or a.endswith('cycle')})
PY
