#!/bin/bash
cd $GRAFT_REPO_ROOT; export TMPDIR=/tmp; O=gpurun_out; mkdir -p $O
EMME_TEST_DUMP=$O/r2_cfg3_gpu.npz timeout -k 10 700 python -m pytest tests -m gpu -q --timeout 300 > $O/r2_gpu_tests6.log 2>&1; rc=$?; echo pytest rc $rc; tail -12 $O/r2_gpu_tests6.log | cut -c1-300
if [ $rc -gt 1 ]; then exit 1; fi
python - <<'PY'
import numpy as np, os, sys
sys.path.insert(0,'.')
import emme_amd
from oracle.binding import example_stellarator
z=np.load('tests/golden/stellarator_k8.npz')
for n in (32,48):
    g,want=z[f"n{n}_guesses"],z[f"n{n}_iterates"]
    with emme_amd.Context(emme_amd.params_from_dict(example_stellarator(npoints=n))) as ctx:
        r,it,inf,its=ctx.solve_roots(g,tol=0.0,step_limit=7,want_iterates=True)
    print("K8 n",n,"max rel err per chain",(np.abs(its[:,:8]-want)/np.abs(want)).max(axis=1))
gp=np.load('gpurun_out/r2_cfg3_gpu.npz'); zz=np.load('tests/golden/cfg3_chains.npz')
conv=zz['converged']==1
e=np.abs(gp['roots']-zz['roots'])/np.abs(zz['roots'])
print("cfg3 root rel err: stable max",e[conv & (np.arange(128)!=30)].max(),"chain30",e[30])
PY
timeout -k 10 300 python bench.py --gpus 1 --steps 20 --warmup 5 > $O/r02_bench.json.log 2> $O/r02_bench.err; echo bench rc $?; python3 -c "
import json
d=json.loads(open('$O/r02_bench.json.log').read().strip().splitlines()[-1])
print({k:d[k] for k in ('value','ms_per_step','kernels_ms_per_step','parity_golden','parity_sample','speedup_vs_cpu_baseline')}); print(d['cpu_baseline'])"
