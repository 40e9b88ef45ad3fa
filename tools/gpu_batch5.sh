#!/bin/bash
cd $GRAFT_REPO_ROOT; export TMPDIR=/tmp; O=gpurun_out; mkdir -p $O
timeout -k 10 700 python -m pytest tests -m gpu -q --timeout 300 --deselect tests/test_gpu_round2.py::test_cfg3_headline_shape_matches_reference_chains > $O/r2_gpu_tests5.log 2>&1; rc=$?; echo pytest rc $rc; tail -12 $O/r2_gpu_tests5.log | cut -c1-300
if [ $rc -gt 1 ]; then exit 1; fi
timeout -k 10 300 python bench.py --no-cpu-baseline --steps 10 --warmup 2 > $O/r2_bench5.log 2> $O/r2_bench5.err; echo bench rc $?; python3 -c "
import json,sys
d=json.loads(open('$O/r2_bench5.log').read().strip().splitlines()[-1])
print({k:d[k] for k in ('value','ms_per_step','kernels_ms_per_step','parity_golden','cold')})
print(d['roofline'])
"
timeout -k 10 100 python tools/size_sweep.py 1024 2>&1 | grep -v amdgpu.ids
cd /tmp && timeout -k 10 200 rocprofv3 --kernel-trace --output-format csv -d $GRAFT_REPO_ROOT/$O/r2_kt5 -- python3 $GRAFT_REPO_ROOT/tools/iter_profile.py 1 > $GRAFT_REPO_ROOT/$O/r2_kt5.log 2>&1; echo rc $?
