#!/bin/bash
cd $GRAFT_REPO_ROOT; export TMPDIR=/tmp; O=gpurun_out; mkdir -p $O
timeout -k 10 700 python -m pytest tests -m gpu -q --timeout 300 > $O/r2_gpu_tests8.log 2>&1; rc=$?; echo pytest rc $rc; tail -25 $O/r2_gpu_tests8.log | cut -c1-300
if [ $rc -gt 1 ]; then exit 1; fi
timeout -k 10 300 python bench.py --no-cpu-baseline --steps 10 --warmup 2 > $O/r2_bench8.log 2> $O/r2_bench8.err; echo bench rc $?; python3 -c "
import json
d=json.loads(open('$O/r2_bench8.log').read().strip().splitlines()[-1])
print({k:d[k] for k in ('value','ms_per_step','kernels_ms_per_step','parity_golden','node_cache_gib')})"; tail -3 $O/r2_bench8.err
