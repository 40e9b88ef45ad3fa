#!/bin/bash
cd $GRAFT_REPO_ROOT; export TMPDIR=/tmp
timeout -k 10 400 python -m pytest tests/test_gpu_parity.py -m gpu -q --timeout 300 -x -k "trace_solve or lu_ or root_search_is_independent" 2>&1 | tail -2 || exit 1
for cfg in "512 128" "512 64" "384 128" "256 128"; do
  set -- $cfg
  echo "== n=$1 batch=$2 EMME_LU_SPLIT=2 grouped"; EMME_LU_SPLIT=2 timeout -k 10 120 python tools/lu_bench.py $1 $2 2>&1 | tail -1
done
