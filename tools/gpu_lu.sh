#!/bin/bash
cd $GRAFT_REPO_ROOT; export TMPDIR=/tmp
timeout -k 10 400 python -m pytest tests/test_gpu_parity.py -m gpu -q --timeout 300 -x -k "trace_solve or lu_ or root_search_is_independent" 2>&1 | tail -4 || exit 1
for cfg in "512 128" "512 64" "448 128" "384 128" "256 128"; do
  set -- $cfg
  for g in 0 16; do
    echo "== n=$1 batch=$2 EMME_LU_SPLIT=2 group=$g"; EMME_LU_SPLIT=2 EMME_LU_GROUP=$g timeout -k 10 120 python tools/lu_bench.py $1 $2 2>&1 | tail -1
  done
done
export EMME_LIB=$PWD/build/variants/lustamps.so EMME_DEBUG_STAMPS=1
for cfg in "512 128" "256 128"; do
  set -- $cfg
  echo "== stamps n=$1 batch=$2 grouped"; EMME_LU_SPLIT=2 EMME_LU_GROUP=16 timeout -k 10 120 python tools/lu_bench.py $1 $2 2>&1 | tail -3
done
