#!/bin/bash
cd $GRAFT_REPO_ROOT; export TMPDIR=/tmp
export EMME_LIB=$PWD/build/variants/lustamps.so EMME_DEBUG_STAMPS=1
for cfg in "512 128" "512 64" "256 128"; do
  set -- $cfg
  for g in 0 16; do
    echo "== n=$1 batch=$2 EMME_LU_SPLIT=2 group=$g"; EMME_LU_SPLIT=2 EMME_LU_GROUP=$g timeout -k 10 120 python tools/lu_bench.py $1 $2 2>&1 | tail -3
  done
done
echo "== n=512 batch=32 auto"; timeout -k 10 120 python tools/lu_bench.py 512 32 2>&1 | tail -3
echo "== n=256 batch=20 auto"; timeout -k 10 120 python tools/lu_bench.py 256 20 2>&1 | tail -3
