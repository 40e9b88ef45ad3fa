#!/bin/bash
# n = 512 LU: how the launch time depends on the matrices in flight and the workgroups per matrix
cd $GRAFT_REPO_ROOT; export TMPDIR=/tmp
for cfg in "128 2" "64 2" "64 4" "32 2" "32 4" "32 8" "16 8" "16 16"; do
set -- $cfg
EMME_LU_SPLIT=$2 timeout -k 10 120 python tools/lu_bench.py 512 $1 2>&1 | tail -1 | sed "s/^/nwg=$2 /" || exit 1
done
