#!/bin/bash
cd $GRAFT_REPO_ROOT; export TMPDIR=/tmp
for v in p2 p3; do
echo "== $v"
EMME_LIB=$PWD/build/variants/$v.so EMME_DEBUG_STAMPS=1 timeout -k 10 200 python tools/iter_profile.py 1 > gpurun_out/dbg_iter.out 2> gpurun_out/dbg_iter_$v.err
grep "dense launch" gpurun_out/dbg_iter_$v.err | tail -23 | awk 'NR%4==3' | cut -c1-260
done
echo "== default"; timeout -k 10 100 python tools/iter_profile.py 2 2>&1 | grep "asm ms" | tail -1
for v in q2 q3 w3; do
echo "== $v"; EMME_LIB=$PWD/build/variants/$v.so timeout -k 10 100 python tools/iter_profile.py 2 2>&1 | grep "asm ms" | tail -1
done
