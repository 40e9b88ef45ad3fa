#!/bin/bash
cd $GRAFT_REPO_ROOT; export TMPDIR=/tmp
for v in cw3 cw4; do
echo "== $v"; EMME_LIB=$PWD/build/variants/$v.so timeout -k 10 100 python tools/iter_profile.py 2 2>&1 | grep "asm ms" | tail -1
done
echo "== default"; timeout -k 10 100 python tools/iter_profile.py 2 2>&1 | grep "asm ms" | tail -1
