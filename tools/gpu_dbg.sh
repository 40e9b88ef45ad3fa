#!/bin/bash
cd $GRAFT_REPO_ROOT; export TMPDIR=/tmp
for rep in 1 2; do
for v in default nopairs; do
if [ $v = default ]; then unset EMME_LIB; else export EMME_LIB=$PWD/build/variants/$v.so; fi
echo "== $v"; timeout -k 10 100 python tools/iter_profile.py 3 2>&1 | grep "asm ms\|wall" | tail -2 | tr '\n' ' '; echo
done; done
