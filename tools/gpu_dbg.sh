#!/bin/bash
cd $GRAFT_REPO_ROOT; export TMPDIR=/tmp
for v in dstamps; do
EMME_LIB=$PWD/build/variants/$v.so EMME_DEBUG_STAMPS=1 timeout -k 10 200 python tools/iter_profile.py 1 > gpurun_out/dbg_iter.out 2> gpurun_out/dbg_iter.err
echo "== $v"; grep "dense stamps" gpurun_out/dbg_iter.err | tail -1 | cut -c1-400
done
echo "== product"; timeout -k 10 100 python tools/iter_profile.py 2 2>&1 | grep "asm ms" | tail -1
