#!/bin/bash
cd $GRAFT_REPO_ROOT; export TMPDIR=/tmp
EMME_LIB=$PWD/build/variants/dstamps.so EMME_DEBUG_STAMPS=1 timeout -k 10 200 python tools/iter_profile.py 1 > gpurun_out/dbg_iter.out 2> gpurun_out/dbg_iter.err
grep "dense stamps" gpurun_out/dbg_iter.err | tail -1 | cut -c1-400
