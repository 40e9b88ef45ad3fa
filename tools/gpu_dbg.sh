#!/bin/bash
cd $GRAFT_REPO_ROOT; export TMPDIR=/tmp
EMME_DEBUG=1 timeout -k 10 200 python tools/iter_profile.py 2 > gpurun_out/dbg_iter.out 2> gpurun_out/dbg_iter.err
grep -n "deferrals of class\|node cache:" gpurun_out/dbg_iter.err | cut -c1-300 | tail -40
