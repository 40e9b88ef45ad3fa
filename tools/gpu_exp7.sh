#!/bin/bash
cd $GRAFT_REPO_ROOT; export TMPDIR=/tmp; O=gpurun_out
export EMME_DENSE=1 EMME_DENSE_COST_RATIO=100
echo "== stamps"; EMME_LIB=build/variants/stamps.so EMME_DEBUG_STAMPS=1 timeout -k 10 100 python tools/iter_profile.py 1 2>&1 | grep -v amdgpu.ids | grep "stamps\|dense rounds\|asm ms"
