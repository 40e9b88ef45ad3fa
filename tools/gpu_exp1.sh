#!/bin/bash
# one-off experiment batch (GPU box)
cd $GRAFT_REPO_ROOT; export TMPDIR=/tmp; O=gpurun_out
echo "== default (w2)"; timeout -k 10 100 python tools/iter_profile.py 2 2>&1 | grep -v amdgpu.ids
echo "== stamps"; EMME_LIB=build/variants/stamps.so EMME_DEBUG_STAMPS=1 timeout -k 10 100 python tools/iter_profile.py 1 2>&1 | grep -v amdgpu.ids
echo "== w3"; EMME_LIB=build/variants/w3.so timeout -k 10 100 python tools/iter_profile.py 2 2>&1 | grep "wall\|asm ms"
echo "== w4"; EMME_LIB=build/variants/w4.so timeout -k 10 100 python tools/iter_profile.py 2 2>&1 | grep "wall\|asm ms"
echo "== all sparse"; EMME_DENSE_MIN_COLS=17 timeout -k 10 100 python tools/iter_profile.py 2 2>&1 | grep "wall\|asm ms\|dense rounds"
echo "== all mfma"; EMME_DENSE_MIN_COLS=1 timeout -k 10 100 python tools/iter_profile.py 2 2>&1 | grep "wall\|asm ms\|dense rounds"
echo "== probe"; timeout -k 10 200 python tests/analysis/chain30_probe.py 2>&1 | grep -v amdgpu.ids
echo "== ktrace dense"; cd /tmp && timeout -k 10 200 rocprofv3 --kernel-trace --output-format csv -d $GRAFT_REPO_ROOT/$O/r2_kt_dense -- python3 $GRAFT_REPO_ROOT/tools/iter_profile.py 1 > $GRAFT_REPO_ROOT/$O/r2_kt_dense.log 2>&1; echo rc $?
EMME_DENSE=0 timeout -k 10 200 rocprofv3 --kernel-trace --output-format csv -d $GRAFT_REPO_ROOT/$O/r2_kt_union -- python3 $GRAFT_REPO_ROOT/tools/iter_profile.py 1 > $GRAFT_REPO_ROOT/$O/r2_kt_union.log 2>&1; echo rc $?
