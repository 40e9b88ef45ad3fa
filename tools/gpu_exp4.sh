#!/bin/bash
cd $GRAFT_REPO_ROOT; export TMPDIR=/tmp; O=gpurun_out
export EMME_DENSE=1
echo "== parity of dense modes"; timeout -k 10 300 python -m pytest tests/test_gpu_parity.py -m gpu -q --timeout 200 -k "every_fill_kernel or root_search_same" 2>&1 | tail -3
echo "== w3 default"; timeout -k 10 100 python tools/iter_profile.py 2 2>&1 | grep "wall\|asm ms"
echo "== w2"; EMME_LIB=build/variants/w2.so timeout -k 10 100 python tools/iter_profile.py 2 2>&1 | grep "wall\|asm ms"
echo "== w4 (spills)"; EMME_LIB=build/variants/w4.so timeout -k 10 100 python tools/iter_profile.py 2 2>&1 | grep "wall\|asm ms"
echo "== w3 upfront loads"; EMME_LIB=build/variants/upfront.so timeout -k 10 100 python tools/iter_profile.py 2 2>&1 | grep "wall\|asm ms"
echo "== w3 mincols 2"; EMME_DENSE_MIN_COLS=2 timeout -k 10 100 python tools/iter_profile.py 2 2>&1 | grep "wall\|asm ms"
echo "== w3 mincols 5"; EMME_DENSE_MIN_COLS=5 timeout -k 10 100 python tools/iter_profile.py 2 2>&1 | grep "wall\|asm ms"
echo "== union"; EMME_DENSE=0 timeout -k 10 100 python tools/iter_profile.py 2 2>&1 | grep "wall\|asm ms"
echo "== ktrace"; cd /tmp && timeout -k 10 200 rocprofv3 --kernel-trace --output-format csv -d $GRAFT_REPO_ROOT/$O/r2_kt_bfs2 -- python3 $GRAFT_REPO_ROOT/tools/iter_profile.py 1 > $GRAFT_REPO_ROOT/$O/r2_kt_bfs2.log 2>&1; echo rc $?
