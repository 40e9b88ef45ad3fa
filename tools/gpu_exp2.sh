#!/bin/bash
cd $GRAFT_REPO_ROOT; export TMPDIR=/tmp; O=gpurun_out
export EMME_DENSE=1
echo "== parity of dense modes"; timeout -k 10 300 python -m pytest tests/test_gpu_parity.py -m gpu -q --timeout 200 -k "every_fill_kernel or root_search_same" 2>&1 | tail -5
echo "== bfs (w2)"; timeout -k 10 100 python tools/iter_profile.py 2 2>&1 | grep -v amdgpu.ids
echo "== bfs w3 (spills)"; EMME_LIB=build/variants/bfs3.so timeout -k 10 100 python tools/iter_profile.py 2 2>&1 | grep "wall\|asm ms"
echo "== dfs"; EMME_DENSE_BFS=0 timeout -k 10 100 python tools/iter_profile.py 2 2>&1 | grep "wall\|asm ms"
echo "== bfs all mfma"; EMME_DENSE_MIN_COLS=1 timeout -k 10 100 python tools/iter_profile.py 2 2>&1 | grep "wall\|asm ms"
echo "== bfs mincols 2"; EMME_DENSE_MIN_COLS=2 timeout -k 10 100 python tools/iter_profile.py 2 2>&1 | grep "wall\|asm ms\|dense rounds"
echo "== ktrace bfs"; cd /tmp && timeout -k 10 200 rocprofv3 --kernel-trace --output-format csv -d $GRAFT_REPO_ROOT/$O/r2_kt_bfs -- python3 $GRAFT_REPO_ROOT/tools/iter_profile.py 1 > $GRAFT_REPO_ROOT/$O/r2_kt_bfs.log 2>&1; echo rc $?
