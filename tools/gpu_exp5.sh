#!/bin/bash
cd $GRAFT_REPO_ROOT; export TMPDIR=/tmp; O=gpurun_out
export EMME_DENSE=1
echo "== parity of dense modes"; timeout -k 10 300 python -m pytest tests/test_gpu_parity.py -m gpu -q --timeout 200 -k "every_fill_kernel or root_search_same" 2>&1 | tail -3
for mt in 8000 16000 4000 1; do for cr in 1.5 3.0 100; do echo "== min_tasks $mt cost_ratio $cr"; EMME_DENSE_MIN_TASKS=$mt EMME_DENSE_COST_RATIO=$cr timeout -k 10 100 python tools/iter_profile.py 2 2>&1 | grep "wall\|asm ms" | tail -2; done; done
echo "== ktrace"; cd /tmp && timeout -k 10 200 rocprofv3 --kernel-trace --output-format csv -d $GRAFT_REPO_ROOT/$O/r2_kt_bfs3 -- python3 $GRAFT_REPO_ROOT/tools/iter_profile.py 1 > $GRAFT_REPO_ROOT/$O/r2_kt_bfs3.log 2>&1; echo rc $?
