#!/bin/bash
cd $GRAFT_REPO_ROOT; export TMPDIR=/tmp; O=gpurun_out
export EMME_DENSE=1
echo "== parity of dense modes"; timeout -k 10 300 python -m pytest tests/test_gpu_parity.py -m gpu -q --timeout 200 -k "every_fill_kernel or root_search_same or batch_items" 2>&1 | tail -8
for cr in 3.0 100; do for mc in 3 2 5; do echo "== cost_ratio $cr min_cols $mc"; EMME_DENSE_MIN_COLS=$mc EMME_DENSE_COST_RATIO=$cr timeout -k 10 100 python tools/iter_profile.py 2 2>&1 | grep "wall\|asm ms\|dense rounds" | tail -3; done; done
echo "== ktrace"; cd /tmp && timeout -k 10 200 rocprofv3 --kernel-trace --output-format csv -d $GRAFT_REPO_ROOT/$O/r2_kt_bfs4 -- python3 $GRAFT_REPO_ROOT/tools/iter_profile.py 1 > $GRAFT_REPO_ROOT/$O/r2_kt_bfs4.log 2>&1; echo rc $?
