#!/bin/bash
cd $GRAFT_REPO_ROOT; export TMPDIR=/tmp; O=gpurun_out
echo "== parity of dense modes"; timeout -k 10 300 python -m pytest tests/test_gpu_parity.py -m gpu -q --timeout 200 -k "every_fill_kernel or root_search_same or batch_items" 2>&1 | tail -4
for mc in 3 2 4; do echo "== min_cols $mc"; EMME_DENSE_MIN_COLS=$mc timeout -k 10 100 python tools/iter_profile.py 2 2>&1 | grep "wall\|asm ms" | tail -2; done
