#!/bin/bash
cd $GRAFT_REPO_ROOT; export TMPDIR=/tmp
timeout -k 10 400 python -m pytest tests/test_gpu_parity.py tests/test_gpu_round2.py -m gpu -q --timeout 300 -x -k "every_fill_kernel or root_search_same or batch_items or cfg3 or deferred or coop or settle" 2>&1 | tail -2 || exit 1
timeout -k 10 100 python tools/iter_profile.py 2 2>&1 | grep "wall\|asm ms" | tail -2
