#!/bin/bash
cd $GRAFT_REPO_ROOT; export TMPDIR=/tmp; O=gpurun_out
export EMME_DENSE=1 EMME_DENSE_COST_RATIO=100
for mc in 1 2 3; do echo "== min_cols $mc"; EMME_DENSE_MIN_COLS=$mc timeout -k 10 100 python tools/iter_profile.py 2 2>&1 | grep "wall\|asm ms" | tail -2; done
