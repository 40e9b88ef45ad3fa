#!/bin/bash
cd $GRAFT_REPO_ROOT; export TMPDIR=/tmp
timeout -k 10 1000 python tools/lu_soak.py 40 2>&1 | tee gpurun_out/r02_lu_soak.txt | tail -14
