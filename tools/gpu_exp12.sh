#!/bin/bash
cd $GRAFT_REPO_ROOT; export TMPDIR=/tmp; O=gpurun_out
echo "== parity pipe2"; EMME_LIB=build/variants/pipe2.so timeout -k 10 300 python -m pytest tests/test_gpu_parity.py -m gpu -q --timeout 200 -k "every_fill_kernel or root_search_same or batch_items" 2>&1 | tail -3
echo "== pipe w2"; EMME_LIB=build/variants/pipe2.so timeout -k 10 100 python tools/iter_profile.py 2 2>&1 | grep "wall\|asm ms" | tail -2
echo "== pipe w3 (spills)"; EMME_LIB=build/variants/pipe3.so timeout -k 10 100 python tools/iter_profile.py 2 2>&1 | grep "wall\|asm ms" | tail -2
echo "== baseline"; timeout -k 10 100 python tools/iter_profile.py 2 2>&1 | grep "wall\|asm ms" | tail -2
