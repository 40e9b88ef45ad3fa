#!/bin/bash
cd $GRAFT_REPO_ROOT; export TMPDIR=/tmp
echo "== parity split"; EMME_LIB=build/variants/split.so timeout -k 10 300 python -m pytest tests/test_gpu_parity.py -m gpu -q --timeout 200 -k "every_fill_kernel" 2>&1 | tail -2
echo "== split acc"; EMME_LIB=build/variants/split.so timeout -k 10 100 python tools/iter_profile.py 2 2>&1 | grep "wall\|asm ms" | tail -2
echo "== split acc stamps"; EMME_LIB=build/variants/splitst.so EMME_DEBUG_STAMPS=1 timeout -k 10 100 python tools/iter_profile.py 1 2>&1 | grep "stamps" | tail -1
echo "== baseline"; timeout -k 10 100 python tools/iter_profile.py 2 2>&1 | grep "wall\|asm ms" | tail -2
