#!/bin/bash
cd $GRAFT_REPO_ROOT; export TMPDIR=/tmp; O=gpurun_out
echo "== parity"; timeout -k 10 300 python -m pytest tests/test_gpu_parity.py -m gpu -q --timeout 200 -k "every_fill_kernel or root_search_same or batch_items" 2>&1 | tail -3
echo "== prefetch w3"; timeout -k 10 100 python tools/iter_profile.py 2 2>&1 | grep "wall\|asm ms" | tail -2
echo "== prefetch w2"; EMME_LIB=build/variants/pfw2.so timeout -k 10 100 python tools/iter_profile.py 2 2>&1 | grep "wall\|asm ms" | tail -2
echo "== no prefetch w3"; EMME_LIB=build/variants/nopf.so timeout -k 10 100 python tools/iter_profile.py 2 2>&1 | grep "wall\|asm ms" | tail -2
