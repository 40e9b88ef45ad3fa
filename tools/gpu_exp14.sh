#!/bin/bash
cd $GRAFT_REPO_ROOT; export TMPDIR=/tmp
echo "== stamps w3"; EMME_LIB=build/variants/stamps.so EMME_DEBUG_STAMPS=1 timeout -k 10 100 python tools/iter_profile.py 1 2>&1 | grep "stamps\|dense rounds\|asm ms" | tail -3
echo "== stamps w1 (one wave per SIMD: a wave alone)"; EMME_LIB=build/variants/stamps2.so EMME_DEBUG_STAMPS=1 timeout -k 10 100 python tools/iter_profile.py 1 2>&1 | grep "stamps\|dense rounds\|asm ms" | tail -3
