#!/bin/bash
cd $GRAFT_REPO_ROOT; export TMPDIR=/tmp; O=gpurun_out
export EMME_DENSE=1
echo "== bfs (grouped loads)"; EMME_DEBUG_STAMPS=1 timeout -k 10 100 python tools/iter_profile.py 2 2>&1 | grep -v amdgpu.ids | grep "wall\|asm ms\|handed"
echo "== bfs upfront loads"; EMME_LIB=build/variants/upfront.so timeout -k 10 100 python tools/iter_profile.py 2 2>&1 | grep "wall\|asm ms"
echo "== deferred counts dense"; EMME_DEBUG=1 timeout -k 10 100 python tools/iter_profile.py 1 2>&1 | grep "integrals deferred" | awk '{print $5, $6, $7, $8}' | tr '\n' ';' | cut -c1-1500
echo; echo "== deferred counts union"; EMME_DENSE=0 EMME_DEBUG=1 timeout -k 10 100 python tools/iter_profile.py 1 2>&1 | grep "integrals deferred" | awk '{print $5, $6, $7, $8}' | tr '\n' ';' | cut -c1-1500
