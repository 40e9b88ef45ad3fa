#!/bin/bash
# same-box A/B of LU variants: tools/gpu_lu_ab.sh <variant> ...   (n = 512 x 128, n = 256 x 128, n = 256 x 32)
cd $GRAFT_REPO_ROOT; export TMPDIR=/tmp
variants="$*"
for rep in 1 2; do
for v in $variants; do
if [ $v = default ]; then unset EMME_LIB; else export EMME_LIB=$PWD/build/variants/$v.so; fi
for cfg in "512 128" "256 128" "256 32"; do
set -- $cfg; n_=$1; nb_=$2
timeout -k 10 120 python tools/lu_bench.py $n_ $nb_ 2>&1 | tail -1 | sed "s/^/$v /" || exit 1
done; done; done
