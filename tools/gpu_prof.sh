#!/bin/bash
cd $GRAFT_REPO_ROOT; export TMPDIR=/tmp
timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/r02f_ktrace -- python3 bench.py --steps 2 --warmup 1 --no-cpu-baseline --no-cold > gpurun_out/r02f_ktrace.log 2>&1 || { tail -5 gpurun_out/r02f_ktrace.log; exit 1; }
find gpurun_out/r02f_ktrace -name "*kernel_stats.csv" | head -2
f=$(find gpurun_out/r02f_ktrace -name "*kernel_stats.csv" | head -1); head -12 $f | cut -c1-220
