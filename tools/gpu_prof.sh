#!/bin/bash
cd $GRAFT_REPO_ROOT; export TMPDIR=/tmp
bash tools/pmc_collect.sh r02r > gpurun_out/r02r_collect.log 2>&1 || { tail -5 gpurun_out/r02r_collect.log; exit 1; }
timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/r02r_ktrace -- python3 bench.py --steps 2 --warmup 1 --no-cpu-baseline --no-cold > gpurun_out/r02r_ktrace.log 2>&1 || { tail -5 gpurun_out/r02r_ktrace.log; exit 1; }
f=$(find gpurun_out/r02r_ktrace -name "*kernel_stats.csv" | head -1); cp $f gpurun_out/r02r_kernel_stats.csv; head -8 $f | cut -c1-200
