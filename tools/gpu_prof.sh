#!/bin/bash
# round 3: PMC summaries + kernel stats + bench lines of all three bench configurations -> gpurun_out/r03_*
cd $GRAFT_REPO_ROOT; export TMPDIR=/tmp
for cfg in 3 4 5; do
  tag=r03; [ $cfg != 3 ] && tag=r03_cfg$cfg
  bash tools/pmc_collect.sh $tag $cfg > gpurun_out/${tag}_collect.log 2>&1 || { tail -5 gpurun_out/${tag}_collect.log; exit 1; }
  extra=""; [ $cfg != 3 ] && extra="--config $cfg"
  timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/${tag}_ktrace -- python3 bench.py --steps 2 --warmup 1 --no-cpu-baseline --no-cold $extra > gpurun_out/${tag}_ktrace.log 2>&1 || { tail -5 gpurun_out/${tag}_ktrace.log; exit 1; }
  f=$(find gpurun_out/${tag}_ktrace -name "*kernel_stats.csv" | head -1); cp $f gpurun_out/${tag}_kernel_stats.csv; head -6 $f | cut -c1-160
done
