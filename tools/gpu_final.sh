#!/bin/bash
# round 3, final measurement pass ON THE GPU BOX: for BASELINE configs[2], [3], [4] (bench --config 3, 4, 5):
#   PMC passes (tools/pmc_collect.sh) -> summary into profiles/ (so that the bench line of the same build is not
#   stale), kernel trace of a short bench run, then the full bench line with cpu_baseline.
# Everything lands under gpurun_out/final/ (copied into profiles/ by hand afterwards).
cd $GRAFT_REPO_ROOT; export TMPDIR=/tmp
mkdir -p gpurun_out/final
for cfg in 3 4 5; do
  tag=r03; [ $cfg != 3 ] && tag=r03_cfg$cfg
  bash tools/pmc_collect.sh $tag $cfg > gpurun_out/final/${tag}_collect.log 2>&1 || { tail -5 gpurun_out/final/${tag}_collect.log; exit 1; }
  cp gpurun_out/${tag}_pmc_summary.json profiles/${tag}_pmc_summary.json
  cp gpurun_out/${tag}_pmc_summary.json gpurun_out/final/
  extra=""; [ $cfg != 3 ] && extra="--config $cfg"
  rm -rf gpurun_out/${tag}_ktrace
  timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/${tag}_ktrace -- python3 bench.py --steps 2 --warmup 1 --no-cpu-baseline --no-cold $extra > gpurun_out/final/${tag}_ktrace.log 2>&1 || { tail -5 gpurun_out/final/${tag}_ktrace.log; exit 1; }
  f=$(find gpurun_out/${tag}_ktrace -name "*kernel_stats.csv" | head -1); cp $f gpurun_out/final/${tag}_kernel_stats.csv; head -4 $f | cut -c1-140
  timeout -k 10 600 python bench.py --gpus 1 --steps 20 --warmup 5 $extra > gpurun_out/final/${tag}_bench.json.log 2> gpurun_out/final/${tag}_bench.err || { tail -5 gpurun_out/final/${tag}_bench.err; exit 1; }
  python - <<PY
import json
d=json.loads(open("gpurun_out/final/${tag}_bench.json.log").read().strip().splitlines()[-1])
r=d["roofline"]; print("cfg$cfg", round(d["value"],1), round(d["ms_per_step"],2), r.get("kernel"), r.get("bound"), r.get("frac"), r.get("pmc_stale"), d.get("cpu_baseline",{}).get("value"))
PY
done
