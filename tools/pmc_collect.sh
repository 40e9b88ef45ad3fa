#!/bin/bash
# rocprofv3 --pmc passes over one bench workload (tools/iter_profile.py 3: one preparing + three more root searches of the
# 128-guess lattice) + the kernel trace of bench.py; run ON THE GPU BOX from the repo root:
#   bash tools/pmc_collect.sh <tag> [config 3|4|5]  ->  gpurun_out/<tag>_pmc/pass*/..., gpurun_out/<tag>_pmc_summary.json,
#                                            gpurun_out/<tag>_ktrace/..._kernel_stats.csv
# Counters go in their own runs (--pmc only); FETCH_SIZE and WRITE_SIZE need a pass each (TCC slots).
set -u
tag=${1:-r03}
cfg=${2:-3}
prog=tools/iter_profile.py
out=gpurun_out/${tag}_pmc
mkdir -p "$out"
export TMPDIR=/tmp
passes=(
 "SQ_INSTS_VALU SQ_INSTS_VALU_ADD_F64 SQ_INSTS_VALU_MUL_F64 SQ_INSTS_VALU_FMA_F64 SQ_INSTS_VALU_TRANS_F64 SQ_INSTS_VALU_MFMA_MOPS_F64 SQ_THREAD_CYCLES_VALU SQ_INSTS_VALU_FLOPS_FP64"
 "SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_BUSY_CYCLES SQ_VALU_MFMA_BUSY_CYCLES SQ_WAVES"
 "FETCH_SIZE"
 "WRITE_SIZE"
 "TCC_HIT_sum TCC_MISS_sum"
 "SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_INSTS_LDS SQ_INSTS_SALU SQ_INSTS_VALU_INT32 SQ_INSTS_VALU_INT64 SQ_INSTS_VALU_CVT SQ_ACTIVE_INST_LDS"
 "TCP_TOTAL_CACHE_ACCESSES_sum TCP_TCC_READ_REQ_sum GRBM_GUI_ACTIVE"
)
i=0
for p in "${passes[@]}"; do
  i=$((i+1))
  timeout -k 10 300 rocprofv3 --pmc $p --output-format csv -d "$out/pass$i" -- python3 "$prog" 3 "$cfg" > "$out/pass$i.log" 2>&1
  rc=$?
  echo "pass $i ($p): rc $rc"
  if [ $rc -ge 124 ]; then echo "pass $i timed out: stopping"; exit 1; fi
done
python3 tools/pmc_summary.py "$out" "gpurun_out/${tag}_pmc_summary.json" 4 > "$out/summary.log" 2>&1
tail -5 "$out/summary.log"
