set -e
EMME_LU_SPLIT=4 timeout -k 10 120 python3 tools/lu_bench.py 256 128
for cfg in "256 128" "512 128" "256 64" "512 64" "256 20"; do timeout -k 10 120 python3 tools/lu_bench.py $cfg; done
timeout -k 10 600 python -m pytest tests -x -q -m gpu -k "trace_solve" 2>&1 | tail -3
timeout -k 10 400 python bench.py > gpurun_out/bench_split.log 2>&1; python3 -c "
import json
l=[x for x in open('gpurun_out/bench_split.log') if x.startswith('{')][-1]; d=json.loads(l); print(d['value'], d['ms_per_step'], d['converged_fraction'], d['kernels_ms_per_step'])"
