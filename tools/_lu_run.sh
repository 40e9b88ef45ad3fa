set -e
for cfg in "256 128" "512 128"; do
  for b0 in 0 0.0625 0.125 0.25; do echo "b0=$b0"; EMME_LU_B0=$b0 EMME_LU_SPLIT=2 timeout -k 10 120 python3 tools/lu_bench.py $cfg; done
done
for cfg in "256 64" "512 64"; do
  for b0 in 0 0.125; do echo "b0=$b0 (auto split)"; EMME_LU_B0=$b0 timeout -k 10 120 python3 tools/lu_bench.py $cfg; done
done
timeout -k 10 900 python -m pytest tests -x -q -m gpu 2>&1 | tail -5
