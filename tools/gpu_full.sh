#!/bin/bash
# full GPU check of a build: every -m gpu test, then the driver's bench command
cd $GRAFT_REPO_ROOT; export TMPDIR=/tmp
timeout -k 10 900 python -m pytest tests -m gpu -x -q --timeout 600 2>&1 | tail -3 || exit 1
timeout -k 10 500 python bench.py --gpus 1 --steps 20 --warmup 5 2> gpurun_out/bench_full.err | tee gpurun_out/bench_full.json
