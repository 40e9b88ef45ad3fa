#!/bin/bash
cd $GRAFT_REPO_ROOT; export TMPDIR=/tmp; O=gpurun_out; mkdir -p $O
timeout -k 10 700 python -m pytest tests -m gpu -q --timeout 300 --deselect tests/test_gpu_round2.py::test_cfg3_headline_shape_matches_reference_chains > $O/r2_gpu_tests4.log 2>&1; rc=$?; echo pytest rc $rc; tail -30 $O/r2_gpu_tests4.log | cut -c1-400
if [ $rc -gt 1 ]; then exit 1; fi
echo "== bench under torchrun, 1 rank (dist path, C++ RCCL gather)"
timeout -k 10 300 python -m torch.distributed.run --nnodes=1 --nproc-per-node 1 --master-addr 127.0.0.1 --master-port 29611 bench.py --gpus 1 --steps 3 --warmup 1 --no-cpu-baseline > $O/r2_bench_dist1.log 2> $O/r2_bench_dist1.err; echo rc $?; tail -c 600 $O/r2_bench_dist1.log; grep -i "warn\|error" $O/r2_bench_dist1.err | head -5
echo "== bench --config 4"
timeout -k 10 300 python bench.py --config 4 --steps 3 --warmup 1 > $O/r2_bench_cfg4.log 2> $O/r2_bench_cfg4.err; echo rc $?; cut -c1-900 $O/r2_bench_cfg4.log; tail -3 $O/r2_bench_cfg4.err
echo "== bench --config 5"
timeout -k 10 400 python bench.py --config 5 --steps 2 --warmup 1 > $O/r2_bench_cfg5.log 2> $O/r2_bench_cfg5.err; echo rc $?; cut -c1-900 $O/r2_bench_cfg5.log; tail -3 $O/r2_bench_cfg5.err
echo "== size sweep"
timeout -k 10 400 python tools/size_sweep.py 256 512 1024 > $O/r2_size_sweep.log 2>&1; echo rc $?; cat $O/r2_size_sweep.log | grep -v amdgpu.ids
echo "== pmc"
bash tools/pmc_collect.sh r02
echo "== ktrace stats of bench"
cd /tmp && timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d $GRAFT_REPO_ROOT/$O/r02_ktrace -- python3 $GRAFT_REPO_ROOT/bench.py --steps 2 --warmup 1 --no-cpu-baseline --no-cold > $GRAFT_REPO_ROOT/$O/r02_ktrace.log 2>&1; echo rc $?
