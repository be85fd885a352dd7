#!/bin/bash
cd /root/repo
mkdir -p gpurun_out
timeout 2400 python -m pytest tests -m gpu -x -q > gpurun_out/r3b_tests.log 2>&1
echo "tests rc=$?" >> gpurun_out/r3b_tests.log
tail -8 gpurun_out/r3b_tests.log
timeout 600 python bench.py --extra none --no-cpu-baseline > gpurun_out/r3b_bench.json 2> gpurun_out/r3b_bench.err
echo "bench rc=$?"; tail -3 gpurun_out/r3b_bench.err; head -c 2500 gpurun_out/r3b_bench.json
timeout 600 bash tools/bench_ranks_one_gpu.sh 2 > gpurun_out/r3b_ranks.log 2>&1
echo "ranks rc=$?"; tail -12 gpurun_out/r3b_ranks.log
