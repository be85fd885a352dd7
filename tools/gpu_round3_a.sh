#!/bin/bash
# round 3, GPU call A: the Model-driven training step (tests + bench + N ranks on one GPU)
cd /root/repo
mkdir -p gpurun_out
timeout 1500 python -m pytest tests/test_model_gpu.py tests/test_loss_curve_gpu.py tests/test_dropin_gpu.py -x -q > gpurun_out/r3a_tests.log 2>&1
echo "tests rc=$?" >> gpurun_out/r3a_tests.log
tail -5 gpurun_out/r3a_tests.log
timeout 600 python bench.py --extra none --no-cpu-baseline > gpurun_out/r3a_bench.json 2> gpurun_out/r3a_bench.err
echo "bench rc=$?"; tail -3 gpurun_out/r3a_bench.err; head -c 1500 gpurun_out/r3a_bench.json
timeout 600 bash tools/bench_ranks_one_gpu.sh 2 > gpurun_out/r3a_ranks.log 2>&1
echo "ranks rc=$?"; tail -12 gpurun_out/r3a_ranks.log
