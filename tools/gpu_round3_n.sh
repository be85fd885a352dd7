#!/bin/bash
cd /root/repo
mkdir -p gpurun_out
timeout 1500 python -m pytest tests/test_dense_gpu.py tests/test_embedding_gpu.py tests/test_model_gpu.py -x -q > gpurun_out/r3n_tests.log 2>&1
echo "tests rc=$?" >> gpurun_out/r3n_tests.log
tail -6 gpurun_out/r3n_tests.log
timeout 600 bash tools/bench_ranks_one_gpu.sh 2 > gpurun_out/r3n_ranks.log 2>&1
echo "ranks rc=$?"; tail -10 gpurun_out/r3n_ranks.log
