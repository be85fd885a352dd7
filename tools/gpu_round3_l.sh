#!/bin/bash
cd /root/repo
mkdir -p gpurun_out
timeout 900 python -m pytest tests/test_sort_gpu.py tests/test_embedding_gpu.py tests/test_ebc_gpu.py tests/test_unique_exchange_gpu.py tests/test_cache_gpu.py -x -q > gpurun_out/r3l_tests.log 2>&1
echo "tests rc=$?" >> gpurun_out/r3l_tests.log
tail -4 gpurun_out/r3l_tests.log
timeout 600 python bench.py --extra none --no-cpu-baseline > gpurun_out/r3l_bench.json 2> gpurun_out/r3l_bench.err
python - <<'PY'
import json
j=json.load(open('gpurun_out/r3l_bench.json'))
print(j['ms_per_step'], j['value'], j['stage_us_per_step'], j['roofline_update']['frac'])
PY
