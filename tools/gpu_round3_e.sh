#!/bin/bash
cd /root/repo
mkdir -p gpurun_out
timeout 900 python -m pytest tests/test_hash_gpu.py tests/test_embedding_gpu.py tests/test_det_gpu.py tests/test_golden_gpu.py -x -q > gpurun_out/r3e_tests.log 2>&1
echo "tests rc=$?" >> gpurun_out/r3e_tests.log
tail -4 gpurun_out/r3e_tests.log
timeout 600 python bench.py --extra none --no-cpu-baseline > gpurun_out/r3e_bench.json 2> gpurun_out/r3e_bench.err
echo "bench rc=$?"; tail -3 gpurun_out/r3e_bench.err
python - <<'PY'
import json
j=json.load(open('gpurun_out/r3e_bench.json'))
print(j['ms_per_step'], j['value'], j['stage_us_per_step'], j['stage_us_per_step_no_new_keys'], j['roofline']['frac'], j['roofline']['frac_compulsory'], j['config']['new_keys_per_step'])
PY
