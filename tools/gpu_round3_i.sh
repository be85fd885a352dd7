#!/bin/bash
cd /root/repo
mkdir -p gpurun_out
timeout 1500 python -m pytest tests/test_model_gpu.py tests/test_dropin_gpu.py tests/test_ebc_gpu.py -x -q > gpurun_out/r3i_tests.log 2>&1
echo "tests rc=$?" >> gpurun_out/r3i_tests.log
tail -6 gpurun_out/r3i_tests.log
timeout 900 python bench.py --extra uniform --steps 6 --warmup 3 --no-cpu-baseline > gpurun_out/r3i_uniform.json 2> gpurun_out/r3i_uniform.err
tail -2 gpurun_out/r3i_uniform.err
python - <<'PY'
import json
j=json.load(open('gpurun_out/r3i_uniform.json'))
u=j['extra']['uniform_big_tables']
print(json.dumps(u)[:1500])
PY
timeout 600 python bench.py --config c1 --steps 100 --warmup 30 > gpurun_out/r3i_c1.json 2> gpurun_out/r3i_c1.err; tail -1 gpurun_out/r3i_c1.err
python -c "
import json; j=json.load(open('gpurun_out/r3i_c1.json')); print(j['ms_per_step'], j['config'].get('hip_graph'))"
