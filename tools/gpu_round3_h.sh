#!/bin/bash
cd /root/repo
mkdir -p gpurun_out
bash tools/measure_round.sh v1 all > gpurun_out/r3h_measure.log 2>&1
ls -la gpurun_out/ | tail -20
python - <<'PY'
import json
j=json.load(open('gpurun_out/r3_bench_n1_v1.json'))
print(j['ms_per_step'], j['value'], j['stage_us_per_step'], j['stage_us_per_step_no_new_keys'])
print({k:(v.get('ms_per_step') or v.get('forward_us') or v.get('lookup_us') or v.get('error')) for k,v in j['extra'].items()})
print(j['extra'].get('uniform_big_tables',{}).get('roofline'))
print(j.get('cpu_baseline',{}).get('value'))
PY
timeout 1500 python -m pytest tests -m gpu -x -q > gpurun_out/r3h_tests.log 2>&1
echo "tests rc=$?" >> gpurun_out/r3h_tests.log
tail -5 gpurun_out/r3h_tests.log
