#!/bin/bash
cd /root/repo
mkdir -p gpurun_out
( time bash tools/measure_round.sh v2 all ) > gpurun_out/r3m_measure.log 2>&1
python - <<'PY'
import json
j=json.load(open('gpurun_out/r3_bench_n1_v2.json'))
print(j['ms_per_step'], j['value'], j['stage_us_per_step'], j['stage_us_per_step_no_new_keys'])
print({k:(v.get('ms_per_step') or v.get('forward_us') or v.get('lookup_us') or v.get('error')) for k,v in j['extra'].items()})
print(j['roofline'])
print(j['extra'].get('uniform_big_tables',{}).get('roofline'))
PY
grep -E "^real|^user" gpurun_out/r3m_measure.log | head -3
timeout 600 bash tools/bench_ranks_one_gpu.sh 2 > gpurun_out/r3m_ranks.log 2>&1
echo "ranks rc=$?"; tail -12 gpurun_out/r3m_ranks.log
timeout 1500 python -m pytest tests -m gpu -x -q > gpurun_out/r3m_tests.log 2>&1
echo "tests rc=$?" >> gpurun_out/r3m_tests.log
tail -4 gpurun_out/r3m_tests.log
