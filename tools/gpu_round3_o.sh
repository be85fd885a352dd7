#!/bin/bash
cd /root/repo
mkdir -p gpurun_out
timeout 1800 python -m pytest tests -m gpu -x -q > gpurun_out/r3o_tests.log 2>&1
echo "tests rc=$?" >> gpurun_out/r3o_tests.log
tail -4 gpurun_out/r3o_tests.log
python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | tail -2
( time python bench.py > gpurun_out/r3o_bench.json 2> gpurun_out/r3o_bench.err ) 2>&1 | grep real
python - <<'PY'
import json
j=json.load(open('gpurun_out/r3o_bench.json'))
print(j['ms_per_step'], j['value'], j['stage_us_per_step'], j['roofline']['traffic'], j['roofline']['traffic_source'])
print({k:(v.get('ms_per_step') or v.get('forward_us') or v.get('lookup_us') or v.get('error')) for k,v in j['extra'].items()})
print(j['extra']['uniform_big_tables']['roofline']['traffic_source'], j['extra']['ebc_multi_hot']['roofline']['traffic_source'])
PY
