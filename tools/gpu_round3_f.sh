#!/bin/bash
cd /root/repo
mkdir -p gpurun_out
timeout 1500 python -m pytest tests/test_hash_gpu.py tests/test_embedding_gpu.py tests/test_det_gpu.py tests/test_golden_gpu.py tests/test_ebc_gpu.py tests/test_ebc_dynamic_gpu.py tests/test_sok_gpu.py tests/test_cache_gpu.py tests/test_unique_exchange_gpu.py -x -q > gpurun_out/r3f_tests.log 2>&1
echo "tests rc=$?" >> gpurun_out/r3f_tests.log
tail -4 gpurun_out/r3f_tests.log
cd /tmp && export TMPDIR=/tmp
W="--extra none --no-cpu-baseline --steps 30 --warmup 8"
rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/ks -o k -- python /root/repo/bench.py $W > /tmp/ks.out 2>&1
cp $(find /tmp/ks -name "*kernel_stats.csv" | head -1) /root/repo/gpurun_out/r3f_kernel_stats.csv
python /root/repo/tools/timeline.py $(find /tmp/ks -name "*kernel_trace.csv" | head -1) > /root/repo/gpurun_out/r3f_step_timeline.txt
grep "^{" /tmp/ks.out | tail -1 > /root/repo/gpurun_out/r3f_bench_under_rocprof.json
cd /root/repo
python - <<'PY'
import json, csv, re
j=json.load(open('gpurun_out/r3f_bench_under_rocprof.json'))
print(j['ms_per_step'], j['value'], j['stage_us_per_step'], j['stage_us_per_step_no_new_keys'])
for r in csv.DictReader(open('gpurun_out/r3f_kernel_stats.csv')):
    n=r['Name']
    if 'ht_' in n or 'seg_' in n or 'rs_' in n or 'expand' in n or 'interaction' in n:
        short=re.split(r'[<(]', n.split('(anonymous namespace)::')[-1])[0]
        print(f"{short:36s} calls={r['Calls']:>5s} avg={float(r['AverageNs'])/1e3:8.1f} min={float(r['MinNs'])/1e3:8.1f} max={float(r['MaxNs'])/1e3:8.1f}")
PY
tail -60 gpurun_out/r3f_step_timeline.txt | cut -c1-150
