#!/bin/bash
cd /root/repo
mkdir -p gpurun_out
cd /tmp && export TMPDIR=/tmp
W="--extra none --no-cpu-baseline --steps 30 --warmup 8"
rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/ks -o k -- python /root/repo/bench.py $W > /tmp/ks.out 2>&1
cp $(find /tmp/ks -name "*kernel_stats.csv" | head -1) /root/repo/gpurun_out/r3d_kernel_stats.csv
python /root/repo/tools/timeline.py $(find /tmp/ks -name "*kernel_trace.csv" | head -1) > /root/repo/gpurun_out/r3d_step_timeline.txt
grep "^{" /tmp/ks.out | tail -1 > /root/repo/gpurun_out/r3d_bench_under_rocprof.json
head -40 /root/repo/gpurun_out/r3d_kernel_stats.csv | cut -c1-200
cd /root/repo
timeout 900 python -m pytest tests/test_model_gpu.py tests/test_loss_curve_gpu.py tests/test_dropin_gpu.py -x -q > gpurun_out/r3d_tests.log 2>&1
echo "tests rc=$?" >> gpurun_out/r3d_tests.log
tail -8 gpurun_out/r3d_tests.log
timeout 600 python bench.py --config c1 --steps 100 --warmup 30 > gpurun_out/r3d_c1.json 2> gpurun_out/r3d_c1.err; tail -2 gpurun_out/r3d_c1.err
timeout 600 python bench.py --config c2 --steps 50 --warmup 20 > gpurun_out/r3d_c2.json 2> gpurun_out/r3d_c2.err; tail -2 gpurun_out/r3d_c2.err
HCTR_HIP_GRAPH=0 timeout 600 python bench.py --config c1 --steps 100 --warmup 30 > gpurun_out/r3d_c1_eager.json 2>/dev/null
HCTR_HIP_GRAPH=0 timeout 600 python bench.py --config c2 --steps 50 --warmup 20 > gpurun_out/r3d_c2_eager.json 2>/dev/null
python - <<'PY'
import json
for n in ("c1","c1_eager","c2","c2_eager"):
    try:
        j=json.load(open(f'gpurun_out/r3d_{n}.json'))
        print(n, j['ms_per_step'], j['value'], j['config'].get('hip_graph'), j['config']['final_loss'], j['stage_us_per_step'])
    except Exception as e: print(n, 'ERR', e)
PY
