#!/bin/bash
cd /root/repo
mkdir -p gpurun_out
timeout 1200 python -m pytest tests/test_hash_gpu.py tests/test_dense_gpu.py tests/test_embedding_gpu.py tests/test_model_gpu.py tests/test_loss_curve_gpu.py tests/test_det_gpu.py -x -q > gpurun_out/r3c_tests.log 2>&1
echo "tests rc=$?" >> gpurun_out/r3c_tests.log
tail -8 gpurun_out/r3c_tests.log
timeout 600 python bench.py --extra none --no-cpu-baseline > gpurun_out/r3c_bench.json 2> gpurun_out/r3c_bench.err
echo "bench rc=$?"; tail -3 gpurun_out/r3c_bench.err
python - <<'PY'
import json
j=json.load(open('gpurun_out/r3c_bench.json'))
print(j['ms_per_step'], j['value'], j['stage_us_per_step'], j['roofline']['frac'], j['roofline']['frac_compulsory'], j['roofline']['kernel'][:40], j['config']['new_keys_per_step'], j['config']['dense_gemm_selection'])
PY
HCTR_FUSED_GATHER=0 timeout 600 python bench.py --extra none --no-cpu-baseline > gpurun_out/r3c_bench_unfused.json 2> gpurun_out/r3c_bench_unfused.err
python - <<'PY'
import json
j=json.load(open('gpurun_out/r3c_bench_unfused.json'))
print(j['ms_per_step'], j['value'], j['stage_us_per_step'], j['roofline']['frac'], j['roofline']['frac_compulsory'], j['roofline']['kernel'][:40])
PY
