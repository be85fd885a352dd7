cd /root/repo
timeout 300 python -m pytest tests/test_dense_gpu.py -x -q -m gpu -k "cross" 2>&1 | tail -3
timeout 600 python -m pytest tests/test_fullsize_fused_gpu.py tests/test_model_gpu.py -x -q -m gpu 2>&1 | tail -3
python bench.py --extra dense --steps 3 --warmup 2 --no-cpu-baseline 2>gpurun_out/r4_dense.err | python -c "import json,sys; j=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(json.dumps(j['extra'], indent=1))" > gpurun_out/r4_dense.json; tail -3 gpurun_out/r4_dense.err
python bench.py --extra dcnv2 --steps 3 --warmup 2 --no-cpu-baseline 2>gpurun_out/r4_dcnv2.err | python -c "import json,sys; j=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(json.dumps(j['extra'], indent=1))" > gpurun_out/r4_dcnv2.json; tail -3 gpurun_out/r4_dcnv2.err
