#!/bin/bash
# (on the GPU box) MultiCross v2 on the own GEMM vs the library's: tests, then the bench's cross leg
cd "$(dirname "$0")/.."
mkdir -p gpurun_out
python -m pytest tests/test_dense_gpu.py -m gpu -x -q -k "own_gemm or cross_v2" 2>&1 | tail -4
for CFG in "HCTR_CROSS_GEMM=1" "HCTR_CROSS_GEMM=0" "HCTR_CROSS_GEMM=1 HCTR_GEMM_BM=128" "HCTR_CROSS_GEMM=1 HCTR_GEMM_BM=64"; do
  echo "==== $CFG"
  env $CFG python - <<'P'
import sys, json, torch
sys.path.insert(0, ".")
import bench
r = bench.cross_leg(torch.device("cuda", 0))["cases"]
for k, v in r.items():
    if k.startswith("v2"):
        print(k, "fwd_us", round(v["forward_us"], 1), "frac", round(v["roofline"]["frac"], 3), "fwd+bwd_us", round(v["forward_backward_us"], 1), "frac", round(v["roofline_forward_backward"]["frac"], 3))
P
done
