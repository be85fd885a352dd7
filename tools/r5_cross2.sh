#!/bin/bash
# (on the GPU box) own-GEMM variants: tests, X1 microbench per variant, kernel stats of the default
cd "$(dirname "$0")/.."
mkdir -p gpurun_out
python -m pytest tests/test_dense_gpu.py -m gpu -x -q -k "own_gemm" 2>&1 | tail -3
for CFG in ${CFGS:-"HCTR_CROSS_GEMM=0" "X=1" "HCTR_GEMM_BM=128" "HCTR_GEMM_BM=64" "HCTR_GEMM_STAGES=3"}; do
  echo "==== $CFG"
  env $CFG python - <<'P'
import sys, torch
sys.path.insert(0, ".")
import bench
from hugectr_amd.layers import MultiCrossLayer
dev = torch.device("cuda", 0)
for B in (8192, 65536):
    w, pdim, L = 3456, 512, 3
    layer = MultiCrossLayer(w, L, pdim).to(dev)
    x = torch.randn(B, w, device=dev).half().requires_grad_(True)
    out = layer(x); g = torch.randn_like(out)
    fwd = bench._timed_us(lambda: layer(x), it=10)
    def both():
        o = layer(x); o.backward(g); x.grad = None; layer.zero_grad(set_to_none=True)
    fb = bench._timed_us(both, it=10)
    fl = L * 4 * B * w * pdim
    print(f"B={B} fwd_us {fwd:.1f} frac {fl/fwd/1e6/2500e0/1e0:.3f} fwd+bwd_us {fb:.1f} frac {3*fl/fb/1e6/2500:.3f}")
    del layer, x, out, g
P
done
BATCHES="8192 65536" bash tools/r5_cross_prof.sh "X=1"
