#!/bin/bash
# (on the GPU box) the tiered table: tests, then the bench's tiered leg whole and in pieces
cd "$(dirname "$0")/.."
python -m pytest tests/test_cache_gpu.py -m gpu -x -q 2>&1 | tail -2
for CFG in "HCTR_TIER_PIECES=1" "HCTR_TIER_PIECES=4" "HCTR_TIER_PIECES=4 HCTR_TIER_FILL_GRID=32" "HCTR_TIER_PIECES=4 HCTR_TIER_FILL_GRID=16" "HCTR_TIER_PIECES=3 HCTR_TIER_FILL_GRID=32" "HCTR_TIER_PIECES=1"; do
  echo "==== $CFG"
  env $CFG python - <<'P'
import sys, torch
sys.path.insert(0, ".")
import bench
r = bench.tiered_leg(10, 3, torch.device("cuda", 0))
print({k: (round(v, 1) if isinstance(v, float) else v) for k, v in r.items() if k in ("lookup_us", "update_us", "lookup_update_us", "miss_rate")}, "link frac", round(r["roofline"]["frac"], 3))
P
done
