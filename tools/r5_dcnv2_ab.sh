#!/bin/bash
# (on the GPU box) the MLPerf DCNv2 model leg under environment settings, interleaved on one box
cd "$(dirname "$0")/.."
for R in 1 2; do
for CFG in "$@"; do
  echo "==== $CFG (round $R)"
  env $CFG python bench.py --extra dcnv2 --steps 2 --warmup 2 --extra-steps 10 --no-cpu-baseline --extra-file gpurun_out/r5_dcnv2_tmp.json > /dev/null 2>&1
  python -c "
import json; d=json.load(open('gpurun_out/r5_dcnv2_tmp.json')); u=d['extra']['dcnv2_model']; print('ms_per_step', round(u['ms_per_step'],3), 'loss', u.get('final_loss'))"
done
done
