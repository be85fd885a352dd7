#!/bin/bash
# (on the GPU box) the fused gather under library variants, main leg + uniform leg, interleaved on ONE box
cd "$(dirname "$0")/.."
mkdir -p gpurun_out
OUT=gpurun_out/r5_gather_ab_$1.txt; : > $OUT; shift
for R in 1 2; do
for V in "$@"; do
  VV=$V; [ "$V" = product ] && VV=""
  echo "==== library: $V (round $R)" >> $OUT
  HCTR_LIB_VARIANT=$VV python bench.py --extra uniform --steps 12 --warmup 6 --extra-steps 10 --no-cpu-baseline --extra-file gpurun_out/r5_gather_ab_tmp.json > /dev/null 2>&1
  python -c "
import json; d=json.load(open('gpurun_out/r5_gather_ab_tmp.json')); u=d['extra']['uniform_big_tables']
print('main: ms', round(d['ms_per_step'],4), 'gather_us', round(d['roofline']['avg_launch_us'],1), 'frac', round(d['roofline']['frac'],3), '| uniform: ms', round(u['ms_per_step'],3), 'gather_us', round(u['roofline']['avg_launch_us'],1), 'frac', round(u['roofline']['frac'],3))" >> $OUT 2>&1
done
done
cat $OUT
