#!/bin/bash
# (on the GPU box) the uniform-key leg under a list of environment settings: bash tools/r4_uniform_ab.sh TAG "ENV=.." ...
TAG=$1; shift
cd "$(dirname "$0")/.."
OUT=gpurun_out/r4_uniform_ab_$TAG.txt; : > $OUT
for CFG in "$@"; do
  echo "==== $CFG" >> $OUT
  env $CFG python bench.py --extra uniform --steps 2 --warmup 2 --extra-steps 10 --no-cpu-baseline 2>/dev/null | tail -1 | python -c "
import json,sys; d=json.loads(sys.stdin.read()); u=d['extra']['uniform_big_tables']; print(round(u['ms_per_step'],3), {k:round(v,1) for k,v in u['stage_us_per_step'].items()}, round(u['roofline']['frac'],3))" >> $OUT 2>&1
done
cat $OUT
