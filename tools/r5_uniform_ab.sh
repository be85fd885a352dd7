#!/bin/bash
# (on the GPU box) the uniform-key leg under a list of environment settings: bash tools/r5_uniform_ab.sh TAG "ENV=.." ...
TAG=$1; shift
cd "$(dirname "$0")/.."
mkdir -p gpurun_out
OUT=gpurun_out/r5_uniform_ab_$TAG.txt; : > $OUT
i=0
for CFG in "$@"; do
  i=$((i+1))
  echo "==== $CFG" >> $OUT
  env $CFG python bench.py --extra uniform --steps 2 --warmup 2 --extra-steps 10 --no-cpu-baseline --extra-file gpurun_out/r5_uniform_ab_${TAG}_$i.json > /dev/null 2>&1
  python -c "
import json,sys; d=json.load(open('gpurun_out/r5_uniform_ab_${TAG}_$i.json')); u=d['extra']['uniform_big_tables']; print(round(u['ms_per_step'],3), {k:round(v,1) for k,v in u['stage_us_per_step'].items()}, 'gather frac', round(u['roofline']['frac'],3), 'upd', u['roofline_update'].get('us'), u['roofline_update'].get('us_grouping_ahead'))" >> $OUT 2>&1
done
cat $OUT
