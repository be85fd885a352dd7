#!/bin/bash
# (on the GPU box) index stage under a list of environment settings: main leg + the uniform leg
TAG=$1; shift
cd "$(dirname "$0")/.."
mkdir -p gpurun_out
OUT=gpurun_out/r4_index_$TAG.txt; : > $OUT
for CFG in "$@"; do
  echo "==== $CFG" >> $OUT
  env $CFG timeout 400 python bench.py --extra uniform --no-cpu-baseline --steps 30 --warmup 8 --extra-steps 10 2>/dev/null | grep "^{" | tail -1 | python -c "
import json,sys; j=json.loads(sys.stdin.read())
u=j['extra']['uniform_big_tables']
print('main: ms', round(j['ms_per_step'],4), 'index', round(j['roofline_index']['us'],1), 'steady', round(j['stage_us_per_step_no_new_keys']['hash_index'],1), 'new/step', j['config']['new_keys_per_step'])
print('uniform: ms', round(u['ms_per_step'],4), 'index', round(u['roofline_index']['us'],1), 'gather frac', round(u['roofline']['frac'],3), 'gather us', round(u['roofline']['avg_launch_us'],1), 'update', round(u['roofline_update']['us'],1), 'new/step', u['new_keys_per_step'])" >> $OUT 2>&1
done
cat $OUT
