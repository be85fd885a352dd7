#!/bin/bash
# (on the GPU box) main bench leg under a list of environment settings, no profiler:
#   bash tools/r5_ab.sh TAG "ENV=.." "ENV=.." ...   -> gpurun_out/r5_ab_TAG.txt
TAG=$1; shift
cd "$(dirname "$0")/.."
mkdir -p gpurun_out
OUT=gpurun_out/r5_ab_$TAG.txt; : > $OUT
W="--extra none --no-cpu-baseline --steps 40 --warmup 10"
for CFG in "$@"; do
  echo "==== $CFG" >> $OUT
  env $CFG timeout 300 python bench.py $W 2>/dev/null | grep "^{" | tail -1 | python -c "import json,sys; j=json.loads(sys.stdin.read()); print('ms_per_step', round(j['ms_per_step'],4), 'update', round(j['roofline_update']['us'],1), round(j['roofline_update']['frac'],3), 'index', round(j['roofline_index']['us'],1), 'gather', round(j['roofline']['avg_launch_us'],1), 'loss', j['config']['final_loss'])" >> $OUT 2>&1
done
cat $OUT
