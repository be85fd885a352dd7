#!/bin/bash
# (on the GPU box) kernel timeline of the update part of the main bench leg under a list of
# environment settings:  bash tools/r5_quick.sh TAG "ENV1=.. ENV2=.." "ENV=.." ...
TAG=$1; shift
cd "$(dirname "$0")/.."
mkdir -p gpurun_out
OUT=gpurun_out/r5_quick_$TAG.txt; : > $OUT
W="--extra none --no-cpu-baseline --steps 24 --warmup 6"
i=0
for CFG in "$@"; do
  i=$((i+1))
  echo "==== [$i] $CFG" >> $OUT
  rm -rf /tmp/ks$i
  ( cd /tmp && export TMPDIR=/tmp && env $CFG timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/ks$i -o k -- python /root/repo/bench.py $W > /tmp/ks$i.out 2>&1 )
  grep "^{" /tmp/ks$i.out | tail -1 | python -c "import json,sys; j=json.loads(sys.stdin.read()); print('ms_per_step', round(j['ms_per_step'],4), 'update', round(j['roofline_update']['us'],1), round(j['roofline_update']['frac'],3), 'index', round(j['roofline_index']['us'],1), 'loss', j['config']['final_loss'])" >> $OUT 2>&1
  python tools/timeline.py $(find /tmp/ks$i -name "*kernel_trace.csv" | head -1) > gpurun_out/r5_quick_timeline_${TAG}_$i.txt 2>&1
  grep "expand_pairs\|rs_\|cold_\|hot_\|seg_\|ht_\|busy" gpurun_out/r5_quick_timeline_${TAG}_$i.txt | cut -c1-120 >> $OUT
done
cat $OUT
