#!/bin/bash
cd /root/repo
mkdir -p gpurun_out
for w in 8 12 16 24; do
  HCTR_GATHER_WAVES=$w timeout 600 python bench.py --extra uniform --steps 12 --warmup 8 --extra-steps 10 --no-cpu-baseline > gpurun_out/r3j_w$w.json 2> gpurun_out/r3j_w$w.err
  python - <<PY
import json
j=json.load(open('gpurun_out/r3j_w$w.json'))
u=j['extra']['uniform_big_tables']
print($w, 'main', round(j['ms_per_step'],3), round(j['roofline']['avg_launch_us'],1), round(j['roofline']['frac'],3), '| uniform', round(u['roofline']['avg_launch_us'],1), round(u['roofline']['frac'],3), u['stage_us_per_step'])
PY
done
