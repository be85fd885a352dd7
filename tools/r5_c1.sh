#!/bin/bash
# (on the GPU box) BASELINE configs[0] (DCN README, batch 1024) under environment settings + the
# kernel timeline of its last steps
cd "$(dirname "$0")/.."
mkdir -p gpurun_out
for CFG in "$@"; do
  echo "==== $CFG"
  env $CFG python bench.py --config c1 --steps 200 --warmup 50 --extra-file gpurun_out/r5_c1_tmp.json 2>/dev/null | python -c "import json,sys; j=json.loads(sys.stdin.read()); print('ms_per_step', j['ms_per_step'], 'value', j['value'])"
done
rm -rf /tmp/kc1
( cd /tmp && export TMPDIR=/tmp && env $1 rocprofv3 --kernel-trace --output-format csv -d /tmp/kc1 -o k -- python /root/repo/bench.py --config c1 --steps 30 --warmup 30 --extra-file /tmp/c1.json > /tmp/kc1.out 2>&1 )
python tools/tail_timeline.py $(find /tmp/kc1 -name "*kernel_trace.csv" | head -1) 130 | cut -c1-150 > gpurun_out/r5_c1_timeline.txt
tail -75 gpurun_out/r5_c1_timeline.txt
