#!/bin/bash
# (on the GPU box) kernel timeline of the last step of the uniform-key leg
TAG=$1; shift
cd "$(dirname "$0")/.."
mkdir -p gpurun_out
rm -rf /tmp/ku
( cd /tmp && export TMPDIR=/tmp && env "$@" timeout 400 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/ku -o k -- python /root/repo/bench.py --extra uniform --no-cpu-baseline --steps 2 --warmup 2 --extra-steps 6 > /tmp/ku.out 2>&1 )
python tools/timeline.py $(find /tmp/ku -name "*kernel_trace.csv" | head -1) ht_probe_insert ${BACK:-15} > gpurun_out/r4_uniform_timeline_$TAG.txt 2>&1
grep "ht_\|hot_\|rs_\|seg_\|expand\|interaction_fwd16_gather\|busy" gpurun_out/r4_uniform_timeline_$TAG.txt | cut -c1-110
