#!/bin/bash
# (on the GPU box) kernel traces of the legs VERDICT items 6 / 8 name: bash tools/leg_profile.sh TAG LEG...
TAG=$1; shift
cd "$(dirname "$0")/.."
mkdir -p gpurun_out
for LEG in "$@"; do
  OUT=gpurun_out/r4_next_${LEG}_$TAG.txt; : > $OUT
  rm -rf /tmp/np_$LEG
  ( cd /tmp && export TMPDIR=/tmp && timeout 400 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/np_$LEG -o k -- python /root/repo/tools/leg_profile.py $LEG > /tmp/np_$LEG.out 2>&1 )
  grep "^{" /tmp/np_$LEG.out | tail -1 >> $OUT
  tail -3 /tmp/np_$LEG.out | cut -c1-300 >> $OUT
  echo "---- kernel stats" >> $OUT
  head -40 $(find /tmp/np_$LEG -name "*kernel_stats.csv" | head -1) | cut -c1-200 >> $OUT
  echo "---- last kernels" >> $OUT
  python tools/tail_timeline.py $(find /tmp/np_$LEG -name "*kernel_trace.csv" | head -1) ${NTAIL:-160} >> $OUT 2>&1
done
