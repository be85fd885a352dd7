#!/bin/bash
# A/B of the sparse update inside the bench step: kernel stats of the update's kernels for the
# product library and for variant libraries (HCTR_LIB_VARIANT).  bash tools/ab_update.sh "" noatomic ...
cd /tmp && export TMPDIR=/tmp
for v in "$@"; do
  echo "== variant '${v}' env: $ENVX"
  rm -rf /tmp/ab_$v
  env HCTR_LIB_VARIANT=$v $ENVX rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/ab_$v -o k -- python /root/repo/bench.py --extra none --no-cpu-baseline --steps 30 --warmup 8 > /tmp/ab_$v.out 2>&1
  grep '^{' /tmp/ab_$v.out | tail -1 | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('ms_per_step', d['ms_per_step'], 'update', json.dumps(d.get('roofline_update')))"
  python /root/repo/tools/kstats.py $(find /tmp/ab_$v -name "*kernel_stats.csv" | head -1) 38 200 | grep -i "hot_\|cold_\|total"
done
