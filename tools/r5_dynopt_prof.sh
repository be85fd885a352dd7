#!/bin/bash
# kernel totals of the dynamic tables' AdaGrad leg on the flat row store and on the unique-key flow
cd /tmp && export TMPDIR=/tmp
for F in 1 0; do
  rm -rf /tmp/dp$F
  HCTR_DYNAMIC_FLAT=$F timeout 80 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/dp$F -o k -- python /root/repo/tools/dyn_leg.py adagrad 6 > /tmp/dp$F.out 2>&1
  echo "==== HCTR_DYNAMIC_FLAT=$F  $(grep '^{' /tmp/dp$F.out | tail -1)"
  f=$(find /tmp/dp$F -name "*kernel_stats.csv" | head -1)
  [ -n "$f" ] && python -c "
import csv, sys
for i, r in enumerate(csv.DictReader(open('$f'))):
    if i >= 22: break
    print('%6s calls %9.1f us avg %5s %%  %s' % (r['Calls'], float(r['AverageNs']) / 1e3, r['Percentage'], r['Name'][:110]))"
done
