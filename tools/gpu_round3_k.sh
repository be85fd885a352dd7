#!/bin/bash
cd /tmp && export TMPDIR=/tmp
rm -rf /tmp/ks
rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/ks -o k -- python /root/repo/bench.py --extra next --steps 2 --warmup 2 --extra-steps 6 --no-cpu-baseline > /tmp/ks.out 2>&1
cd /root/repo
cp $(find /tmp/ks -name "*kernel_stats.csv" | head -1) gpurun_out/r3k_ebc_dynamic_kernel_stats.csv
python - <<'PY'
import csv, re, json
rows=list(csv.DictReader(open('gpurun_out/r3k_ebc_dynamic_kernel_stats.csv')))
for r in rows[:45]:
    n=r['Name']
    m=re.match(r"\s*(?:void\s+)?(?:hctr::)?(?:\(anonymous namespace\)::)?([A-Za-z_][\w:]*)", n)
    print(f"{(m.group(1) if m else n)[:44]:44s} calls={r['Calls']:>5s} total_ms={float(r['TotalDurationNs'])/1e6:8.2f} avg_us={float(r['AverageNs'])/1e3:9.1f}")
import glob
j=json.loads([l for l in open('/tmp/ks.out') if l.startswith('{')][-1])
print({k:(v.get('forward_us'), v.get('backward_update_us'), v.get('lookup_us'), v.get('update_us')) for k,v in j['extra'].items()})
PY
