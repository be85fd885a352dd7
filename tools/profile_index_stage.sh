#!/bin/bash
# per-launch medians of the index-stage kernels (steps that insert / steps that do not) from a
# rocprofv3 kernel trace of the main bench leg
cd /tmp && export TMPDIR=/tmp
W="--extra none --no-cpu-baseline --steps 30 --warmup 8"
rm -rf /tmp/ks
rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/ks -o k -- python /root/repo/bench.py $W > /tmp/ks.out 2>&1
cd /root/repo
python - <<'PY'
import csv, glob, statistics, re
f = glob.glob('/tmp/ks/**/*kernel_trace.csv', recursive=True)[0]
rows = list(csv.DictReader(open(f)))
rows.sort(key=lambda r: int(r["Start_Timestamp"]))
per = {}
for r in rows:
    n = r["Kernel_Name"]
    if "hctr" not in n: continue
    short = re.split(r'[<(]', n.split('(anonymous namespace)::', 1)[-1])[0]
    per.setdefault(short, []).append((int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3)
for k in ("ht_probe_insert_kernel", "ht_finish_kernel", "interaction_fwd16_gather_kernel", "seg_reduce_kernel", "expand_pairs_kernel"):
    v = per.get(k, [])
    if len(v) >= 50:
        print(k, "insert steps median", round(statistics.median(v[10:38]), 1), "steady median", round(statistics.median(v[-8:]), 1), "n", len(v))
    else:
        print(k, len(v), [round(x,1) for x in v[:60]])
PY
