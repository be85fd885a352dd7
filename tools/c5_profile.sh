#!/bin/bash
# Where the configs[4] leg (Wide & Deep + MMoE on dynamic tables through Model.train()) spends its
# step: host profile (cProfile, cumulative) and kernel totals (rocprofv3 --stats) of bench.py --extra c5
mkdir -p gpurun_out
python -m cProfile -o /tmp/c5.prof bench.py --extra c5 --steps 2 --warmup 2 --no-cpu-baseline > /tmp/c5.out 2>&1
python - <<'PY' > gpurun_out/r6_c5_host_profile.txt 2>&1
import pstats
p = pstats.Stats('/tmp/c5.prof')
p.sort_stats('cumulative').print_stats(70)
PY
tail -3 /tmp/c5.out >> gpurun_out/r6_c5_host_profile.txt
cd /tmp && export TMPDIR=/tmp
rm -rf /tmp/c5ks
rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/c5ks -o k -- python /root/repo/bench.py --extra c5 --steps 2 --warmup 2 --no-cpu-baseline > /tmp/c5ks.out 2>&1
cd /root/repo
python - <<'PY' > gpurun_out/r6_c5_kernel_trace.txt 2>&1
import csv, glob, collections
f = glob.glob('/tmp/c5ks/**/*kernel_trace.csv', recursive=True)[0]
rows = list(csv.DictReader(open(f)))
rows.sort(key=lambda r: int(r["Start_Timestamp"]))
# the c5 leg's steps are the tail of the trace: take the last 30 % of the kernels by time window
t0, t1 = int(rows[0]["Start_Timestamp"]), int(rows[-1]["End_Timestamp"])
# find the last big gap-free window: use the last 24 train steps ~ approximate by the last N kernels
N = len(rows)
tail = rows[int(N * 0.6):]
span = (int(tail[-1]["End_Timestamp"]) - int(tail[0]["Start_Timestamp"])) / 1e6
busy = sum(int(r["End_Timestamp"]) - int(r["Start_Timestamp"]) for r in tail) / 1e6
print("kernels in window", len(tail), "window ms", round(span, 2), "kernel-busy ms", round(busy, 2))
agg = collections.defaultdict(lambda: [0, 0])
for r in tail:
    n = r["Kernel_Name"][:100]
    agg[n][0] += 1
    agg[n][1] += int(r["End_Timestamp"]) - int(r["Start_Timestamp"])
for n, (c, t) in sorted(agg.items(), key=lambda kv: -kv[1][1])[:40]:
    print(f"{t/1e3:10.1f} us {c:6d}  {n}")
PY
