"""print the top kernels of a rocprofv3 *_kernel_stats.csv as microseconds per bench step"""
import csv, sys
f, steps = sys.argv[1], float(sys.argv[2])
top = int(sys.argv[3]) if len(sys.argv) > 3 else 32
rows = list(csv.DictReader(open(f)))
tot = 0.0
for r in rows:
    tot += float(r["TotalDurationNs"])
for r in rows[:top]:
    print(f"{r['Name'][:100]:100s} {r['Calls']:>6s} {float(r['TotalDurationNs'])/1e3/steps:9.1f} us/step  avg {float(r['AverageNs'])/1e3:8.1f}")
print("total us/step", tot / 1e3 / steps)
