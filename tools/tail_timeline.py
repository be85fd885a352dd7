"""the last N kernels (argv[2], default 150) of a rocprofv3 *_kernel_trace.csv with the gaps between them"""
import csv, sys
rows = list(csv.DictReader(open(sys.argv[1])))
rows.sort(key=lambda r: int(r["Start_Timestamp"]))
n = int(sys.argv[2]) if len(sys.argv) > 2 else 150
rows = rows[-n:]
t0 = int(rows[0]["Start_Timestamp"])
prev = t0
for r in rows:
    s, e = int(r["Start_Timestamp"]), int(r["End_Timestamp"])
    print(f"{(s-t0)/1e3:9.1f} gap {(s-prev)/1e3:7.1f} dur {(e-s)/1e3:8.1f}  {r['Kernel_Name'][:110]}")
    prev = max(prev, e)
