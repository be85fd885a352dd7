"""print the kernel timeline of the last full bench step in a rocprofv3 *_kernel_trace.csv
(argv[2]: anchor kernel, argv[3]: the step counted from the end, 1 = last)"""
import csv, sys
rows = list(csv.DictReader(open(sys.argv[1])))
rows.sort(key=lambda r: int(r["Start_Timestamp"]))
anchor = sys.argv[2] if len(sys.argv) > 2 else "ht_probe_insert"
idx = [i for i, r in enumerate(rows) if anchor in r["Kernel_Name"]]
back = int(sys.argv[3]) if len(sys.argv) > 3 else 1  # which step, counted from the end
a, b = idx[-1 - back], idx[-back]
t0 = int(rows[a]["Start_Timestamp"])
prev_end, tot, gaps = t0, 0, 0
for r in rows[a:b]:
    s, e = int(r["Start_Timestamp"]), int(r["End_Timestamp"])
    print(f"{(s-t0)/1e3:9.1f} gap {(s-prev_end)/1e3:7.1f} dur {(e-s)/1e3:8.1f}  {r['Kernel_Name'][:100]}")
    gaps += max(0, s - prev_end)
    prev_end = e
    tot += e - s
print("busy", tot / 1e3, "gaps", gaps / 1e3, "span", (int(rows[b]["Start_Timestamp"]) - t0) / 1e3)
