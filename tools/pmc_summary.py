"""Turn rocprofv3 --pmc passes of `bench.py` into profiles/pmc_<kernel>.json.

Usage (on the GPU box, separate passes as MI355X_MICROARCH.md prescribes -- FETCH_SIZE and
WRITE_SIZE do not fit one pass):
  rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d out/fetch -o p -- python bench.py ...
  rocprofv3 --pmc WRITE_SIZE --kernel-trace --output-format csv -d out/write -o p -- python bench.py ...
  python tools/pmc_summary.py out/fetch/p_counter_collection.csv out/write/p_counter_collection.csv \
         --kernel pool_vec4_kernel --out profiles/pmc_gather_pool.json
Units / corrections (guide, section HBM): FETCH_SIZE and WRITE_SIZE are in KiB-like units of 1024 B
(`hbm_bytes = (FETCH_SIZE + WRITE_SIZE) * 1024`); on gfx950 FETCH_SIZE reports exactly half of the
bytes of a wide (16 B/lane) coalesced read, so the read side is doubled.  WRITE_SIZE is calibrated
here against the gather's known output size (it matches 1:1).
"""
import argparse
import csv
import json
import statistics


def collect(path, kernel, counter):
    vals = []
    for r in csv.DictReader(open(path)):
        if kernel in r["Kernel_Name"] and r["Counter_Name"] == counter:
            vals.append(float(r["Counter_Value"]))
    return vals


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("fetch_csv")
    ap.add_argument("write_csv")
    ap.add_argument("--kernel", default="pool_vec4_kernel")
    ap.add_argument("--out", required=True)
    ap.add_argument("--skip", type=int, default=8, help="launches to skip (warm-up / cold inserts)")
    a = ap.parse_args()
    f = collect(a.fetch_csv, a.kernel, "FETCH_SIZE")[a.skip:]
    w = collect(a.write_csv, a.kernel, "WRITE_SIZE")[a.skip:]
    fetch_raw = statistics.mean(f) * 1024
    write_b = statistics.mean(w) * 1024
    res = {"kernel": a.kernel, "launches_averaged": [len(f), len(w)],
           "fetch_bytes_raw_counter": fetch_raw, "fetch_bytes_corrected_x2": 2 * fetch_raw,
           "write_bytes": write_b, "hbm_bytes_per_launch": 2 * fetch_raw + write_b,
           "method": "rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE in separate passes, x1024 B, gfx950 "
                     "FETCH_SIZE x2 correction for 16-B/lane coalesced reads (MI355X_MICROARCH.md HBM)"}
    json.dump(res, open(a.out, "w"), indent=1)
    print(json.dumps(res))


if __name__ == "__main__":
    main()
