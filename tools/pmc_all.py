"""rocprofv3 --pmc FETCH_SIZE / --pmc WRITE_SIZE passes of bench.py -> one JSON with the HBM bytes
per launch of every hot kernel (same method and corrections as tools/pmc_summary.py).

  python tools/pmc_all.py fetch_counter_collection.csv write_counter_collection.csv out.json \
         --precision fp16 --workload "python bench.py ..." [--skip 8]"""
import argparse
import csv
import json
import statistics

KERNELS = ["ht_probe_insert_kernel", "pool_vec4_kernel", "interaction_fwd", "interaction_bwd",
           "rs_scatter_kernel", "expand_pairs_kernel", "seg_reduce_kernel", "seg_apply_kernel",
           "seg_combine_kernel"]


def collect(path, kernel, counter):
    return [float(r["Counter_Value"]) for r in csv.DictReader(open(path))
            if kernel in r["Kernel_Name"] and r["Counter_Name"] == counter]


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("fetch_csv")
    ap.add_argument("write_csv")
    ap.add_argument("out")
    ap.add_argument("--precision", default="fp16")
    ap.add_argument("--workload", default="")
    ap.add_argument("--skip", type=int, default=8, help="steps to skip (warm-up / cold inserts)")
    a = ap.parse_args()
    res = {"precision": a.precision, "workload": a.workload,
           "method": "rocprofv3 --pmc FETCH_SIZE and --pmc WRITE_SIZE in SEPARATE passes (with "
                     "--kernel-trace only), counter x 1024 B, gfx950 FETCH_SIZE x2 correction for "
                     "16-B/lane coalesced reads (MI355X_MICROARCH.md, section HBM); averages over the "
                     "timed launches (warm-up launches skipped)",
           "kernels": {}}
    for k in KERNELS:
        f, w = collect(a.fetch_csv, k, "FETCH_SIZE"), collect(a.write_csv, k, "WRITE_SIZE")
        if not f or not w:
            continue
        per_step = max(1, round(len(f) / max(1, len(collect(a.fetch_csv, "pool_vec4_kernel", "FETCH_SIZE")))))
        f, w = f[a.skip * per_step:], w[a.skip * per_step:]
        if not f or not w:
            continue
        fr, wb = statistics.mean(f) * 1024, statistics.mean(w) * 1024
        res["kernels"][k] = {"kernel": k, "launches_averaged": [len(f), len(w)],
                             "launches_per_step": per_step, "fetch_bytes_raw_counter": fr,
                             "fetch_bytes_corrected_x2": 2 * fr, "write_bytes": wb,
                             "hbm_bytes_per_launch": 2 * fr + wb}
    json.dump(res, open(a.out, "w"), indent=1)
    print(json.dumps({k: round(v["hbm_bytes_per_launch"] / 1e6, 1) for k, v in res["kernels"].items()}))


if __name__ == "__main__":
    main()
