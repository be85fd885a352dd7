"""rocprofv3 --pmc FETCH_SIZE / --pmc WRITE_SIZE passes of bench.py -> one JSON with the HBM bytes
per launch of every hot kernel of this library (method and corrections of
/opt/skills/guides/MI355X_MICROARCH.md, section HBM: separate passes, counter x 1024 B, gfx950
FETCH_SIZE x 2).

  python tools/pmc_all.py fetch_counter_collection.csv write_counter_collection.csv out.json \
         --precision fp16 --workload "python bench.py ..." [--skip-frac 0.4] [--commit HASH]"""
import argparse
import csv
import json
import re
import statistics
from collections import defaultdict


def short(name):
    """void hctr::(anonymous namespace)::pool_vec4_kernel<32, 4, long long, __half>(...) ->
    pool_vec4_kernel (the FIRST qualified name: parameter types carry the namespace too)"""
    m = re.match(r"\s*(?:void\s+)?(?:hctr::)?(?:\(anonymous namespace\)::)?([A-Za-z_]\w*)", name)
    return m.group(1) if m else name


def collect(path, counter):
    per = defaultdict(list)
    for r in csv.DictReader(open(path)):
        if r["Counter_Name"] != counter:
            continue
        k = r["Kernel_Name"]
        if "hctr::" not in k and "hctr_" not in k:
            continue
        per[short(k)].append(float(r["Counter_Value"]))
    return per


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("fetch_csv")
    ap.add_argument("write_csv")
    ap.add_argument("out")
    ap.add_argument("--precision", default="fp16")
    ap.add_argument("--workload", default="")
    ap.add_argument("--alpha", type=float, default=1.1)
    ap.add_argument("--commit", default="")
    ap.add_argument("--csrc-hash", default="")
    ap.add_argument("--skip-frac", type=float, default=0.4,
                    help="leading share of every kernel's launches left out (warm-up / cold inserts)")
    a = ap.parse_args()
    res = {"precision": a.precision, "workload": a.workload, "alpha": a.alpha, "commit": a.commit,
           "csrc_hash": a.csrc_hash,
           "method": "rocprofv3 --pmc FETCH_SIZE and --pmc WRITE_SIZE in SEPARATE passes (with "
                     "--kernel-trace only), counter x 1024 B, gfx950 FETCH_SIZE x2 correction for "
                     "16-B/lane coalesced reads (MI355X_MICROARCH.md, section HBM); averages over the "
                     f"last {100 - int(a.skip_frac * 100)} % of every kernel's launches",
           "kernels": {}}
    fetch, write = collect(a.fetch_csv, "FETCH_SIZE"), collect(a.write_csv, "WRITE_SIZE")
    for k in sorted(set(fetch) & set(write)):
        f, w = fetch[k], write[k]
        f, w = f[int(len(f) * a.skip_frac):], w[int(len(w) * a.skip_frac):]
        if not f or not w:
            continue
        fr, wb = statistics.mean(f) * 1024, statistics.mean(w) * 1024
        res["kernels"][k] = {"kernel": k, "launches_averaged": [len(f), len(w)],
                             "fetch_bytes_raw_counter": fr, "fetch_bytes_corrected_x2": 2 * fr,
                             "write_bytes": wb, "hbm_bytes_per_launch": 2 * fr + wb}
    json.dump(res, open(a.out, "w"), indent=1)
    print(json.dumps({k: round(v["hbm_bytes_per_launch"] / 1e6, 1)
                      for k, v in res["kernels"].items() if v["hbm_bytes_per_launch"] > 5e6}))


if __name__ == "__main__":
    main()
