"""rocprofv3 --pmc SQ_* pass of bench.py -> per-kernel means (second half of the launches) and the
derived fractions: matrix-pipe utilisation and where the waves' cycles go.

  python tools/pmc_sq.py counter_collection.csv out.json --csrc-hash H --workload "..."
Units (MI355X_MICROARCH.md, counter table): SQ_WAVE_CYCLES / SQ_WAIT_* / SQ_ACTIVE_INST_* count
quad-cycles summed over waves; SQ_VALU_MFMA_BUSY_CYCLES and SQ_BUSY_CYCLES count cycles;
GRBM_GUI_ACTIVE is summed over the 8 XCDs."""
import argparse
import csv
import json
import re
import statistics
from collections import defaultdict


def short(name):
    m = re.match(r"\s*(?:void\s+)?(?:hctr::)?(?:\(anonymous namespace\)::)?([A-Za-z_]\w*)", name)
    return m.group(1) if m else name


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("csv")
    ap.add_argument("out")
    ap.add_argument("--csrc-hash", default="")
    ap.add_argument("--commit", default="")
    ap.add_argument("--workload", default="")
    a = ap.parse_args()
    per = defaultdict(lambda: defaultdict(list))
    for r in csv.DictReader(open(a.csv)):
        k = r["Kernel_Name"]
        if "hctr::" not in k and "hctr_" not in k:
            continue
        per[short(k)][r["Counter_Name"]].append(float(r["Counter_Value"]))
    res = {"method": "rocprofv3 --pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_WAIT_ANY "
                     "SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY GRBM_GUI_ACTIVE --kernel-trace (own pass, no "
                     "other trace domain); mean of the second half of every kernel's launches; "
                     "GRBM_GUI_ACTIVE is summed over the 8 XCDs; SQ_*_CYCLES wave counters are "
                     "quad-cycles, SQ_VALU_MFMA_BUSY_CYCLES cycles (32 per v_mfma_f32_32x32x16)",
           "workload": a.workload, "commit": a.commit, "csrc_hash": a.csrc_hash, "kernels": {}}
    for k, cs in sorted(per.items()):
        m = {c: statistics.mean(v[len(v) // 2:]) for c, v in cs.items() if v}
        if m.get("SQ_WAVE_CYCLES", 0) <= 0:
            continue
        gui = m.get("GRBM_GUI_ACTIVE", 0.0) / 8.0  # cycles the kernel was on the device
        d = dict(m)
        d["launches"] = len(next(iter(cs.values())))
        if gui > 0:
            d["mfma_util_of_1024_simds"] = m.get("SQ_VALU_MFMA_BUSY_CYCLES", 0.0) / (gui * 1024.0)
        w = m["SQ_WAVE_CYCLES"]
        d["wave_cycles_split"] = {"active": m.get("SQ_ACTIVE_INST_ANY", 0.0) / w,
                                  "wait_any(waitcnt/barrier)": m.get("SQ_WAIT_ANY", 0.0) / w,
                                  "wait_inst_any(issue stall)": m.get("SQ_WAIT_INST_ANY", 0.0) / w}
        res["kernels"][k] = d
    json.dump(res, open(a.out, "w"), indent=1)
    print(json.dumps({k: {"mfma": round(v.get("mfma_util_of_1024_simds", 0), 4),
                          "wait": round(v["wave_cycles_split"]["wait_any(waitcnt/barrier)"], 3)}
                      for k, v in res["kernels"].items()}))


if __name__ == "__main__":
    main()
