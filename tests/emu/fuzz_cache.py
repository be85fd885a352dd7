"""TEST INFRASTRUCTURE ONLY: randomized differential test of the GPU embedding cache (hctr_cache_*:
Query / Replace / Update / Dump; kernels' source under the host interpreter, tests/emu) against
oracle/cache_oracle.py over random vector sizes, set counts, key widths, key skews and batch sizes
(the parametrized test, tests/test_cache_gpu.py, with its constants drawn at random).

    python tests/emu/fuzz_cache.py --seed 0 --cases 100"""
import argparse
import os
import sys
import traceback

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
for p in (ROOT, os.path.join(ROOT, "tests"), HERE):
    if p not in sys.path:
        sys.path.insert(0, p)
os.environ["HCTR_EMU"] = "1"

import fakecuda  # noqa: E402

fakecuda.install(os.environ.get("HCTR_EMU_VARIANT"))

import torch  # noqa: E402

from hugectr_amd.cache import GpuCache  # noqa: E402
from oracle.cache_oracle import CacheOracle  # noqa: E402


def one_case(seed):
    rng = np.random.default_rng(seed)
    key_dtype = str(rng.choice(["i64", "u32"]))
    D = int(rng.choice([1, 2, 6, 10, 16, 32, 128, 256]))
    num_sets = int(rng.choice([1, 2, 3, 5, 16, 64]))
    factor = float(rng.choice([0.5, 1.0, 3.0, 20.0]))   # universe / capacity
    skew = float(rng.choice([0.4, 0.8, 2.0]))
    desc = dict(seed=seed, key_dtype=key_dtype, D=D, num_sets=num_sets, factor=factor, skew=skew)
    tdt = torch.int64 if key_dtype == "i64" else torch.int32
    ndt = np.int64 if key_dtype == "i64" else np.int32
    c = GpuCache(num_sets, D, tdt)
    o = CacheOracle(num_sets, D, 8 if key_dtype == "i64" else 4)
    universe = max(int(num_sets * 64 * factor), 2)
    table = rng.standard_normal((universe, D)).astype(np.float32)
    for it in range(int(rng.integers(3, 10))):
        n = int(rng.choice([1, 2, 63, 64, 65, 400, 3000]))
        keys = np.minimum(rng.pareto(skew, size=n) * 20, universe - 1).astype(ndt)
        tk = torch.from_numpy(keys).cuda()
        vals = torch.full((n, D), -7.0, device="cuda")
        mi, mk = c.Query(tk, vals)
        want = np.full((n, D), -7.0, np.float32)
        wmi, wmk = o.query(keys, want)
        assert mi.numel() == wmi.size and (mi.cpu().numpy() == wmi).all(), (desc, it, "missing index")
        assert (mk.cpu().numpy().astype(np.int64) == wmk).all(), (desc, it, "missing keys")
        assert (vals.cpu().numpy() == want).all(), (desc, it, "hit values")
        mkeys = keys[wmi]
        if mkeys.size:
            c.Replace(torch.from_numpy(mkeys).cuda(), torch.from_numpy(table[mkeys]).cuda())
            o.replace(mkeys, table[mkeys])
        if rng.random() < 0.4:
            m = int(rng.choice([1, 150, 1000]))
            uk = rng.integers(0, universe, size=m).astype(ndt)
            uv = rng.standard_normal((m, D)).astype(np.float32)
            c.Update(torch.from_numpy(uk).cuda(), torch.from_numpy(uv).cuda())
            o.update(uk, uv)
            last = {}
            for i, k in enumerate(uk):
                last[int(k)] = i
            for k, i in last.items():
                table[k] = uv[i]
        got = c.Dump().cpu().numpy().astype(np.int64)
        assert (got == o.dump(0, num_sets)).all(), (desc, it, "dump")
        a, b = sorted(rng.integers(0, num_sets + 1, size=2))
        assert (c.Dump(int(a), int(b)).cpu().numpy().astype(np.int64) == o.dump(int(a), int(b))).all(), \
            (desc, it, "partial dump")
    ks = o.dump(0, num_sets).astype(ndt)
    if ks.size:
        vals = torch.zeros((ks.size, D), device="cuda")
        mi, _ = c.Query(torch.from_numpy(ks).cuda(), vals)
        want = np.zeros((ks.size, D), np.float32)
        o.query(ks, want)
        assert mi.numel() == 0 and (vals.cpu().numpy() == want).all(), (desc, "final query")
    return desc


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--seed", type=int, default=0)
    ap.add_argument("--cases", type=int, default=50)
    a = ap.parse_args()
    bad = 0
    for i in range(a.cases):
        seed = a.seed * 1_000_003 + i
        try:
            one_case(seed)
        except Exception as e:  # noqa: BLE001
            bad += 1
            print(f"FAIL seed {seed}: {type(e).__name__} {str(e)[:500]}", flush=True)
            if os.environ.get("FUZZ_TRACE"):
                traceback.print_exc()
    print(f"{a.cases - bad} / {a.cases} cases agree with the oracle", flush=True)
    sys.exit(1 if bad else 0)


if __name__ == "__main__":
    main()
