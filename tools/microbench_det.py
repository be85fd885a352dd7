"""Dynamic embedding table throughput: lookup (steady state), lookup with inserts, fused Adam step.
Usage: python tools/microbench_det.py"""
import json
import os
import sys

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from hugectr_amd import _lib  # noqa: E402
from hugectr_amd.dynamic_table import DynamicEmbeddingTable, DynamicTableOptimizer  # noqa: E402


def timed(fn, it=10):
    for _ in range(2):
        fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(it):
        fn()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / it * 1e3


def main():
    res = []
    rng = np.random.default_rng(0)
    for D in (16, 128):
        t = DynamicEmbeddingTable([D], "", 1 << 22)
        n = 1 << 20
        pool = torch.from_numpy(rng.integers(0, 2**40, size=4_000_000, dtype=np.int64)).cuda()
        keys = pool[torch.randint(0, 3_000_000, (n,), device="cuda")]
        t.lookup(keys)  # insert
        us = timed(lambda: t.lookup(keys))
        res.append({"op": "lookup (all keys known)", "D": D, "keys": n, "us": round(us, 1),
                    "GBps": round(n * (D * 8 + 8 + 16) / us / 1e3, 1)})
        uk = torch.unique(keys)
        opt = DynamicTableOptimizer(t, _lib.OPT_ADAM, 0.001, initial_capacity=1 << 22)
        ev = torch.arange(0, (uk.numel() + 1) * D, D, dtype=torch.int32, device="cuda")
        wg = torch.randn(uk.numel() * D, device="cuda")
        opt.update(uk, ev, wg)
        us = timed(lambda: opt.update(uk, ev, wg))
        res.append({"op": "adam update (unique keys)", "D": D, "keys": uk.numel(),
                    "us": round(us, 1), "GBps": round(uk.numel() * D * 4 * 7 / us / 1e3, 1)})
        fresh = [pool[3_000_000 + i * 100_000:3_000_000 + (i + 1) * 100_000] for i in range(10)]
        i = [0]
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for f in fresh:
            t.lookup(f)
        e1.record()
        torch.cuda.synchronize()
        res.append({"op": "lookup (100k unseen keys / call)", "D": D, "keys": 100_000,
                    "us": round(e0.elapsed_time(e1) / 10 * 1e3, 1)})
    for r in res:
        print(r)
    json.dump(res, open(os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))),
                                     "gpurun_out", "det.json"), "w"), indent=1)


if __name__ == "__main__":
    main()
