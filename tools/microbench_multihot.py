"""Gather+pool bandwidth on ragged multi-hot buckets (WDL / MMoE-like lookups), stateless ABI.
Usage: python tools/microbench_multihot.py [--dim 128] [--hot 20] [--dist uniform|skew]"""
import argparse
import json
import os
import sys

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from hugectr_amd import _lib  # noqa: E402


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--buckets", type=int, default=65536 * 8)
    ap.add_argument("--rows", type=int, default=40_000_000)
    ap.add_argument("--iters", type=int, default=10)
    a = ap.parse_args()
    res = []
    for D in (16, 64, 128):
        table = torch.rand((a.rows, D), device="cuda")
        for dist, hot in (("onehot", 1), ("uniform", 8), ("uniform", 40), ("skew", 8), ("skew", 40)):
            rng = np.random.default_rng(1)
            nb = a.buckets if D >= 64 else a.buckets * 4
            if dist == "onehot":
                lens = np.ones(nb, dtype=np.int64)
            elif dist == "uniform":
                lens = rng.integers(0, 2 * hot + 1, size=nb)
            else:  # a few very long buckets among short ones, same mean
                lens = np.minimum(rng.geometric(1.0 / hot, size=nb), 64 * hot)
            ro = np.zeros(nb + 1, dtype=np.int64)
            np.cumsum(lens, out=ro[1:])
            nnz = int(ro[-1])
            vi = torch.from_numpy(rng.integers(0, a.rows, size=nnz).astype(np.int64)).cuda()
            rot = torch.from_numpy(ro).cuda()
            for comb in (0,):
                for odt, oc, esz in ((torch.float32, _lib.F32, 4),):
                    out = torch.empty((nb, D), dtype=odt, device="cuda")

                    fn = (_lib.lib.hctr_forward_pool_multihot if os.environ.get("FLAT") == "1"
                          else _lib.lib.hctr_forward_pool)

                    def run():
                        _lib.check(fn(nb, D, comb, _lib.ptr(rot), _lib.KEY_I64,
                                                              _lib.ptr(vi), _lib.ptr(table),
                                                              _lib.ptr(out), oc, _lib.stream_ptr()))
                    for _ in range(2):
                        run()
                    torch.cuda.synchronize()
                    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                    e0.record()
                    for _ in range(a.iters):
                        run()
                    e1.record()
                    torch.cuda.synchronize()
                    ms = e0.elapsed_time(e1) / a.iters
                    alg = nnz * 8 + nnz * D * 4 + nb * (8 + D * esz)
                    res.append({"D": D, "dist": dist, "hot": hot, "comb": comb, "out": str(odt)[6:],
                                "buckets": nb, "nnz": nnz, "ms": round(ms, 4),
                                "alg_GBps": round(alg / ms / 1e6, 1)})
                    print(res[-1], flush=True)
        del table
    json.dump(res, open(os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))),
                                     "gpurun_out", "multihot.json"), "w"), indent=1)


if __name__ == "__main__":
    main()
