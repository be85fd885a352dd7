"""Embedding cache / tiered table throughput (MI355X, D = 128 fp32 rows):
Query at 100 % hits, Replace of new keys, tiered lookup at the hit rate a power-law stream settles
at (misses come out of pinned host memory over the host link)."""
import json
import os
import sys

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from hugectr_amd.cache import GpuCache, TieredTable  # noqa: E402
from hugectr_amd._lib import check, lib, ptr, stream_ptr  # noqa: E402
from microbench_embedding import powerlaw  # noqa: E402


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
    D, n = 128, 1 << 20
    sets = 1 << 16                                   # 4.2 M slots, 2.1 GB of vectors
    rng = np.random.default_rng(0)
    res = {"vec": D, "keys_per_call": n, "slots": sets * 64}
    c = GpuCache(sets, D)
    resident = torch.from_numpy(rng.permutation(sets * 40)[:sets * 32].astype(np.int64)).cuda()
    vals = torch.randn((resident.numel(), D), device="cuda")
    c.Query(resident[:1])
    c.Replace(resident, vals)
    del vals
    q = resident[torch.randint(0, resident.numel(), (n,), device="cuda")]
    out = torch.empty((n, D), device="cuda")
    mi = torch.empty(n, dtype=torch.int64, device="cuda")
    mk = torch.empty(n, dtype=torch.int64, device="cuda")
    ml = torch.zeros(1, dtype=torch.int64, device="cuda")

    def query():
        check(lib.hctr_cache_query(c._h, ptr(q), n, ptr(out), ptr(mi), ptr(mk), ptr(ml), stream_ptr()))
    t = timed(query)
    res["query_hit_us"] = round(t, 1)
    res["query_hit_rate"] = 1.0 - int(ml.item()) / n
    # bytes: key + set line (512 B) + vector read + vector write per key
    res["query_GBps"] = round(n * (8 + 512 + 2 * D * 4) / t / 1e3, 1)
    fresh = torch.from_numpy((sets * 64 + rng.permutation(4 * n)[:n]).astype(np.int64)).cuda()
    fv = torch.randn((n, D), device="cuda")
    res["replace_new_us"] = round(timed(lambda: c.Replace(fresh, fv), it=3), 1)
    res["update_us"] = round(timed(lambda: c.Update(fresh, fv), it=3), 1)
    del c, out, fv

    rows = 20_000_000                                # 10 GB host table, 2.1 GB cache
    tt = TieredTable(rows, D, sets)
    tt.host[:] = 1.0
    # the reference's power-law key generator (data_generator.hpp:118-123), alpha 1.1, scattered
    # over the table by a fixed multiplicative permutation
    stream = [torch.from_numpy((powerlaw(rng, n, rows, 1.1) * 7919) % rows).cuda() for _ in range(8)]
    for k in stream:                                 # warm the cache
        tt.lookup(k)
    o = torch.empty((n, D), device="cuda")
    # steady state: every timed call sees a NEW batch of the same distribution
    fresh = [torch.from_numpy((powerlaw(rng, n, rows, 1.1) * 7919) % rows).cuda() for _ in range(8)]
    misses = [torch.zeros(1, dtype=torch.int64, device="cuda") for _ in fresh]
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for k, m in zip(fresh, misses):
        check(lib.hctr_tiered_lookup(tt._h, ptr(k), n, ptr(o), ptr(m), stream_ptr()))
    e1.record()
    torch.cuda.synchronize()
    res["tiered_lookup_us"] = round(e0.elapsed_time(e1) / len(fresh) * 1e3, 1)
    res["tiered_miss_rate"] = round(sum(int(m.item()) for m in misses) / (n * len(fresh)), 4)
    k = fresh[-1]
    miss = misses[-1]
    cold = torch.from_numpy(rng.integers(rows // 2, rows, size=n).astype(np.int64)).cuda()
    lookup_cold = lambda: check(lib.hctr_tiered_lookup(tt._h, ptr(cold), n, ptr(o), ptr(miss), stream_ptr()))
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    lookup_cold()
    e1.record()
    torch.cuda.synchronize()
    res["tiered_cold_lookup_us"] = round(e0.elapsed_time(e1) * 1e3, 1)
    res["tiered_cold_miss_rate"] = int(miss.item()) / n
    res["tiered_cold_host_GBps"] = round(int(miss.item()) * D * 4 / (e0.elapsed_time(e1) * 1e-3) / 1e9, 1)
    uk = torch.unique(k)
    g = torch.randn((uk.numel(), D), device="cuda")
    res["tiered_scatter_add_us"] = round(timed(lambda: tt.scatter_add(uk, g), it=3), 1)
    res["tiered_scatter_rows"] = int(uk.numel())
    print(json.dumps(res))


if __name__ == "__main__":
    main()
