"""GPU parity: embedding cache (hctr_cache_*: Query / Replace / Update / Dump of gpu_cache::gpu_cache,
R/gpu_cache/src/nv_gpu_cache.cu) and the host<->HBM tiered table (hctr_tiered_*) against the
sequential restatement in oracle/cache_oracle.py.  Everything is compared exactly: which keys hit,
the order of the missing list, which key every eviction removes, the slot order of Dump."""
import numpy as np
import pytest

pytestmark = pytest.mark.gpu


@pytest.mark.parametrize("key_dtype", ["i64", "u32"])
@pytest.mark.parametrize("D,num_sets", [(16, 3), (10, 5), (128, 2)])
def test_cache_matches_oracle(key_dtype, D, num_sets):
    import torch
    from hugectr_amd.cache import GpuCache
    from oracle.cache_oracle import CacheOracle
    rng = np.random.default_rng(D * 10 + num_sets)
    tdt = torch.int64 if key_dtype == "i64" else torch.int32
    ndt = np.int64 if key_dtype == "i64" else np.int32
    c = GpuCache(num_sets, D, tdt)
    o = CacheOracle(num_sets, D, 8 if key_dtype == "i64" else 4)
    universe = num_sets * 64 * 3                    # 3x the capacity: evictions every round
    table = rng.standard_normal((universe, D)).astype(np.float32)
    for it in range(12):
        n = int(rng.integers(1, 400))
        keys = np.minimum(rng.pareto(0.8, size=n) * 20, universe - 1).astype(ndt)  # skewed, duplicates
        tk = torch.from_numpy(keys).cuda()
        vals = torch.full((n, D), -7.0, device="cuda")
        mi, mk = c.Query(tk, vals)
        want = np.full((n, D), -7.0, np.float32)
        wmi, wmk = o.query(keys, want)
        assert (mi.cpu().numpy() == wmi).all() and mi.numel() == wmi.size, f"missing index it{it}"
        assert (mk.cpu().numpy().astype(np.int64) == wmk).all(), f"missing keys it{it}"
        assert (vals.cpu().numpy() == want).all(), f"hit values it{it}"
        # fetch the missing rows "from the backing store" and insert them
        mkeys = keys[wmi]
        c.Replace(torch.from_numpy(mkeys).cuda(), torch.from_numpy(table[mkeys]).cuda())
        o.replace(mkeys, table[mkeys])
        if it % 3 == 1:                             # refresh some vectors in place
            uk = rng.integers(0, universe, size=150).astype(ndt)
            uv = rng.standard_normal((150, D)).astype(np.float32)
            c.Update(torch.from_numpy(uk).cuda(), torch.from_numpy(uv).cuda())
            o.update(uk, uv)
            last = {}
            for i, k in enumerate(uk):
                last[int(k)] = i
            for k, i in last.items():
                table[k] = uv[i]                    # the backing store sees the update too
        got = c.Dump().cpu().numpy().astype(np.int64)
        assert (got == o.dump(0, num_sets)).all(), f"dump it{it}"
        a, b = sorted(rng.integers(0, num_sets + 1, size=2))
        assert (c.Dump(int(a), int(b)).cpu().numpy().astype(np.int64) == o.dump(int(a), int(b))).all()
    # every cached key returns the oracle's vector
    ks = o.dump(0, num_sets).astype(ndt)
    vals = torch.zeros((ks.size, D), device="cuda")
    mi, _ = c.Query(torch.from_numpy(ks).cuda(), vals)
    want = np.zeros((ks.size, D), np.float32)
    o.query(ks, want)
    assert mi.numel() == 0
    assert (vals.cpu().numpy() == want).all()


def test_cache_large_batch_and_duplicates():
    """one Replace with far more keys than slots and heavy duplication: the survivors of every set
    are the ones the position-ordered walk leaves"""
    import torch
    from hugectr_amd.cache import GpuCache
    from oracle.cache_oracle import CacheOracle
    rng = np.random.default_rng(1)
    D, num_sets = 8, 16
    c, o = GpuCache(num_sets, D), CacheOracle(num_sets, D)
    keys = rng.integers(0, 6000, size=20000).astype(np.int64)
    vals = rng.standard_normal((keys.size, D)).astype(np.float32)
    for _ in range(2):
        c.Query(torch.from_numpy(keys[:10]).cuda())
        o.query(keys[:10], None)
        c.Replace(torch.from_numpy(keys).cuda(), torch.from_numpy(vals).cuda())
        o.replace(keys, vals)
        assert (c.Dump().cpu().numpy() == o.dump(0, num_sets)).all()
    ks = o.dump(0, num_sets)
    got = torch.zeros((ks.size, D), device="cuda")
    c.Query(torch.from_numpy(ks).cuda(), got)
    want = np.zeros((ks.size, D), np.float32)
    o.query(ks, want)
    assert (got.cpu().numpy() == want).all()


@pytest.mark.parametrize("flush", ["each_step", "at_the_end"])
@pytest.mark.parametrize("pieces", [4, 1, 3], ids=["four_pieces", "whole", "three_pieces"])
@pytest.mark.parametrize("D", [32, 6])
def test_tiered_table_matches_oracle(monkeypatch, D, pieces, flush):
    """lookup (hits from the cache, misses over the host link, Replace) + write-through scatter
    against the oracle: values, miss counts, the host table and the cache's key sets after every
    call -- with the lookup as one pass and as a pipeline of pieces whose host-link copies run on
    a private stream next to the following pieces' cache queries (the same cache state by
    construction: one tick of the clock, one Replace over the pieces' misses in position order).
    The cache is write-back: "at_the_end" never flushes between the steps, so updated rows reach the
    host table only by being evicted (and are read back from there by later misses)."""
    import torch
    from hugectr_amd.cache import TieredTable
    from oracle.cache_oracle import TieredOracle
    monkeypatch.setenv("HCTR_TIER_PIECES", str(pieces))
    monkeypatch.setenv("HCTR_TIER_PIECE_MIN", "0")
    rng = np.random.default_rng(D)
    rows, num_sets = 5000, 8                         # cache holds 512 of 5000 rows
    t = TieredTable(rows, D, num_sets)
    o = TieredOracle(rows, D, num_sets)
    init = rng.standard_normal((rows, D)).astype(np.float32)
    t.host[:] = init
    o.host[:] = init
    misses = []
    for it in range(14):
        n = int(rng.integers(50, 900))
        keys = np.minimum(rng.pareto(1.05, size=n) * 30, rows + 5).astype(np.int64)  # some out of range
        out = t.lookup(torch.from_numpy(keys).cuda())
        want, nmiss = o.lookup(keys)
        assert (out.cpu().numpy() == want).all(), f"lookup it{it}"
        assert t.last_missing() == nmiss
        misses.append(nmiss / n)
        uk = np.unique(keys)
        g = rng.standard_normal((uk.size, D)).astype(np.float32)
        if it % 4 == 3:
            t.scatter_update(torch.from_numpy(uk).cuda(), torch.from_numpy(g).cuda())
            o.scatter(uk, g, add=False)
        else:
            t.scatter_add(torch.from_numpy(uk).cuda(), torch.from_numpy(g).cuda())
            o.scatter(uk, g, add=True)
        if flush == "each_step":
            t.flush()
            assert (t.host == o.host).all(), f"host table it{it}"
        else:
            torch.cuda.synchronize()
        assert (t.cache.Dump().cpu().numpy() == o.cache.dump(0, num_sets)).all(), f"cache keys it{it}"
    t.flush()
    assert (t.host == o.host).all(), "host table after the last flush"
    assert misses[-1] < misses[0]                   # the hot rows stay cached
    ks = o.cache.dump(0, num_sets)
    got = torch.zeros((ks.size, D), device="cuda")
    t.cache.Query(torch.from_numpy(ks).cuda(), got)
    want = np.zeros((ks.size, D), np.float32)
    o.cache.query(ks, want)
    assert (got.cpu().numpy() == want).all()


@pytest.mark.parametrize("D", [16, 128])
def test_tiered_embedding_trains_like_a_dense_table(D):
    """BASELINE config 4 in miniature: a one-hot embedding whose table lives in host memory behind
    the cache; forward rows and the SGD-updated table must follow a plain in-memory table"""
    import torch
    from hugectr_amd.cache import TieredEmbedding
    rng = np.random.default_rng(D)
    rows, n, lr = 30000, 4096, 0.05
    emb = TieredEmbedding(rows, D, 4, n, lr=lr)           # 256 cached rows of 30000
    ref = rng.standard_normal((rows, D)).astype(np.float32)
    emb.table.host[:] = ref
    ref = ref.astype(np.float64)
    for it in range(6):
        keys = np.minimum(rng.pareto(1.05, size=n) * 3, rows - 1).astype(np.int64)
        out = emb.forward(torch.from_numpy(keys).cuda())
        assert np.allclose(out.cpu().numpy(), ref[keys], rtol=1e-6, atol=1e-6), f"forward it{it}"
        g = rng.standard_normal((n, D)).astype(np.float32)
        emb.backward_update(torch.from_numpy(g).cuda())
        np.subtract.at(ref, keys, lr * g.astype(np.float64))
    emb.table.flush()
    assert np.allclose(emb.table.host, ref, rtol=2e-5, atol=2e-5)
    # the cached copies agree with the host table (write-through)
    ks = emb.table.cache.Dump()
    got = torch.zeros((ks.numel(), D), device="cuda")
    emb.table.cache.Query(ks, got)
    assert (got.cpu().numpy() == emb.table.host[ks.cpu().numpy()]).all()


@pytest.mark.parametrize("kb", [8, 4])
@pytest.mark.parametrize("D,default", [(32, 0.0), (6, -1.5)])
def test_uvm_table_of_arbitrary_keys_matches_oracle(monkeypatch, D, default, kb):
    """gpu_cache::UvmTable as a whole (hctr_uvm_*): sparse random keys out of a 10^10 key space
    (BASELINE configs[3]: the store cannot be as large as the key space) through add (host keys /
    vectors, keys listed twice, keys added again), query (unknown keys -> the default value), the
    training lookup (first touch takes the next host row) + write-through scatter on the rows it
    hands back, clear -- values, rows, miss counts, the host store and the cache's key sets against
    the oracle after every call; batches longer than max_batch_size run in pieces."""
    import torch
    from hugectr_amd.cache import UvmTable
    from oracle.cache_oracle import UvmOracle
    monkeypatch.setenv("HCTR_TIER_PIECE_MIN", "0")
    rng = np.random.default_rng(D + kb)
    hi = 10**10 if kb == 8 else 2**32 - 2
    kdt, tdt = (np.int64, torch.int64) if kb == 8 else (np.uint32, torch.uint32)
    cap, dev_cap, max_batch = 6000, 512, 700
    t = UvmTable(dev_cap, cap, max_batch, D, default, key_dtype=tdt)
    o = UvmOracle(dev_cap, cap, D, default, max_batch)
    init = rng.standard_normal((cap, D)).astype(np.float32)  # what a first-touched row holds
    t.host[:] = init
    o.tier.host[:] = init
    pool = rng.integers(0, hi, size=9000).astype(np.int64)

    def dev(k):
        a = np.ascontiguousarray(k.astype(kdt))
        return torch.from_numpy(a.view(np.int32) if kb == 4 else a).cuda().view(tdt)

    for it in range(8):
        # add: 400 .. 1500 keys (several pieces), a tenth of them twice, some known already
        n = int(rng.integers(400, 1500))
        ks = pool[rng.integers(0, 500 * (it + 1), size=n)]
        vs = rng.standard_normal((n, D)).astype(np.float32)
        t.add(ks.astype(kdt), vs)
        o.add(ks, vs)
        assert t.size() == len(o.row)
        t.flush()
        assert (t.host == o.tier.host).all(), f"host store after add it{it}"
        # query: known and unknown keys
        q = np.concatenate([pool[rng.integers(0, 500 * (it + 1), size=900)],
                            rng.integers(0, hi, size=100)]).astype(np.int64)
        rng.shuffle(q)
        got = t.query(dev(q)).cpu().numpy()
        assert (got == o.query(q)).all(), f"query it{it}"
        # training lookup of fresh keys + SGD-like write-through on the distinct rows
        lk = pool[rng.integers(0, 600 * (it + 1), size=max_batch)]
        out, rows = t.lookup(dev(lk))
        want, wrows, nmiss = o.lookup(lk)
        assert (rows.cpu().numpy() == wrows).all() and (out.cpu().numpy() == want).all(), f"lookup it{it}"
        assert t.last_missing() == nmiss
        ur = np.unique(wrows)
        g = rng.standard_normal((ur.size, D)).astype(np.float32)
        t.scatter_rows(torch.from_numpy(ur).cuda(), torch.from_numpy(g).cuda(), add=True, alpha=-0.1)
        o.tier.scatter(ur, (g * np.float32(-0.1)).astype(np.float32), add=True)
        t.check_overflow()
        t.flush()
        assert (t.host == o.tier.host).all(), f"host store after scatter it{it}"
        assert (t.cache.Dump().cpu().numpy() == o.tier.cache.dump(0, dev_cap // 64)).all(), f"cache it{it}"
    assert t.size() == len(o.row) > 3000
    t.clear()
    torch.cuda.synchronize()
    assert t.size() == 0 and t.cache.Dump().numel() == 0
    assert (t.query(dev(pool[:50])).cpu().numpy() == np.float32(default)).all()


def test_uvm_table_reports_a_full_host_store():
    import torch
    from hugectr_amd import _lib
    from hugectr_amd.cache import UvmTable
    t = UvmTable(64, 100, 256, 4)
    t.add(np.arange(100, dtype=np.int64) * 7919, np.ones((100, 4), np.float32))
    with pytest.raises(_lib.HugeCTRAmdError, match="more distinct keys"):
        t.add(np.arange(100, 130, dtype=np.int64) * 7919, np.ones((30, 4), np.float32))


@pytest.mark.parametrize("D", [16, 128])
def test_uvm_embedding_of_a_huge_key_space_trains_like_a_dict_of_rows(D):
    """BASELINE configs[3] in miniature: 4 tables with a 10^10 key space each behind ONE UvmTable
    (key = table * 10^10 + key), power-law keys, 256 cached rows of 20 000; forward vectors and the
    SGD-updated store must follow a plain key -> vector dictionary kept in fp64"""
    import torch
    from hugectr_amd.cache import UvmEmbedding
    rng = np.random.default_rng(D)
    cap, n, lr = 20000, 4096, 0.05
    emb = UvmEmbedding(256, cap, D, n, lr=lr)
    init = rng.standard_normal((cap, D)).astype(np.float32)
    emb.table.host[:] = init
    ref, row_of = {}, {}
    for it in range(6):
        tab = rng.integers(0, 4, size=n)
        rank = np.minimum(rng.pareto(1.05, size=n) * 3, 4000).astype(np.int64)
        keys = tab * 10**10 + (rank * 2654435761) % 10**10
        for k in keys.tolist():  # first touch takes the next row of the store, in position order
            if k not in row_of:
                row_of[k] = len(row_of)
                ref[k] = init[row_of[k]].astype(np.float64)
        out = emb.forward(torch.from_numpy(keys).cuda())
        want = np.stack([ref[k] for k in keys.tolist()])
        assert np.allclose(out.cpu().numpy(), want, rtol=1e-6, atol=1e-6), f"forward it{it}"
        g = rng.standard_normal((n, D)).astype(np.float32)
        emb.backward_update(torch.from_numpy(g).cuda())
        for k, gi in zip(keys.tolist(), g.astype(np.float64)):
            ref[k] = ref[k] - lr * gi
    emb.table.check_overflow()
    emb.table.flush()
    assert emb.table.size() == len(row_of)
    got = np.stack([emb.table.host[row_of[k]] for k in ref])
    assert np.allclose(got, np.stack(list(ref.values())), rtol=2e-5, atol=2e-5)
