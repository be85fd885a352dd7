"""GPU parity: dynamic embedding table (hctr_det_*) vs the dict/numpy restatement in
oracle/det_oracle.py -- lookup with insertion, growth across re-allocations, scatter_add /
scatter_update (missing keys skipped), remove + re-insert, export, all seven optimizer steps."""
import numpy as np
import pytest

from util import assert_close

pytestmark = pytest.mark.gpu


def _dev(torch, a, dtype=None):
    t = torch.from_numpy(np.ascontiguousarray(a)).cuda()
    return t if dtype is None else t.to(dtype)


def _keys(torch, k, kb):
    if kb == 8:
        return _dev(torch, k.astype(np.int64))
    return _dev(torch, k.astype(np.uint32).view(np.int32))


@pytest.mark.parametrize("kb", [8, 4])
def test_det_lookup_growth_scatter_remove_export(kb):
    import torch
    from hugectr_amd.dynamic_table import DynamicEmbeddingTable
    from oracle.det_oracle import DetOracle
    rng = np.random.default_rng(kb)
    dims = [8, 20]
    t = DynamicEmbeddingTable(dims, "0.25", initial_capacity=64,
                              key_dtype=torch.int64 if kb == 8 else torch.uint32)
    o = DetOracle(dims, 0.25)
    hi = 2**40 if kb == 8 else 2**31
    pool = rng.integers(0, hi, size=3000, dtype=np.int64)
    for it in range(6):  # 64 -> thousands of rows: several re-allocations per class
        n0, n1 = int(rng.integers(50, 700)), int(rng.integers(50, 700))
        keys = pool[rng.integers(0, 400 * (it + 1), size=n0 + n1)]  # duplicates inside a call
        sp, so = [0, 1], [0, n0, n0 + n1]
        got = t.lookup(_keys(torch, keys, kb), sp, so).cpu().numpy()
        want = o.lookup(keys, sp, so)
        assert np.array_equal(got, want), f"lookup it{it}"
        # scatter_add on UNIQUE keys (duplicates would make the float sum order-dependent),
        # a third of them unknown to the table -> skipped
        uk0 = np.unique(np.concatenate([keys[:n0][:40], rng.integers(0, hi, 20)]))
        uk1 = np.unique(np.concatenate([keys[n0:][:40], rng.integers(0, hi, 20)]))
        uk = np.concatenate([uk0, uk1])
        upd = rng.standard_normal(uk0.size * dims[0] + uk1.size * dims[1]).astype(np.float32)
        so2 = [0, uk0.size, uk.size]
        if it % 2 == 0:
            t.scatter_add(_keys(torch, uk, kb), _dev(torch, upd), sp, so2)
        else:
            t.scatter_update(_keys(torch, uk, kb), _dev(torch, upd), sp, so2)
        o.scatter(uk, upd, sp, so2, add=(it % 2 == 0))
        assert t.size_per_class() == o.size_per_class()
    assert min(t.capacity_per_class()) > 64
    # remove some keys (with duplicates and unknown ones), then look them up again: re-initialised
    rm = np.concatenate([pool[:100], pool[:30], rng.integers(0, hi, 10)])
    t.remove(_keys(torch, rm, kb), [0], [0, rm.size])
    o.remove(rm, [0], [0, rm.size])
    assert t.size_per_class() == o.size_per_class()
    again = pool[:150]
    got = t.lookup(_keys(torch, again, kb), [0], [0, again.size]).cpu().numpy()
    assert np.array_equal(got, o.lookup(again, [0], [0, again.size]))
    # export == the oracle's map, as a key -> vector dictionary
    for c in range(2):
        k, v = t.export(c)
        k = k.cpu().numpy()
        k = k.astype(np.int64) if kb == 8 else k.view(np.uint32).astype(np.int64)
        v = v.cpu().numpy()
        assert len(k) == len(o.maps[c]) and len(set(k.tolist())) == len(k)
        for kk, vv in zip(k.tolist(), v):
            assert np.array_equal(vv, o.maps[c][kk])
    t.clear()
    assert t.size() == 0


def test_det_random_initializer_is_uniform_and_deterministic():
    import torch
    from hugectr_amd.dynamic_table import DynamicEmbeddingTable
    keys = torch.arange(0, 20000, dtype=torch.int64, device="cuda")
    a = DynamicEmbeddingTable([16], "", 1024, seed=7).lookup(keys).cpu().numpy()
    b = DynamicEmbeddingTable([16], "random", 1 << 16, seed=7).lookup(keys).cpu().numpy()
    c = DynamicEmbeddingTable([16], "", 1024, seed=8).lookup(keys).cpu().numpy()
    assert np.array_equal(a, b), "initial values must not depend on capacity / growth history"
    assert not np.array_equal(a, c)
    assert a.min() > 0.0 and a.max() <= 1.0          # curand_uniform's range (0, 1]
    assert abs(a.mean() - 0.5) < 5e-3 and abs(a.var() - 1 / 12) < 2e-3


OPTS = ["sgd", "momentum", "nesterov", "adagrad", "rmsprop", "adam", "ftrl"]


@pytest.mark.parametrize("name", OPTS)
def test_det_update_matches_oracle(name):
    import torch
    from hugectr_amd import _lib
    from hugectr_amd.dynamic_table import DynamicEmbeddingTable, DynamicTableOptimizer
    from oracle import det_oracle as D
    code = {"sgd": (_lib.OPT_SGD, D.SGD), "momentum": (_lib.OPT_MOMENTUM_SGD, D.MOMENTUM),
            "nesterov": (_lib.OPT_NESTEROV, D.NESTEROV), "adagrad": (_lib.OPT_ADAGRAD, D.ADAGRAD),
            "rmsprop": (_lib.OPT_RMSPROP, D.RMSPROP), "adam": (_lib.OPT_ADAM, D.ADAM),
            "ftrl": (_lib.OPT_FTRL, D.FTRL)}[name]
    rng = np.random.default_rng(5)
    dims = [16, 6]
    ns = {"sgd": 0, "adam": 2, "ftrl": 2}.get(name, 1)
    kw = dict(lr=0.05, scaler=2.0, beta1=0.9, beta2=0.999, epsilon=1e-6, momentum_factor=0.8,
              rmsprop_beta=0.95, ftrl_lambda1=0.01, ftrl_lambda2=0.02, ftrl_beta=0.5)
    t = DynamicEmbeddingTable(dims, "0.5", 128)
    opt = DynamicTableOptimizer(t, code[0], initial_capacity=128, **kw)
    ow = D.DetOracle(dims, 0.5)
    os_ = D.DetOracle([d * max(ns, 1) for d in dims], 0.0)
    pool = rng.integers(0, 2**40, size=600, dtype=np.int64)
    for step in range(4):
        n0, n1 = int(rng.integers(40, 200)), int(rng.integers(40, 200))
        k0 = np.unique(pool[rng.integers(0, 600, size=n0)])
        k1 = np.unique(pool[rng.integers(0, 600, size=n1)])
        keys = np.concatenate([k0, k1])
        sp, so = [0, 1], [0, k0.size, keys.size]
        # forward lookup first (training order), except a few keys Ftrl must insert by itself
        seen = keys if name != "ftrl" else np.concatenate([k0[5:], k1])
        so_seen = so if name != "ftrl" else [0, k0.size - 5, seen.size]
        t.lookup(_dev(torch, seen), sp, so_seen)
        ow.lookup(seen, sp, so_seen)
        lens = np.array([dims[0]] * k0.size + [dims[1]] * k1.size)
        ev = np.concatenate([[0], np.cumsum(lens)]).astype(np.int32)
        wg = rng.standard_normal(int(ev[-1])).astype(np.float32)
        opt.update(_dev(torch, keys), _dev(torch, ev), _dev(torch, wg), sp, so)
        D.update(ow, os_, code[1], keys, sp, so, ev, wg, lr=kw["lr"], scaler=kw["scaler"],
                 beta1=kw["beta1"], beta2=kw["beta2"], eps=kw["epsilon"],
                 momentum=kw["momentum_factor"], rms_beta=kw["rmsprop_beta"],
                 lambda1=kw["ftrl_lambda1"], lambda2=kw["ftrl_lambda2"], ftrl_beta=kw["ftrl_beta"],
                 times=step + 1)
        got = t.lookup(_dev(torch, keys), sp, so).cpu().numpy()
        want = ow.lookup(keys, sp, so)
        assert_close(got, want, 1e-5, 1e-6, f"{name} weights step {step}")
        if ns:
            gs = opt.states.lookup(_dev(torch, keys), sp, so).cpu().numpy()
            assert_close(gs, os_.lookup(keys, sp, so), 1e-5, 1e-7, f"{name} state step {step}")
    assert t.size_per_class() == ow.size_per_class()


def _rows_of(torch, addr, rows, ev):
    """fp32 rows `rows` of the flat [.][ev] array at device address addr (the library's own gather)"""
    from hugectr_amd import _lib
    from hugectr_amd._lib import check, lib, ptr, stream_ptr
    n = rows.numel()
    out = torch.empty((n, ev), dtype=torch.float32, device="cuda")
    rng = torch.arange(n + 1, dtype=torch.int64, device="cuda")
    check(lib.hctr_forward_pool(n, ev, 0, ptr(rng), _lib.KEY_I64, ptr(rows), addr, ptr(out),
                                _lib.F32, stream_ptr()))
    torch.cuda.synchronize()
    return out


def test_state_store_shares_the_row_numbers_stays_put_through_growth_and_clears():
    """hctr_det_state_store: zero for rows never updated, one array more when a later caller asks for
    two; when a class grows NOTHING moves -- the row store's address, every class's row numbers and
    the state behind them stay (memory is mapped behind the grown class, zero-filled for the
    state); zeroed by clear(); refused for classes of several dimensions (no flat row store)."""
    import ctypes
    import torch
    from hugectr_amd import _lib
    from hugectr_amd._lib import check, lib, ptr, stream_ptr
    from hugectr_amd.dynamic_table import DynamicEmbeddingTable
    ev = 8
    t = DynamicEmbeddingTable([ev, ev, ev], "0.5", initial_capacity=16)
    k0 = torch.arange(100, 110, dtype=torch.int64).cuda()
    k2 = torch.arange(500, 506, dtype=torch.int64).cuda()
    keys = torch.cat([k0, k2])
    _, rows, base = t.lookup_rows(keys, [0, 2], [0, 10, 16], want_ptrs=False)
    stride = base[1]  # rows of address range per class: a power of two, classes * stride < 2^32
    assert base == [0, stride, 2 * stride, 3 * stride] and stride & (stride - 1) == 0
    assert 3 * stride < 2**32 - 16 and stride >= 2**24
    assert int(rows[:10].min()) == 0 and int(rows[10:].min()) == 2 * stride
    store0, total0 = t.row_store()
    assert total0 == 3 * stride
    s0, s1 = t.state_store(1)
    assert s0 and not s1
    assert float(_rows_of(torch, s0, rows, ev).abs().max()) == 0.0
    # one AdaGrad step of the static tables' update on those rows: state = (g / scaler)^2
    upd = ctypes.c_void_p()
    check(lib.hctr_updater_create(64, 0xFFFFFFEF, ev, ctypes.byref(upd)))
    try:
        g = torch.arange(1, 16 * ev + 1, dtype=torch.float32).cuda().view(16, ev) / 64
        br = torch.arange(17, dtype=torch.int64).cuda()
        store, total = t.row_store()
        check(lib.hctr_updater_set_row_bound(upd, total))
        check(lib.hctr_updater_update(upd, 16, 16, ptr(br), ptr(rows), ptr(g), _lib.F32,
                                      _lib.OPT_ADAGRAD, _lib.UPDATE_LOCAL, 0.1, 0.9, 0.999, 1e-6,
                                      0.0, 2.0, 1, store, s0, None, stream_ptr()))
        want = (g / 2.0) ** 2
        assert torch.equal(_rows_of(torch, s0, rows, ev), want)
        # a second array for a later caller: the first keeps its place and content
        a0, a1 = t.state_store(2)
        assert a0 == s0 and a1
        assert torch.equal(_rows_of(torch, a0, rows, ev), want)
        assert float(_rows_of(torch, a1, rows, ev).abs().max()) == 0.0
        # class 1 grows (16 -> 64): nobody's rows move, neither does the state
        k1 = torch.arange(1000, 1040, dtype=torch.int64).cuda()
        t.lookup_rows(k1, [1], [0, 40], want_ptrs=False)
        assert t.capacity_per_class() == [16, 64, 16]
        _, rows2, base2 = t.lookup_rows(keys, [0, 2], [0, 10, 16], insert=False, want_ptrs=False)
        assert base2 == base and torch.equal(rows2, rows)
        assert t.row_store() == (store0, total0)
        b0, b1 = t.state_store(2)
        assert (b0, b1) == (a0, a1)
        assert torch.equal(_rows_of(torch, b0, rows2, ev), want)
        assert float(_rows_of(torch, b1, rows2, ev).abs().max()) == 0.0
        _, r1, _ = t.lookup_rows(k1, [1], [0, 40], insert=False, want_ptrs=False)
        assert float(_rows_of(torch, b0, r1, ev).abs().max()) == 0.0  # new rows: zero state
        # clear(): rows are handed out from 0 again and start from zero state
        t.clear()
        _, rows3, _ = t.lookup_rows(keys, [0, 2], [0, 10, 16], want_ptrs=False)
        c0, _ = t.state_store(2)
        assert float(_rows_of(torch, c0, rows3, ev).abs().max()) == 0.0
    finally:
        lib.hctr_updater_destroy(upd)
    mixed = DynamicEmbeddingTable([8, 16], "0.5", initial_capacity=16)
    assert mixed.row_store() == (None, 0)
    with pytest.raises(_lib.HugeCTRAmdError, match="flat row store"):
        mixed.state_store(1)


def test_an_index_that_gives_up_is_repaired_inside_the_call_never_a_silent_zero(monkeypatch):
    """a finish kernel whose grid barrier does not open used to leave "no row"
    at the batch's unseen keys -- pooled as zeros, skipped by the update, reported by nobody.
    HCTR_HT_SPIN_LIMIT=0 forces that path on every inserting lookup with more than one finish
    workgroup: the calls must still return what an undisturbed table returns (same row numbers,
    same vectors, same sizes), and the table must say that it repaired them."""
    import torch
    from hugectr_amd.dynamic_table import DynamicEmbeddingTable
    rng = np.random.default_rng(3)
    dims = [16, 16, 16]

    def run(disturbed):
        t = DynamicEmbeddingTable(dims, "", initial_capacity=4096, seed=11)
        outs = []
        r = np.random.default_rng(5)
        for it in range(3):
            ns = [int(r.integers(9000, 30000)) for _ in dims]
            keys = np.concatenate([r.integers(0, 60000 * (it + 1), size=n) for n in ns]).astype(np.int64)
            so = np.concatenate([[0], np.cumsum(ns)]).tolist()
            if disturbed:
                monkeypatch.setenv("HCTR_HT_SPIN_LIMIT", "0")
            kt = _dev(torch, keys)
            if it == 1:  # the plain lookup (vectors packed back to back)
                outs.append(t.lookup(kt, [0, 1, 2], so).cpu().numpy())
            else:        # the embedding_collection's call: table-wide row numbers
                _, rows, base = t.lookup_rows(kt, [0, 1, 2], so, insert=True, want_ptrs=False)
                outs.append(rows.cpu().numpy())
                outs.append(np.asarray(base))
            monkeypatch.delenv("HCTR_HT_SPIN_LIMIT", raising=False)
            outs.append(np.asarray(t.size_per_class()))
        # every key of the last batch is in the table (a find-only pass meets no "no row")
        _, rows, _ = t.lookup_rows(kt, [0, 1, 2], so, insert=False, want_ptrs=False)
        assert int((rows < 0).sum()) == 0
        return outs, t.repair_count()

    clean, n_clean = run(False)
    forced, n_forced = run(True)
    assert n_clean == 0 and n_forced >= 3, (n_clean, n_forced)
    assert len(clean) == len(forced)
    for a, b in zip(clean, forced):
        assert np.array_equal(a, b)
