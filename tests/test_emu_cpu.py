"""The kernels' OWN SOURCE stepped through on the CPU (tests/emu: a lane-by-lane interpreter of the
HIP execution model -- test infrastructure, never part of the product) and checked against the
oracle: kernel logic (indexing, ranks, scans, barriers where data crosses lanes, the MFMA operand
layout) is verified here before GPU time is spent; the `-m gpu` tests remain the parity tests on
the hardware.  Sizes are small: the interpreter runs ~10^6 lane-steps a second."""
import os
import sys

import numpy as np
import pytest

from util import assert_close, make_csr

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "emu"))
import emu  # noqa: E402

pytestmark = pytest.mark.skipif(not emu.available(), reason="no host clang++ / make")


@pytest.fixture(scope="module")
def elib():
    # HCTR_EMU_VARIANT=<dir>: the files of <dir> laid over hugectr_amd/csrc (a kernel variant is
    # run through this whole file before it replaces anything)
    var = os.environ.get("HCTR_EMU_VARIANT")
    lib = emu.load(var, os.path.basename(os.path.normpath(var))) if var else emu.load()
    emu.bind(lib)
    return lib


def test_hash_rows_match_the_sequential_oracle_over_an_irregular_sequence(oracle, elib):
    """tests/test_hash_gpu.py::test_get_insert_many_batches_of_every_shape at a tenth of the size,
    plus the sequence that exposed the stale region counters of round 3 (a one-workgroup finish
    kernel between two batches of > 1024 regions)."""
    for key_bytes in (8, 4):
        rng = np.random.default_rng(17 + key_bytes)
        cap = 450_000
        g = emu.HashTable(elib, cap, 0 if key_bytes == 4 else 1)
        ref = oracle.HashTable(cap, key_bytes)
        universe = rng.permutation(1_000_000)[:700_000].astype(np.int64) * (3 if key_bytes == 8 else 1)
        seen = 0
        sizes = [100_000, 5, 100_000, 1, 63, 64, 65, 4095, 4096, 4097, 13_107, 1000, 257, 31, 8191, 3]
        for it, n in enumerate(sizes + sizes[3:]):
            kind = 0 if it < 3 else it % 4
            new = [n, 0, max(1, n // 25), n // 2][kind]
            new = min(new, universe.size - seen)
            fresh = universe[seen:seen + new]
            seen += new
            old = universe[rng.integers(0, seen - new, size=n - new)] if n > new else fresh[:0]
            keys = np.concatenate([fresh, old])
            if kind == 3 and n > 8:  # hundreds of copies of a handful of keys, new ones included
                hot = keys[rng.integers(0, keys.size, size=6)]
                keys[rng.integers(0, n, size=n // 3)] = hot[rng.integers(0, 6, size=n // 3)]
            rng.shuffle(keys)
            kk = np.ascontiguousarray(keys if key_bytes == 8 else keys.astype(np.uint32))
            got, want = g.get_insert(kk), ref.get_insert(keys)
            assert (got == want).all(), (key_bytes, it, n, kind, int((got != want).sum()))
            if it % 5 == 4:
                probe = np.concatenate([universe[:min(seen, 500)], universe[-300:]])
                pk = np.ascontiguousarray(probe if key_bytes == 8 else probe.astype(np.uint32))
                assert (g.get_mark(pk) == ref.get_mark(probe)).all()
                assert g.value_head() == ref.size()
        k1, v1 = g.dump()
        k2, v2 = ref.dump()
        o1, o2 = np.argsort(k1), np.argsort(k2)
        assert (k1[o1] == k2[o2]).all() and (v1[o1] == v2[o2]).all()


@pytest.mark.parametrize("end_bit", [10, 20, 22, 24, 31, 32])
def test_radix_sort_is_a_stable_sort(elib, end_bit):
    rng = np.random.default_rng(end_bit)
    for n in (1, 64, 4095, 4097, 20_000):
        hi = (1 << end_bit) - 1
        keys = rng.integers(0, hi, size=n, endpoint=True).astype(np.uint32)
        keys[rng.integers(0, n, size=n // 3)] = keys[0]  # long runs of one key
        vals = np.arange(n, dtype=np.uint32)
        ko, vo = emu.radix_sort_pairs(elib, keys, vals, end_bit)
        order = np.argsort(keys, kind="stable")
        assert (ko == keys[order]).all() and (vo == vals[order]).all()


OPTS = [
    ("sgd", dict(optimizer=6, atomic_update=0)),
    ("adam_local", dict(optimizer=1, update_type=0)),
    ("adam_lazy", dict(optimizer=1, update_type=2)),
    ("adagrad", dict(optimizer=3)),
    ("momentum_global", dict(optimizer=5, update_type=1, momentum_factor=0.9)),
    ("nesterov_local", dict(optimizer=4, update_type=0, momentum_factor=0.9)),
]


@pytest.mark.parametrize("name,kw", OPTS, ids=[o[0] for o in OPTS])
@pytest.mark.parametrize("D,combiner", [(16, 1), (128, 0)])
def test_train_steps_match_oracle(oracle, elib, name, kw, D, combiner):
    """tests/test_embedding_gpu.py::test_train_steps_match_oracle, kernel source on the CPU: index
    stage, gather / pooling, backward and the sorted sparse update of several batches."""
    from hugectr_amd import _lib
    rng = np.random.default_rng(7)
    B, S, hot, vps = 64, 6, 4, 40
    V = S * vps + 16
    opt = dict(lr=0.05, scaler=4.0, beta1=0.9, beta2=0.999, epsilon=1e-7, **kw)
    emb = emu.Embedding(elib, _lib.EMB_LOCALIZED, B, V, D, S * hot, S, combiner, opt)
    table = emb.table().copy()
    ns = {1: 2, 3: 1, 5: 1, 4: 1, 6: 0}[kw["optimizer"]]
    s0 = np.zeros_like(table) if ns >= 1 else None
    s1 = np.zeros_like(table) if ns >= 2 else None
    pt = np.ones(table.shape, dtype=np.uint64) if name == "adam_lazy" else None
    ht = oracle.HashTable(V, 8)
    m = {1: oracle.OPT_ADAM, 3: oracle.OPT_ADAGRAD, 5: oracle.OPT_MOMENTUM, 4: oracle.OPT_NESTEROV,
         6: oracle.OPT_SGD}
    for it in range(3):
        ro, keys = make_csr(rng, B, S, hot, vps, one_hot=(combiner == 0 and it % 2 == 0))
        out = emb.forward(True, ro, keys)
        vi = ht.get_insert(keys)
        assert (emb.value_index(keys.size) == vi).all()
        want = oracle.forward(ro, vi, table, D, combiner)
        assert (out.reshape(-1, D).view(np.uint32) == want.view(np.uint32)).all(), "forward"
        g = rng.standard_normal((B * S, D)).astype(np.float32)
        emb.backward(g.reshape(B, S, D))
        emb.update_params()
        o = oracle.OptParamsC()
        o.optimizer, o.update_type, o.lr = m[kw["optimizer"]], kw.get("update_type", 0), 0.05
        o.beta1, o.beta2, o.epsilon = 0.9, 0.999, 1e-7
        o.momentum_factor, o.scaler, o.times = kw.get("momentum_factor", 0.0), 4.0, it + 1
        oracle.update_params(ro, vi, oracle.backward(ro, g, D, combiner), o, table, s0, s1, pt)
        assert_close(emb.table(), table, 1e-5, 1e-6, f"{name} table it{it}")
        if s0 is not None:
            assert_close(emb.opt_state(0), s0, 1e-5, 1e-6, f"{name} state0 it{it}")
        if s1 is not None:
            assert_close(emb.opt_state(1), s1, 1e-5, 1e-7, f"{name} state1 it{it}")


def test_power_law_update_walks_long_runs(oracle, elib):
    """rows with thousands of gradients (the long-run lists of seg_reduce, seg_combine and the
    chunked combine of the largest runs) next to rows with one: the table after SGD equals the
    oracle's within the fp32 re-association of the long runs"""
    from hugectr_amd import _lib
    rng = np.random.default_rng(3)
    B, S, D = 4096, 4, 32
    vps = 3000
    V = S * vps
    # slot 0: three rows share all samples; slot 1: power law; slots 2, 3: nearly unique
    k = np.empty((B, S), dtype=np.int64)
    k[:, 0] = rng.integers(0, 3, size=B)
    k[:, 1] = np.minimum((rng.pareto(1.1, size=B) * 2).astype(np.int64), vps - 1) + vps
    k[:, 2] = rng.integers(0, vps, size=B) + 2 * vps
    k[:, 3] = rng.integers(0, vps, size=B) + 3 * vps
    keys = k.reshape(-1)
    ro = np.arange(B * S + 1, dtype=np.int64)
    emb = emu.Embedding(elib, _lib.EMB_LOCALIZED, B, V, D, S, S, 0,
                        dict(optimizer=6, lr=0.1, scaler=1.0, atomic_update=0))
    table = emb.table().copy()
    ht = oracle.HashTable(V, 8)
    for it in range(2):
        emb.forward(True, ro, keys)
        vi = ht.get_insert(keys)
        g = rng.standard_normal((B * S, D)).astype(np.float32)
        emb.backward(g.reshape(B, S, D))
        emb.update_params()
        o = oracle.OptParamsC()
        o.optimizer, o.update_type, o.lr, o.scaler, o.times = oracle.OPT_SGD, 0, 0.1, 1.0, it + 1
        oracle.update_params(ro, vi, g, o, table)
        assert_close(emb.table(), table, 2e-4, 2e-4, f"table it{it}")


@pytest.mark.parametrize("dtype,npd,tol", [(0, np.float32, 2e-3), (1, np.float16, 4e-2)])
@pytest.mark.parametrize("B,n_emb,W", [(64, 26, 128), (37, 3, 32), (70, 7, 64)])
def test_interaction_kernels(oracle, elib, dtype, npd, tol, B, n_emb, W):
    """the MFMA interaction kernels with the matrix instruction modelled lane for lane (operand
    layout of the CDNA3/4 ISA): forward and both gradients against the fp32 oracle"""
    rng = np.random.default_rng(B)
    mlp = rng.standard_normal((B, W)).astype(npd)
    emb = rng.standard_normal((B, n_emb, W)).astype(npd)
    n_ins = n_emb + 1
    out = np.empty((B, W + n_ins * (n_ins - 1) // 2 + 1), dtype=npd)
    emu.check(elib, elib.hctr_interaction_fwd(B, n_emb, W, emu.ptr(mlp), emu.ptr(emb),
                                              emu.ptr(out), dtype, None))
    want = oracle.interaction_fwd(mlp.astype(np.float32), emb.astype(np.float32))
    assert_close(out.astype(np.float32), want, tol, tol, "forward")
    g = rng.standard_normal(out.shape).astype(npd)
    mg, eg = np.empty_like(mlp), np.empty_like(emb)
    emu.check(elib, elib.hctr_interaction_bwd(B, n_emb, W, emu.ptr(mlp), emu.ptr(emb), emu.ptr(g),
                                              emu.ptr(mg), emu.ptr(eg), dtype, None))
    wmg, weg = oracle.interaction_bwd(mlp.astype(np.float32), emb.astype(np.float32),
                                      g.astype(np.float32))
    assert_close(mg.astype(np.float32), wmg, tol, tol, "mlp grad")
    assert_close(eg.astype(np.float32), weg, tol, tol, "emb grad")


@pytest.mark.parametrize("W", [32, 128])
def test_gather_fused_into_interaction_equals_pool_then_interaction(oracle, elib, W):
    """hctr_interaction_fwd_gather (table rows read through value_index straight into the
    interaction's tile) is bit-identical to the pooled vectors + hctr_interaction_fwd"""
    rng = np.random.default_rng(W)
    B, n_emb, V = 70, 9, 500
    table = rng.standard_normal((V, W)).astype(np.float32)
    vi = rng.integers(0, V, size=B * n_emb).astype(np.uint64)
    mlp = rng.standard_normal((B, W)).astype(np.float16)
    n_ins = n_emb + 1
    out = np.empty((B, W + n_ins * (n_ins - 1) // 2 + 1), dtype=np.float16)
    pooled = np.empty((B, n_emb, W), dtype=np.float16)
    emu.check(elib, elib.hctr_interaction_fwd_gather(B, n_emb, W, emu.ptr(mlp), emu.ptr(table),
                                                     emu.ptr(vi), emu.ptr(pooled), emu.ptr(out), 1,
                                                     None))
    want_pooled = table[vi.astype(np.int64)].astype(np.float16).reshape(B, n_emb, W)
    assert (pooled.view(np.uint16) == want_pooled.view(np.uint16)).all()
    out2 = np.empty_like(out)
    emu.check(elib, elib.hctr_interaction_fwd(B, n_emb, W, emu.ptr(mlp), emu.ptr(want_pooled),
                                              emu.ptr(out2), 1, None))
    assert (out.view(np.uint16) == out2.view(np.uint16)).all()


HOT_OPTS = [("sgd", dict(optimizer=6, atomic_update=0)), ("adagrad", dict(optimizer=3)),
            ("adam", dict(optimizer=1, update_type=0))]


@pytest.mark.parametrize("name,kw", HOT_OPTS, ids=[o[0] for o in HOT_OPTS])
@pytest.mark.parametrize("B,S,D,odt,hot_rows", [(9000, 3, 8, 0, 64), (5000, 5, 16, 1, 8192),
                                                (4097, 2, 4, 0, 300)])
def test_hot_rows_of_one_hot_batches(oracle, elib, monkeypatch, name, kw, B, S, D, odt, hot_rows):
    """the update's hot-row path (hot_chunk_kernel / hot_apply_kernel + the sort's filtering first
    pass) forced on at a small size: streams of one slot each, more than one chunk per stream, runs
    that cross tiles and chunks (a 3-row table), rows on both sides of the hot bound, a batch that
    is NOT one-hot in between (both kernels exit, the sort keeps every pair).  Against the oracle
    within the re-association of the sums, the same bits on a second handle, and next to the plain
    path (HCTR_HOT_ROWS=0)."""
    from hugectr_amd import _lib
    vps = 2500
    V = S * vps
    opt = dict(lr=0.05, scaler=2.0, beta1=0.9, beta2=0.999, epsilon=1e-7, **kw)
    npd = np.float16 if odt == 1 else np.float32

    def run(rows_env):
        monkeypatch.setenv("HCTR_HOT_MIN", "0")
        monkeypatch.setenv("HCTR_HOT_ROWS", str(rows_env))
        rng = np.random.default_rng(B + D)
        emb = emu.Embedding(elib, _lib.EMB_LOCALIZED, B, V, D, 2 * S, S, 0, opt, out_dtype=odt)
        table = emb.table().copy()
        ns = {1: 2, 3: 1, 6: 0}[kw["optimizer"]]
        s0 = np.zeros_like(table) if ns >= 1 else None
        s1 = np.zeros_like(table) if ns >= 2 else None
        ht = oracle.HashTable(V, 8)
        m = {1: oracle.OPT_ADAM, 3: oracle.OPT_ADAGRAD, 6: oracle.OPT_SGD}
        for it in range(4):
            k = np.empty((B, S), dtype=np.int64)
            k[:, 0] = rng.integers(0, 3, size=B)  # three rows take a whole stream
            for s in range(1, S):
                k[:, s] = np.minimum((rng.pareto(0.9, size=B) * 3).astype(np.int64), vps - 1) + s * vps
            if it == 2:  # ragged: 0..2 keys per bucket -> the plain path on the same handle
                lens = rng.integers(0, 3, size=B * S)
                ro = np.concatenate([[0], np.cumsum(lens)]).astype(np.int64)
                slot_of = np.repeat(np.tile(np.arange(S), B), lens)
                keys = (rng.integers(0, 40, size=slot_of.size) + slot_of * vps).astype(np.int64)
            else:
                ro = np.arange(B * S + 1, dtype=np.int64)
                keys = k.reshape(-1)
            emb.forward(True, ro, keys)
            vi = ht.get_insert(keys)
            assert (emb.value_index(keys.size) == vi).all()
            g = (rng.standard_normal((B * S, D)) * 0.1).astype(npd)
            emb.backward(g.reshape(B, S, D))
            emb.update_params()
            o = oracle.OptParamsC()
            o.optimizer, o.update_type, o.lr = m[kw["optimizer"]], kw.get("update_type", 0), 0.05
            o.beta1, o.beta2, o.epsilon = 0.9, 0.999, 1e-7
            o.momentum_factor, o.scaler, o.times = 0.0, 2.0, it + 1
            o.state_half = 1 if odt == 1 else 0  # fp16 embeddings keep fp16-valued state (q6)
            oracle.update_params(ro, vi, oracle.backward(ro, g.astype(np.float32), D, 0), o, table,
                                 s0, s1, None)
            assert_close(emb.table(), table, 3e-4, 3e-5, f"{name} table it{it} hot={rows_env}")
            if s0 is not None:
                assert_close(emb.opt_state(0), s0, 3e-4, 3e-5, f"{name} state0 it{it}")
        return emb.table().copy()

    a = run(hot_rows)
    b = run(hot_rows)
    assert (a.view(np.uint32) == b.view(np.uint32)).all(), "the hot path is not deterministic"
    c = run(0)
    assert_close(a, c, 3e-4, 3e-5, "hot path vs plain path")


@pytest.mark.parametrize("dtype,npd", [(1, np.float16)])
@pytest.mark.parametrize("M,N,K,bm", [(70, 128, 128, 64), (200, 256, 64, 128), (300, 512, 128, 256)])
def test_own_gemm_and_its_epilogues(elib, monkeypatch, dtype, npd, M, N, K, bm):
    """hctr_gemm_nt16 (cross_gemm.hip) stepped through on the host: MFMA fragment layout, the
    swizzled tile image the LDS DMA fills through permuted source addresses, the staging ring
    (2 and 3 buffers), both tile heights, rows past the end, the three epilogues -- against fp64
    products of the same fp16 operands (tests/test_dense_gpu.py runs the same on the device, where
    the DMA is asynchronous)."""
    rng = np.random.default_rng(M + K)
    a = (rng.standard_normal((M, K)) * 0.5).astype(npd)
    bt = (rng.standard_normal((N, K)) * 0.5).astype(npd)
    bt[:, 0] += (np.arange(N) * 0.01).astype(npd)  # (asymmetric in n)
    bias = rng.standard_normal(N).astype(npd)
    x0 = rng.standard_normal((M, N)).astype(npd)
    xl = rng.standard_normal((M, N)).astype(npd)
    ref = a.astype(np.float64) @ bt.astype(np.float64).T
    monkeypatch.setenv("HCTR_GEMM_BM", str(bm))
    for stages in ("2", "3"):
        monkeypatch.setenv("HCTR_GEMM_STAGES", stages)
        c = np.full((M, N), np.nan, npd)
        emu.check(elib, elib.hctr_gemm_nt16(M, N, K, emu.ptr(a), K, emu.ptr(bt), K, emu.ptr(c), N, 0,
                                            None, None, None, None, dtype, None))
        tol = 2.0 ** -10 * np.maximum(np.abs(ref), 1.0)
        assert (np.abs(c.astype(np.float64) - ref) <= tol).all(), ("plain", stages)
        h = np.full((M, N), np.nan, npd)
        emu.check(elib, elib.hctr_gemm_nt16(M, N, K, emu.ptr(a), K, emu.ptr(bt), K, emu.ptr(c), N, 1,
                                            emu.ptr(bias), emu.ptr(x0), emu.ptr(xl), emu.ptr(h), dtype,
                                            None))
        hw = ref + bias.astype(np.float64)
        assert (np.abs(h.astype(np.float64) - hw) <= 2.0 ** -10 * np.maximum(np.abs(hw), 1.0)).all()
        want = xl.astype(np.float64) + x0.astype(np.float64) * h.astype(np.float64)
        assert (np.abs(c.astype(np.float64) - want) <= 2.0 ** -10 * np.maximum(np.abs(want), 1.0)).all()
        emu.check(elib, elib.hctr_gemm_nt16(M, N, K, emu.ptr(a), K, emu.ptr(bt), K, emu.ptr(c), N, 2,
                                            None, None, emu.ptr(xl), None, dtype, None))
        want = ref + xl.astype(np.float64)
        assert (np.abs(c.astype(np.float64) - want) <= 2.0 ** -10 * np.maximum(np.abs(want), 1.0)).all()


def test_hash_rows_of_batches_full_of_unseen_keys(oracle, elib):
    """a first epoch: nearly every key of a batch is unseen, the finish kernel's list is DENSE (two
    or more entries per thread: the wave-aggregated form of its mask / region atomics) -- 30 000 and
    50 000 keys with duplicates, then a batch that mixes seen and unseen; rows equal the sequential
    oracle's"""
    rng = np.random.default_rng(44)
    cap = 200000
    o = oracle.HashTable(cap, 8)
    g = emu.HashTable(elib, cap, 1)
    for n, hi in ((30000, 10**7), (50000, 10**7), (20000, 60000)):
        keys = rng.integers(0, hi, size=n).astype(np.int64)
        keys[rng.integers(0, n, size=n // 10)] = keys[rng.integers(0, n, size=n // 10)]  # duplicates
        assert (g.get_insert(keys) == o.get_insert(keys)).all(), n


def test_index_ahead_equals_the_sequential_oracle(oracle, elib):
    """hctr_emb_index_ahead / hctr_emb_index_adopt (the next batch's index stage into the spare
    buffers, finish kernel as two launches without its grid barrier): the rows of a sequence of
    batches -- first one indexed in line, the others ahead while the previous batch is still the
    current one -- equal the sequential oracle's, the current batch's rows stay untouched until the
    adoption, and an update in between takes the right batch"""
    import ctypes
    from hugectr_amd import _lib
    rng = np.random.default_rng(21)
    B, S, D, vps = 600, 5, 8, 400
    V = S * vps
    emb = emu.Embedding(elib, _lib.EMB_LOCALIZED, B, V, D, S, S, 0,
                        dict(optimizer=6, lr=0.1, scaler=1.0, atomic_update=0))
    table = emb.table().copy()
    ht = oracle.HashTable(V, 8)
    ro = np.arange(B * S + 1, dtype=np.int64)

    def batch():
        return (np.stack([np.minimum((rng.pareto(1.0, size=B) * 4).astype(np.int64), vps - 1) + s * vps
                          for s in range(S)], 1).reshape(-1)).astype(np.int64)
    keys = batch()
    emu.check(elib, elib.hctr_emb_index(emb.h, 1, emu.ptr(ro), emu.ptr(keys), keys.size, None))
    for it in range(4):
        vi = ht.get_insert(keys)
        assert (emb.value_index(keys.size) == vi).all(), f"rows of batch {it}"
        nxt = batch()
        emu.check(elib, elib.hctr_emb_index_ahead(emb.h, emu.ptr(ro), emu.ptr(nxt), nxt.size, None))
        assert (emb.value_index(keys.size) == vi).all(), "index_ahead touched the current batch"
        g = rng.standard_normal((B * S, D)).astype(np.float32)
        emb.backward(g.reshape(B, S, D))
        emb.update_params()
        o = oracle.OptParamsC()
        o.optimizer, o.update_type, o.lr, o.scaler, o.times = oracle.OPT_SGD, 0, 0.1, 1.0, it + 1
        oracle.update_params(ro, vi, g, o, table)
        assert_close(emb.table(), table, 1e-5, 1e-6, f"table after the update of batch {it}")
        emu.check(elib, elib.hctr_emb_index_adopt(emb.h))
        keys = nxt
    assert (emb.value_index(keys.size) == ht.get_insert(keys)).all()
    emb.check_overflow()


def test_no_collective_met_an_idle_lane(elib):
    """(runs last in this file) no shuffle of the kernels exercised above read a lane that was not
    taking part in it -- on the hardware such a read returns a stale register"""
    s = emu.stats(elib)
    assert s["shfl_from_inactive"] == 0, s
    if not os.environ.get("PYTEST_XDIST_WORKER"):  # (a worker of `-n N` ran only its share of the file)
        assert s["launches"] > 100, s
