"""The CPU oracle (oracle/hctr_oracle.c) against the REFERENCE'S OWN CPU oracle of the legacy sparse
embedding, SparseEmbeddingHashCpu (R/test/utest/embedding/sparse_embedding_hash_cpu.hpp:52-1015),
compiled from the reference checkout into oracle/_ref/libref_embedding.so (oracle/Makefile `ref`;
oracle/ref_shims/common.hpp replaces the CUDA-bound common.hpp with declarations only).

Both sides are driven the way the reference tests drive theirs
(localized_slot_sparse_embedding_hash_test.cu:181-519): a Norm dataset file and a sparse model
(key + emb_vector files) on disk; per step read a batch, forward, backward with the forward output
as top gradient, update.  The reference side reads the files itself, so the Norm writer, the
CheckSum framing and the model file layout of hugectr_amd are pinned on the way."""
import ctypes
import os

import numpy as np
import pytest

from hugectr_amd.data import write_norm
from oracle import pyoracle as po

LIB = os.path.join(os.path.dirname(__file__), "..", "oracle", "_ref", "libref_embedding.so")
pytestmark = pytest.mark.skipif(not os.path.exists(LIB),
                                reason="oracle/_ref not built (needs the reference checkout)")

REF_OPT = {"adam": 1, "adagrad": 3, "nesterov": 4, "momentum": 5, "sgd": 6}  # common.hpp:82-92
MY_OPT = {"adam": 0, "adagrad": 1, "momentum": 2, "nesterov": 3, "sgd": 4}
B, S, D, V, STEPS = 32, 4, 8, 200, 3
HOT = (1, 2, 3, 1)


def _ref():
    L = ctypes.CDLL(LIB)
    L.ref_emb_create.restype = ctypes.c_void_p
    L.ref_emb_create.argtypes = ([ctypes.c_int] * 9 + [ctypes.c_longlong] + [ctypes.c_int] * 3 +
                                 [ctypes.c_float] * 6 + [ctypes.c_char_p] * 2)
    L.ref_emb_step.argtypes = [ctypes.c_void_p, ctypes.c_int, ctypes.c_int, ctypes.c_void_p,
                               ctypes.c_void_p]
    L.ref_emb_table.argtypes = [ctypes.c_void_p, ctypes.c_int, ctypes.c_void_p, ctypes.c_void_p]
    L.ref_emb_destroy.argtypes = [ctypes.c_void_p, ctypes.c_int]
    return L


@pytest.fixture(scope="module")
def dataset(tmp_path_factory):
    d = tmp_path_factory.mktemp("ref_emb")
    rng = np.random.default_rng(0)
    n = B * STEPS
    label = rng.random((n, 1), dtype=np.float32)
    dense = rng.random((n, 2), dtype=np.float32)
    cats = [rng.integers(0, V, size=(n, h)).astype(np.int64) for h in HOT]
    for chk in (False, True):
        write_norm(str(d / f"data{int(chk)}.bin"), label, dense, cats, i64_key=True, check_sum=chk)
        (d / f"list{int(chk)}.txt").write_text(f"1\n{d}/data{int(chk)}.bin\n")
    os.makedirs(d / "model")
    mkeys = rng.permutation(V).astype(np.int64)
    table0 = rng.uniform(-0.1, 0.1, (V, D)).astype(np.float32)
    mkeys.tofile(d / "model" / "key")
    table0.tofile(d / "model" / "emb_vector")
    return d, cats, mkeys, table0


def _batch(cats, st):
    keys = np.concatenate([np.concatenate([c[i] for c in cats]) for i in range(st * B, (st + 1) * B)])
    ro = np.concatenate([[0], np.cumsum(np.tile(HOT, B))]).astype(np.int64)
    return ro, keys


CASES = [("sgd", 0), ("adam", 0), ("adam", 1), ("adam", 2), ("adagrad", 0), ("momentum", 0),
         ("momentum", 1), ("nesterov", 0), ("nesterov", 1)]


@pytest.mark.parametrize("opt,upd", CASES)
@pytest.mark.parametrize("comb", [0, 1])
@pytest.mark.parametrize("chk", [0, 1])
def test_fp32_steps_match_the_reference_cpu_oracle(dataset, opt, upd, comb, chk):
    d, cats, mkeys, table0 = dataset
    if chk and (opt, upd) not in (("sgd", 0), ("adam", 2)):
        pytest.skip("CheckSum framing is covered with two optimizers")
    L = _ref()
    lr, scaler, b1, b2, eps, mom = 0.05, 2.0, 0.9, 0.999, 1e-7, 0.3
    h = L.ref_emb_create(0, B, sum(HOT), V, D, S, 1, 2, chk, B * STEPS, comb, REF_OPT[opt], upd, lr,
                         scaler, b1, b2, eps, mom, str(d / f"list{chk}.txt").encode(),
                         str(d / "model").encode())
    assert h
    ht = po.HashTable(V, 8)
    ht.get_insert(mkeys)  # rows in model-file order, as the reference assigns them
    table = table0.copy()
    s0, s1 = np.zeros_like(table), np.zeros_like(table)
    pt = np.ones(V * D, dtype=np.uint64)
    try:
        for st in range(STEPS):
            fwd = np.empty((B * S, D), np.float32)
            wg = np.empty((B * S, D), np.float32)
            assert L.ref_emb_step(h, 0, 1, fwd.ctypes.data, wg.ctypes.data) == 0
            ro, keys = _batch(cats, st)
            vi = ht.get_mark(keys)
            f = po.forward(ro, vi, table, D, comb)
            w = po.backward(ro, f, D, comb)
            o = po.OptParamsC(MY_OPT[opt], upd, lr, b1, b2, eps, mom, scaler, st + 1, 0)
            po.update_params(ro, vi, w, o, table, s0, s1, pt)
            rk = np.empty(V, np.int64)
            rv = np.empty((V, D), np.float32)
            L.ref_emb_table(h, 0, rk.ctypes.data, rv.ctypes.data)
            assert (rk == mkeys).all()
            if comb == 0 and st == 0:  # sum on equal tables: same additions in the same order
                assert np.array_equal(f, fwd) and np.array_equal(w, wg)
            else:  # mean: the reference's host code divides, the kernels multiply by 1/n;
                   # later steps: the tables may differ in the last bit (next assertion)
                assert np.allclose(f, fwd, rtol=1e-6, atol=1e-7)
                assert np.allclose(w, wg, rtol=1e-6, atol=1e-7)
            assert np.allclose(table, rv, rtol=0, atol=5e-7)
    finally:
        L.ref_emb_destroy(h, 0)


@pytest.mark.parametrize("opt,upd", [("sgd", 0), ("adam", 0), ("adam", 2), ("adagrad", 0),
                                     ("momentum", 1)])
@pytest.mark.parametrize("comb", [0, 1])
def test_fp16_steps_match_the_reference_cpu_oracle(dataset, opt, upd, comb):
    """TypeEmbeddingComp = __half: 16-bit outputs / gradients / optimizer state (quirk q6)"""
    d, cats, mkeys, table0 = dataset
    L = _ref()
    lr, scaler, b1, b2, eps, mom = 0.05, 8.0, 0.9, 0.999, 1e-4, 0.3
    h = L.ref_emb_create(1, B, sum(HOT), V, D, S, 1, 2, 0, B * STEPS, comb, REF_OPT[opt], upd, lr,
                         scaler, b1, b2, eps, mom, str(d / "list0.txt").encode(),
                         str(d / "model").encode())
    assert h
    ht = po.HashTable(V, 8)
    ht.get_insert(mkeys)
    table = table0.copy()
    s0, s1 = np.zeros_like(table), np.zeros_like(table)
    pt = np.ones(V * D, dtype=np.uint64)
    try:
        for st in range(STEPS):
            fwd = np.empty((B * S, D), np.float32)
            wg = np.empty((B * S, D), np.float32)
            assert L.ref_emb_step(h, 1, 1, fwd.ctypes.data, wg.ctypes.data) == 0
            ro, keys = _batch(cats, st)
            vi = ht.get_mark(keys)
            f = po.forward_mixed(ro, vi, table, D, comb, "f16")
            w = po.backward_mixed(ro, f, D, comb, "f16")
            o = po.OptParamsC(MY_OPT[opt], upd, lr, b1, b2, eps, mom, scaler, st + 1, 1)
            po.update_params(ro, vi, w, o, table, s0, s1, pt)
            rk = np.empty(V, np.int64)
            rv = np.empty((V, D), np.float32)
            L.ref_emb_table(h, 1, rk.ctypes.data, rv.ctypes.data)
            # sum: equal up to one fp16 rounding of later steps' tables.  mean: the oracle follows
            # the reference KERNELS (x 1/n rounded to fp16, in half precision: quirk q4), the
            # reference's host oracle divides in fp32 -- up to 2 fp16 ulps apart (the reference's
            # own fp16 tolerance between the two is 5e-3, embedding_test_utils.hpp:30-44)
            rt = 2 ** -10 if comb == 0 else 2e-3
            # from the second step on the tables carry that difference (Adam / AdaGrad normalise
            # the gradient, so an fp16 ulp in it moves a weight by up to ~ lr * 2e-3 per step)
            at = 1e-7 if st == 0 or comb == 0 else 3e-4
            assert np.allclose(f, fwd, rtol=rt, atol=at)
            assert np.allclose(w, wg, rtol=rt, atol=at)
            assert np.allclose(table, rv, rtol=0, atol=(2e-4 if comb == 0 else 8e-3) * lr + 1e-6)
    finally:
        L.ref_emb_destroy(h, 1)


def test_eval_forward_reads_on(dataset):
    """forward without update (the reference's eval pass): consecutive batches of the file"""
    d, cats, mkeys, table0 = dataset
    L = _ref()
    h = L.ref_emb_create(0, B, sum(HOT), V, D, S, 1, 2, 0, B * STEPS, 0, REF_OPT["sgd"], 0, 0.1, 1.0,
                         0.9, 0.999, 1e-7, 0.0, str(d / "list0.txt").encode(),
                         str(d / "model").encode())
    assert h
    ht = po.HashTable(V, 8)
    ht.get_insert(mkeys)
    try:
        for st in range(STEPS):
            fwd = np.empty((B * S, D), np.float32)
            assert L.ref_emb_step(h, 0, 0, fwd.ctypes.data, None) == 0
            ro, keys = _batch(cats, st)
            assert np.array_equal(po.forward(ro, ht.get_mark(keys), table0, D, 0), fwd)
    finally:
        L.ref_emb_destroy(h, 0)


@pytest.mark.parametrize("comb", [0, 1])
def test_u32_keys_match_the_reference_cpu_oracle(tmp_path, comb):
    """TypeHashKey = unsigned int: 4-byte keys in the Norm file and in the hash (the model file
    keeps 8-byte keys, which the reference narrows on load)"""
    L = _ref()
    rng = np.random.default_rng(5)
    n = B * 2
    label = rng.random((n, 1), dtype=np.float32)
    dense = rng.random((n, 2), dtype=np.float32)
    cats = [rng.integers(0, V, size=(n, h)).astype(np.int64) for h in HOT]
    write_norm(str(tmp_path / "d.bin"), label, dense, cats, i64_key=False, check_sum=True)
    (tmp_path / "list.txt").write_text(f"1\n{tmp_path}/d.bin\n")
    os.makedirs(tmp_path / "model")
    mkeys = rng.permutation(V).astype(np.int64)
    table0 = rng.uniform(-0.1, 0.1, (V, D)).astype(np.float32)
    mkeys.tofile(tmp_path / "model" / "key")
    table0.tofile(tmp_path / "model" / "emb_vector")
    lr, scaler = 0.1, 1.0
    h = L.ref_emb_create(2, B, sum(HOT), V, D, S, 1, 2, 1, n, comb, REF_OPT["adagrad"], 0, lr, scaler,
                         0.9, 0.999, 1e-6, 0.0, str(tmp_path / "list.txt").encode(),
                         str(tmp_path / "model").encode())
    assert h
    ht = po.HashTable(V, 4)
    ht.get_insert(mkeys.astype(np.uint32))
    table = table0.copy()
    s0, s1 = np.zeros_like(table), np.zeros_like(table)
    pt = np.ones(V * D, dtype=np.uint64)
    try:
        for st in range(2):
            fwd = np.empty((B * S, D), np.float32)
            wg = np.empty((B * S, D), np.float32)
            assert L.ref_emb_step(h, 2, 1, fwd.ctypes.data, wg.ctypes.data) == 0
            ro, keys = _batch(cats, st)
            vi = ht.get_mark(keys.astype(np.uint32))
            f = po.forward(ro, vi, table, D, comb)
            w = po.backward(ro, f, D, comb)
            o = po.OptParamsC(MY_OPT["adagrad"], 0, lr, 0.9, 0.999, 1e-6, 0.0, scaler, st + 1, 0)
            po.update_params(ro, vi, w, o, table, s0, s1, pt)
            rk = np.empty(V, np.int64)
            rv = np.empty((V, D), np.float32)
            L.ref_emb_table(h, 2, rk.ctypes.data, rv.ctypes.data)
            assert (rk == mkeys).all()
            assert np.allclose(f, fwd, rtol=1e-6, atol=1e-7) and np.allclose(w, wg, rtol=1e-6, atol=1e-7)
            assert np.allclose(table, rv, rtol=0, atol=5e-7)
    finally:
        L.ref_emb_destroy(h, 2)
