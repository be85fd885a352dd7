"""The REFERENCE's device code of the dynamic embedding table's optimizer step
(R/HugeCTR/embedding_storage/optimizers.cuh:29-233, the seven *_update_grad_kernel, compiled from the
checkout into oracle/_ref/libref_det_kernels.so and executed by the host interpreter of tests/emu
with the launch shape of DynamicEmbeddingTable::update, dynamic_embedding.cu:222-317) next to

  * oracle/det_oracle.py `update` -- the oracle the GPU tests of hctr_det_update compare against
    (so far pinned to the reference's CPU mirror, optimizers.hpp, only), and
  * this repo's kernel source, hctr_det_update of hugectr_amd/csrc/det.hip, stepped through by the
    same interpreter (tests/emu/_build/libhctr_emu.so),

on the same keys, vectors, states and summed gradients, several steps, all seven optimizers:
weights and optimizer states agree BIT FOR BIT on all three sides."""
import ctypes
import os
import sys

import numpy as np
import pytest

from oracle import det_oracle as do

HERE = os.path.dirname(os.path.abspath(__file__))
LIB = os.path.join(HERE, "..", "oracle", "_ref", "libref_det_kernels.so")
pytestmark = pytest.mark.skipif(not os.path.exists(LIB),
                                reason="oracle/_ref not built (needs the reference checkout)")
sys.path.insert(0, os.path.join(HERE, "emu"))
import emu  # noqa: E402

DIMS = (4, 128, 1, 6)
KEYS = 70  # per table: the unique keys of a step are a subset
f32 = np.float32

CASES = [("ftrl", do.FTRL, dict(lambda1=0.05, lambda2=0.1, ftrl_beta=0.5)),
         ("ftrl_l1_0", do.FTRL, dict(lambda1=0.0, lambda2=0.0, ftrl_beta=0.0)),
         ("adam", do.ADAM, {}), ("rmsprop", do.RMSPROP, dict(rms_beta=0.8)),
         ("adagrad", do.ADAGRAD, {}), ("nesterov", do.NESTEROV, dict(momentum=0.7)),
         ("momentum", do.MOMENTUM, dict(momentum=0.3)), ("sgd", do.SGD, {})]


def _p(a):
    return a.ctypes.data_as(ctypes.c_void_p)


def _ref():
    L = ctypes.CDLL(LIB)
    F, P = ctypes.c_float, ctypes.c_void_p
    L.refdetk_update.argtypes = [ctypes.c_int, ctypes.c_uint32, P, P, P, F, F, F, F, F, P]
    return L


@pytest.fixture(scope="module")
def elib():
    return emu.load_under_test()


@pytest.mark.parametrize("name,opt,kw", CASES, ids=[c[0] for c in CASES])
@pytest.mark.parametrize("scaler", [1.0, 128.0])
def test_reference_device_optimizer_kernels_next_to_oracle_and_hip_source(elib, name, opt, kw, scaler):
    from hugectr_amd import _lib
    L = _ref()
    rng = np.random.default_rng(opt * 11 + int(scaler))
    lr, b1, b2, eps = 0.05, 0.9, 0.999, 1e-7
    mom, rb = kw.get("momentum", 0.9), kw.get("rms_beta", 0.9)
    l1, l2, fb = kw.get("lambda1", 0.0), kw.get("lambda2", 0.0), kw.get("ftrl_beta", 0.0)
    nt = len(DIMS)
    n_state = {do.FTRL: 2, do.ADAM: 2, do.SGD: 0}.get(opt, 1)
    keys = np.concatenate([rng.choice(10 ** 6, KEYS, replace=False) + t * 10 ** 7
                           for t in range(nt)]).astype(np.int64)
    offs = (np.arange(nt + 1) * KEYS).astype(np.uint64)
    tids = np.arange(nt, dtype=np.uint64)
    vec = (rng.standard_normal(int(sum(d * KEYS for d in DIMS))) * 0.3).astype(f32)
    # ---- the three sides, loaded alike ----------------------------------------------------------
    # (1) reference device kernels: plain arrays per key
    w_ref, s_ref, pos = {}, {}, 0
    for t in range(nt):
        for k in keys[t * KEYS:(t + 1) * KEYS]:
            w_ref[int(k)] = vec[pos:pos + DIMS[t]].copy()
            s_ref[int(k)] = np.zeros(DIMS[t] * max(n_state, 1), dtype=f32)
            pos += DIMS[t]
    # (2) the oracle
    w_orc = do.DetOracle(DIMS, 0.0)
    s_orc = do.DetOracle([d * max(n_state, 1) for d in DIMS], 0.0)
    w_orc.lookup(keys, list(tids), list(offs))
    w_orc.scatter(keys, vec, list(tids), list(offs), add=False)
    # (3) the HIP source under the interpreter
    lib = elib
    emu.bind(lib)
    dims = (ctypes.c_size_t * nt)(*DIMS)
    sdims = (ctypes.c_size_t * nt)(*[d * max(n_state, 1) for d in DIMS])
    hw, hs = ctypes.c_void_p(), ctypes.c_void_p()
    emu.check(lib, lib.hctr_det_create(nt, dims, b"zeros", 64, _lib.KEY_I64, 1, ctypes.byref(hw)))
    emu.check(lib, lib.hctr_det_create(nt, sdims, b"zeros", 64, _lib.KEY_I64, 1, ctypes.byref(hs)))
    ids = (ctypes.c_size_t * nt)(*range(nt))
    ofs = (ctypes.c_size_t * (nt + 1))(*[int(v) for v in offs])
    tmp = np.empty_like(vec)
    emu.check(lib, lib.hctr_det_lookup(hw, _p(keys), _p(tmp), keys.size, ids, ofs, nt, None))
    emu.check(lib, lib.hctr_det_scatter_update(hw, _p(keys), _p(vec), keys.size, ids, ofs, nt, None))
    try:
        for step in range(1, 5):
            pick = [np.sort(rng.choice(KEYS, rng.integers(1, KEYS), replace=False)) for _ in range(nt)]
            uk = np.concatenate([keys[t * KEYS + p] for t, p in enumerate(pick)])
            tid_per_key = np.concatenate([np.full(p.size, t, np.int32) for t, p in enumerate(pick)])
            sizes = np.array([DIMS[t] for t in tid_per_key], dtype=np.uint32)
            ev_start = np.concatenate([[0], np.cumsum(sizes)]).astype(np.uint32)
            g = (rng.standard_normal(int(ev_start[-1])) * scaler).astype(f32)
            uoffs = np.concatenate([[0], np.cumsum([p.size for p in pick])]).astype(np.int64)
            # (1) reference kernels: wgrad -> delta in place, states through per-key pointers
            g_ref = g.copy()
            sp = (ctypes.c_void_p * uk.size)(*[s_ref[int(k)].ctypes.data for k in uk])
            wp = (ctypes.c_void_p * uk.size)(*[w_ref[int(k)].ctypes.data for k in uk])
            bias = f32(np.sqrt(1.0 - float(f32(b2)) ** step) / (1.0 - float(f32(b1)) ** step))
            if opt == do.ADAM:
                args = (f32(lr) * bias, b1, b2, eps)
            elif opt == do.FTRL:
                args = (lr, l1, f32(l2) + f32(fb) / f32(lr), 0.0)
            elif opt == do.RMSPROP:
                args = (lr, rb, eps, 0.0)
            elif opt == do.ADAGRAD:
                args = (lr, eps, 0.0, 0.0)
            else:
                args = (lr, mom, 0.0, 0.0)
            assert L.refdetk_update(opt, uk.size, _p(ev_start), sp, wp, *[float(a) for a in args],
                                    scaler, _p(g_ref)) == 0
            for j, k in enumerate(uk):  # table->scatter_add(unique_keys, wgrad)
                a, b = int(ev_start[j]), int(ev_start[j + 1])
                w_ref[int(k)] = (w_ref[int(k)] + g_ref[a:b]).astype(f32)
            # (2) the oracle
            do.update(w_orc, s_orc, opt, uk, list(range(nt)), list(uoffs), list(ev_start[:-1]), g, lr,
                      scaler=scaler, beta1=b1, beta2=b2, eps=eps, momentum=mom, rms_beta=rb,
                      lambda1=l1, lambda2=l2, ftrl_beta=fb, times=step)
            # (3) the HIP source
            p = _lib.DetOptParams()
            p.optimizer = {do.FTRL: _lib.OPT_FTRL, do.ADAM: _lib.OPT_ADAM, do.RMSPROP: _lib.OPT_RMSPROP,
                           do.ADAGRAD: _lib.OPT_ADAGRAD, do.NESTEROV: _lib.OPT_NESTEROV,
                           do.MOMENTUM: _lib.OPT_MOMENTUM_SGD, do.SGD: _lib.OPT_SGD}[opt]
            p.lr, p.beta1, p.beta2, p.epsilon = lr, b1, b2, eps
            p.momentum_factor, p.rmsprop_beta = mom, rb
            p.ftrl_lambda1, p.ftrl_lambda2, p.ftrl_beta, p.scaler = l1, l2, fb, scaler
            uofs = (ctypes.c_size_t * (nt + 1))(*[int(v) for v in uoffs])
            emu.check(lib, lib.hctr_det_update(hw, hs if n_state else None, ctypes.byref(p), _p(uk),
                                               uk.size, ids, uofs, nt, _p(ev_start), _p(g), None))
            # ---- compare -----------------------------------------------------------------------
            want = np.concatenate([w_ref[int(k)] for k in keys])
            got_orc = w_orc.lookup(keys, list(tids), list(offs))
            got_hip = np.empty_like(vec)
            emu.check(lib, lib.hctr_det_lookup(hw, _p(keys), _p(got_hip), keys.size, ids, ofs, nt, None))
            np.testing.assert_array_equal(got_orc, want,
                                          err_msg=f"{name} step {step}: oracle vs reference device code")
            np.testing.assert_array_equal(got_hip, want,
                                          err_msg=f"{name} step {step}: HIP source vs reference device code")
            if n_state:
                st_ref = np.concatenate([s_ref[int(k)] for k in keys])
                st_orc = s_orc.lookup(keys, list(tids), list(offs))
                np.testing.assert_array_equal(st_orc, st_ref,
                                              err_msg=f"{name} step {step}: oracle state")
    finally:
        lib.hctr_det_destroy(hw)
        lib.hctr_det_destroy(hs)


FLAT_CASES = [("adagrad", do.ADAGRAD, {}), ("adam", do.ADAM, {}),
              ("momentum", do.MOMENTUM, dict(momentum=0.3)), ("sgd", do.SGD, {})]


@pytest.mark.parametrize("name,opt,kw", FLAT_CASES, ids=[c[0] for c in FLAT_CASES])
@pytest.mark.parametrize("D,scaler", [(32, 1.0), (128, 64.0), (8, 1.0)])
def test_reference_device_optimizer_kernels_next_to_the_flat_row_store_step(elib, name, opt, kw, D, scaler):
    """The step the embedding_collection takes on dynamic tables of one vector size for SGD /
    AdaGrad / Adam / MomentumSGD: row numbers from hctr_det_lookup_rows, weights in the flat row
    store, state at the same row numbers (hctr_det_state_store), the static tables' sparse update
    (hctr_updater_update) on the buckets' gradients -- no unique-key list, no wgrad buffer, no state
    table.  Next to the reference's device kernels (optimizers.cuh) fed with the per-key gradient
    sums in ascending bucket order (what LocalReduce hands them) and their scatter_add:
    weights AND states bit for bit, keys met 0 ... 3 times per step, the classes growing on the way."""
    from hugectr_amd import _lib
    L = _ref()
    lib = elib
    emu.bind(lib)
    rng = np.random.default_rng(opt * 13 + D)
    lr, b1, b2, eps = 0.05, 0.9, 0.999, 1e-7
    mom = kw.get("momentum", 0.9)
    nt = 3
    n_state = {do.ADAM: 2, do.SGD: 0}.get(opt, 1)
    code = {do.ADAM: _lib.OPT_ADAM, do.ADAGRAD: _lib.OPT_ADAGRAD, do.MOMENTUM: _lib.OPT_MOMENTUM_SGD,
            do.SGD: _lib.OPT_SGD}[opt]
    pool = [rng.choice(10 ** 6, KEYS, replace=False).astype(np.int64) + t * 10 ** 7 for t in range(nt)]
    dims = (ctypes.c_size_t * nt)(*([D] * nt))
    hw, upd = ctypes.c_void_p(), ctypes.c_void_p()
    emu.check(lib, lib.hctr_det_create(nt, dims, b"", 8, _lib.KEY_I64, 7, ctypes.byref(hw)))
    emu.check(lib, lib.hctr_updater_create(4 * nt * KEYS, 0xFFFFFFEF, D, ctypes.byref(upd)))
    ids = (ctypes.c_size_t * nt)(*range(nt))
    w_ref, s_ref = {}, {}
    times = 0
    try:
        for step in range(1, 6):
            # the batch: per class a multiset of keys (0 ... 3 copies each), one key per bucket
            per = []
            for t in range(nt):
                m = rng.integers(0, 4, size=KEYS) if step != 3 else np.zeros(KEYS, np.int64)
                ks = np.repeat(pool[t], m)
                per.append(ks[rng.permutation(ks.size)])
            keys = np.concatenate(per).astype(np.int64)
            n = keys.size
            ofs = (ctypes.c_size_t * (nt + 1))(*np.concatenate([[0], np.cumsum([p.size for p in per])]).tolist())
            rows = np.zeros(max(n, 1), np.uint64)
            base = (ctypes.c_uint64 * (nt + 1))()
            emu.check(lib, lib.hctr_det_lookup_rows(hw, _p(keys), n, ids, ofs, nt, 1, None, _p(rows),
                                                    base, None))
            store, total = ctypes.c_void_p(), ctypes.c_uint64()
            emu.check(lib, lib.hctr_det_row_store(hw, ctypes.byref(store), ctypes.byref(total)))
            table = np.ctypeslib.as_array(ctypes.cast(store, ctypes.POINTER(ctypes.c_float)),
                                          shape=(total.value, D))
            for k, r in zip(keys, rows[:n]):  # rows created by this lookup: the reference's side starts from them
                if int(k) not in w_ref:
                    w_ref[int(k)] = table[int(r)].copy()
                    s_ref[int(k)] = np.zeros(D * max(n_state, 1), dtype=f32)
            if n == 0:
                continue  # (no keys: no step, `times` stays -- dynamic_embedding.cu:187)
            times += 1
            g = (rng.standard_normal((n, D)) * scaler).astype(f32)
            # (1) reference: unique keys, their gradient sums in ascending bucket order
            uk, first = np.unique(keys, return_index=True)
            wg = np.zeros((uk.size, D), f32)
            slot = {int(k): j for j, k in enumerate(uk)}
            seen = set()
            for i, k in enumerate(keys):
                j = slot[int(k)]
                wg[j] = g[i] if int(k) not in seen else (wg[j] + g[i]).astype(f32)
                seen.add(int(k))
            ev_start = (np.arange(uk.size + 1) * D).astype(np.uint32)
            g_ref = wg.reshape(-1).copy()
            sp = (ctypes.c_void_p * uk.size)(*[s_ref[int(k)].ctypes.data for k in uk])
            wp = (ctypes.c_void_p * uk.size)(*[w_ref[int(k)].ctypes.data for k in uk])
            bias = f32(np.sqrt(1.0 - float(f32(b2)) ** times) / (1.0 - float(f32(b1)) ** times))
            args = {do.ADAM: (f32(lr) * bias, b1, b2, eps), do.ADAGRAD: (lr, eps, 0.0, 0.0)}.get(
                opt, (lr, mom, 0.0, 0.0))
            assert L.refdetk_update(opt, uk.size, _p(ev_start), sp, wp, *[float(a) for a in args],
                                    scaler, _p(g_ref)) == 0
            for j, k in enumerate(uk):
                w_ref[int(k)] = (w_ref[int(k)] + g_ref[j * D:(j + 1) * D]).astype(f32)
            # (2) the flat row store's step
            s0, s1 = ctypes.c_void_p(), ctypes.c_void_p()
            if n_state:
                emu.check(lib, lib.hctr_det_state_store(hw, n_state, ctypes.byref(s0), ctypes.byref(s1), None))
            br = np.arange(n + 1, dtype=np.int64)
            emu.check(lib, lib.hctr_updater_set_row_bound(upd, total.value))
            emu.check(lib, lib.hctr_updater_update(upd, n, n, _p(br), _p(rows), _p(g), _lib.F32, code,
                                                   _lib.UPDATE_LOCAL, lr, b1, b2, eps, mom, scaler,
                                                   times, store, s0, s1, None))
            # ---- compare: every key ever met (absent ones must not have moved) ---------------------
            allk = np.array(sorted(w_ref), np.int64)
            cls = (allk // 10 ** 7).astype(np.int64)
            order = np.argsort(cls, kind="stable")
            allk, cls = allk[order], cls[order]
            aofs = (ctypes.c_size_t * (nt + 1))(*np.concatenate([[0], np.cumsum(np.bincount(cls, minlength=nt))]).tolist())
            arow = np.zeros(allk.size, np.uint64)
            emu.check(lib, lib.hctr_det_lookup_rows(hw, _p(allk), allk.size, ids, aofs, nt, 0, None,
                                                    _p(arow), base, None))
            emu.check(lib, lib.hctr_det_row_store(hw, ctypes.byref(store), ctypes.byref(total)))
            table = np.ctypeslib.as_array(ctypes.cast(store, ctypes.POINTER(ctypes.c_float)),
                                          shape=(total.value, D))
            np.testing.assert_array_equal(table[arow.astype(np.int64)],
                                          np.stack([w_ref[int(k)] for k in allk]),
                                          err_msg=f"{name} step {step}: weights")
            for j, sp_ in enumerate((s0, s1)[:n_state]):
                st = np.ctypeslib.as_array(ctypes.cast(sp_, ctypes.POINTER(ctypes.c_float)),
                                           shape=(total.value, D))
                np.testing.assert_array_equal(st[arow.astype(np.int64)],
                                              np.stack([s_ref[int(k)][j * D:(j + 1) * D] for k in allk]),
                                              err_msg=f"{name} step {step}: state {j}")
        caps = (ctypes.c_size_t * nt)()
        emu.check(lib, lib.hctr_det_capacity_per_class(hw, caps))
        assert min(caps) > 8, "the classes were meant to grow"
    finally:
        lib.hctr_updater_destroy(upd)
        lib.hctr_det_destroy(hw)
