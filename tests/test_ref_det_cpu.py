"""oracle/det_oracle.py (the restated oracle of the dynamic embedding table the GPU tests of
hctr_det_* compare against) against the REFERENCE'S OWN CPU mirror of that table,
embedding::DynamicEmbeddingTableCPU (R/HugeCTR/embedding_storage/dynamic_embedding_cpu.hpp:32-485)
with its optimizer formulas (R/HugeCTR/embedding_storage/optimizers.hpp:25-199), compiled from the
reference checkout into oracle/_ref/libref_det.so (oracle/Makefile `ref`; oracle/ref_shims/det/
core23/logger.hpp replaces the CUDA-bound core23 / interface headers with declarations only).

Both sides are loaded with the same keys and vectors (the reference draws unseen keys from
std::random_device, so every key is preloaded), then take the same `update(unique_keys, table_ids,
ev_start_indices, wgrad)` calls -- the reference's Wgrad pieces -- for all seven optimizers, and
are read back through `lookup`."""
import ctypes
import os

import numpy as np
import pytest

from oracle import det_oracle as do

LIB = os.path.join(os.path.dirname(__file__), "..", "oracle", "_ref", "libref_det.so")
pytestmark = pytest.mark.skipif(not os.path.exists(LIB),
                                reason="oracle/_ref not built (needs the reference checkout)")

DIMS = (4, 6, 1)
KEYS_PER_TABLE = 40
STEPS = 4


def _ref():
    L = ctypes.CDLL(LIB)
    P, F = ctypes.c_void_p, ctypes.c_float
    L.ref_det_create.restype = P
    L.ref_det_create.argtypes = [ctypes.c_int, P, ctypes.c_int] + [F] * 10
    L.ref_det_destroy.argtypes = [P]
    L.ref_det_load.argtypes = [P, P, ctypes.c_size_t, P, ctypes.c_size_t, P, P]
    L.ref_det_lookup.argtypes = [P, P, ctypes.c_size_t, P, ctypes.c_size_t, P, P]
    L.ref_det_update.argtypes = [P, P, ctypes.c_size_t, P, P, P]
    L.ref_det_size.restype = ctypes.c_size_t
    L.ref_det_size.argtypes = [P]
    return L


def _p(a):
    return a.ctypes.data_as(ctypes.c_void_p)


# (det_oracle code == the reference's Optimizer_t value, common.hpp:82-92)
CASES = [("ftrl", do.FTRL, dict(lambda1=0.05, lambda2=0.1, ftrl_beta=0.5)),
         ("ftrl_l1_0", do.FTRL, dict(lambda1=0.0, lambda2=0.0, ftrl_beta=0.0)),
         ("adam", do.ADAM, {}), ("rmsprop", do.RMSPROP, dict(rms_beta=0.8)),
         ("adagrad", do.ADAGRAD, {}), ("nesterov", do.NESTEROV, dict(momentum=0.7)),
         ("momentum", do.MOMENTUM, dict(momentum=0.3)), ("sgd", do.SGD, {})]


@pytest.mark.parametrize("name,opt,kw", CASES, ids=[c[0] for c in CASES])
@pytest.mark.parametrize("scaler", [1.0, 128.0])
def test_det_oracle_matches_the_reference_cpu_table(name, opt, kw, scaler):
    L = _ref()
    rng = np.random.default_rng(opt * 7 + int(scaler))
    lr, b1, b2, eps = 0.05, 0.9, 0.999, 1e-7
    mom, rb = kw.get("momentum", 0.9), kw.get("rms_beta", 0.9)
    l1, l2, fb = kw.get("lambda1", 0.0), kw.get("lambda2", 0.0), kw.get("ftrl_beta", 0.0)
    ev = np.array(DIMS, dtype=np.int32)
    h = L.ref_det_create(len(DIMS), _p(ev), opt, lr, scaler, b1, b2, eps, mom, rb, l1, l2, fb)
    assert h
    try:
        nt = len(DIMS)
        keys = np.concatenate([rng.choice(10 ** 6, KEYS_PER_TABLE, replace=False) + t * 10 ** 7
                               for t in range(nt)]).astype(np.int64)
        offs = (np.arange(nt + 1) * KEYS_PER_TABLE).astype(np.uint32)
        tids = np.arange(nt, dtype=np.int32)
        vec = (rng.standard_normal(int(sum(d * KEYS_PER_TABLE for d in DIMS))) * 0.3).astype(np.float32)
        assert L.ref_det_load(h, _p(keys), keys.size, _p(offs), offs.size, _p(tids), _p(vec)) == 0
        assert L.ref_det_size(h) == keys.size
        w = do.DetOracle(DIMS, 0.0)
        n_state = {do.FTRL: 2, do.ADAM: 2, do.SGD: 0}.get(opt, 1)
        st = do.DetOracle([d * max(n_state, 1) for d in DIMS], 0.0)
        w.scatter(keys, vec, list(tids), list(offs), add=False)  # missing keys are skipped:
        assert w.size_per_class() == [0] * nt                     # insert through lookup first
        w.lookup(keys, list(tids), list(offs))
        w.scatter(keys, vec, list(tids), list(offs), add=False)
        for step in range(1, STEPS + 1):
            # unique keys per table, tables ascending (the layout of Wgrad.unique_keys)
            pick = [np.sort(rng.choice(KEYS_PER_TABLE, rng.integers(1, KEYS_PER_TABLE), replace=False))
                    for _ in range(nt)]
            uk = np.concatenate([keys[t * KEYS_PER_TABLE + p] for t, p in enumerate(pick)])
            tid_per_key = np.concatenate([np.full(p.size, t, np.int32) for t, p in enumerate(pick)])
            sizes = np.array([DIMS[t] for t in tid_per_key], dtype=np.uint32)
            ev_start = np.concatenate([[0], np.cumsum(sizes)]).astype(np.uint32)
            g = (rng.standard_normal(int(ev_start[-1])) * scaler).astype(np.float32)
            assert L.ref_det_update(h, _p(uk), uk.size, _p(tid_per_key), _p(ev_start),
                                    _p(g.copy())) == 0
            uoffs = np.concatenate([[0], np.cumsum([p.size for p in pick])]).astype(np.int64)
            do.update(w, st, opt, uk, list(range(nt)), list(uoffs), list(ev_start[:-1]), g, lr,
                      scaler=scaler, beta1=b1, beta2=b2, eps=eps, momentum=mom, rms_beta=rb,
                      lambda1=l1, lambda2=l2, ftrl_beta=fb, times=step)
            got = np.empty(vec.size, np.float32)
            assert L.ref_det_lookup(h, _p(keys), keys.size, _p(offs), offs.size, _p(tids), _p(got)) == 0
            want = w.lookup(keys, list(tids), list(offs))
            # same float32 formulas; the reference's Ftrl mixes in double literals (1. - 2. * ...)
            np.testing.assert_allclose(want, got, rtol=2e-6, atol=1e-7,
                                       err_msg=f"{name} step {step}")
    finally:
        L.ref_det_destroy(h)
