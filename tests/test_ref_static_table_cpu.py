"""The oracle's embedding_collection optimizer steps AND the HIP updater's source against the
REFERENCE'S DEVICE CODE of the static embedding_collection table,
embedding::RaggedStaticEmbeddingTable (R/HugeCTR/embedding_storage/ragged_static_embedding.cu:
29-355: lookup kernel, key -> row functor, SGDOptimizer / AdaGradOptimizer / FtrlOptimizer,
update_kernel and the 4-wide update4_kernel), cut out of the checkout and executed by the host
interpreter of tests/emu (oracle/_ref/libref_static_table.so, oracle/Makefile `ref`).  The
reference's CPU test table (embedding_table_cpu.hpp, tests/test_ref_ebc_cpu.py) only has SGD;
AdaGrad and Ftrl of the static tables are pinned here, over several steps on the same keys so that
the accumulators matter, for vector sizes that take the vectorized kernel (ev % 4 == 0) and the
scalar one."""
import ctypes
import os
import sys

import numpy as np
import pytest

from util import assert_close

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "emu"))
import emu  # noqa: E402

LIB = os.path.join(os.path.dirname(__file__), "..", "oracle", "_ref", "libref_static_table.so")
pytestmark = pytest.mark.skipif(not os.path.exists(LIB),
                                reason="oracle/_ref not built (needs the reference checkout)")


def _p(a):
    return a.ctypes.data_as(ctypes.c_void_p) if a is not None else None


class RefStatic:
    def __init__(self, rows_per_table, ev):
        L = self.L = ctypes.CDLL(LIB)
        P, Z, I, F = ctypes.c_void_p, ctypes.c_size_t, ctypes.c_int, ctypes.c_float
        L.refstatic_update.argtypes = [I, Z, P, P, P, P, I, P, P, P, P, P, P, P, F, F, F, F, F, F]
        L.refstatic_lookup.argtypes = [Z, P, P, Z, P, I, P, P, P, P, P, P]
        T = len(rows_per_table)
        self.T, self.ev = T, ev
        self.table_ids = np.arange(T, dtype=np.int32)
        self.ev_size = np.full(T, ev, np.int32)
        self.key_offset = np.concatenate([[0], np.cumsum(rows_per_table)]).astype(np.uint64)
        self.ev_offset = (self.key_offset * ev).astype(np.uint64)

    def update(self, optimizer, keys, key_table, wgrad, table, s0, s1, lr, scaler, eps, ftrl):
        n = keys.size
        kt = np.ascontiguousarray(key_table, np.int32)
        # (the table's keys arrive with the table's key offset added, :30-31; the functor takes it off)
        k = np.ascontiguousarray(keys + self.key_offset[kt].astype(np.int64), np.int64)
        w = np.ascontiguousarray(wgrad, np.float32)
        ws = (np.arange(n, dtype=np.uint32) * self.ev).astype(np.uint32)
        self.L.refstatic_update(optimizer, n, _p(k), _p(kt), _p(w), _p(ws), self.T,
                                _p(self.table_ids), _p(self.ev_size), _p(self.key_offset),
                                _p(self.ev_offset), _p(table), _p(s0), _p(s1), lr, scaler, eps,
                                ftrl[0], ftrl[1], ftrl[2])

    def lookup(self, keys, key_table_sorted_offsets, table):
        n = keys.size
        off = np.ascontiguousarray(key_table_sorted_offsets, np.uint64)
        tab = np.repeat(np.arange(self.T), np.diff(off).astype(np.int64))
        k = np.ascontiguousarray(keys + self.key_offset[tab].astype(np.int64), np.int64)
        spaces = np.arange(self.T, dtype=np.int32)
        out = np.zeros(n, np.uint64)
        self.L.refstatic_lookup(n, _p(k), _p(off), off.size, _p(spaces), self.T, _p(self.table_ids),
                                _p(self.ev_size), _p(self.key_offset), _p(self.ev_offset), _p(table),
                                _p(out))
        return out


FTRL = (0.02, 0.05, 0.3)  # lambda1, lambda2, beta


@pytest.mark.parametrize("opt,name", [(0, "sgd"), (1, "adagrad"), (2, "ftrl")])
@pytest.mark.parametrize("ev", [16, 6, 128, 1])
def test_static_table_optimizers_equal_the_reference_device_code(oracle, opt, name, ev):
    """same unique keys, same per-key gradient sums, four steps: the oracle's optimizer step
    (hco_ebc_backward_update on one-hot lookups of distinct keys, so that a key's gradient is its
    bucket's) and, with it, hctr_updater_update (HIP source) against the reference's functors"""
    from hugectr_amd import _lib
    rng = np.random.default_rng(ev * 10 + opt)
    rows = [7, 30, 3, 18]
    T, B = len(rows), 3
    ref = RefStatic(rows, ev)
    total = int(sum(rows))
    t_ref = rng.standard_normal((total, ev)).astype(np.float32)
    t_orc, t_hip = t_ref.copy(), t_ref.copy()
    mk = lambda: np.zeros((total, ev), np.float32)  # noqa: E731
    r0, r1, o0, o1, h0, h1 = mk(), mk(), mk(), mk(), mk(), mk()
    lib = None
    if emu.available():
        lib = emu.load_under_test()
        upd = ctypes.c_void_p()
        emu.check(lib, lib.hctr_updater_create(T * B, total, ev, ctypes.byref(upd)))
        emu.check(lib, lib.hctr_updater_set_ftrl(upd, FTRL[0], FTRL[1], FTRL[2]))
    row_start = np.concatenate([[0], np.cumsum(rows)])[:-1].astype(np.int64)
    lr, scaler, eps = 0.1, 4.0, 1e-7
    for it in range(4):
        # one lookup per table, B samples, one key per bucket, keys of a lookup distinct
        keys = np.concatenate([rng.choice(r, size=B, replace=False) for r in rows]).astype(np.int64)
        key_table = np.repeat(np.arange(T), B).astype(np.int32)
        g = rng.standard_normal((T * B, ev)).astype(np.float32)  # bucket = lookup * B + b
        ref.update(opt, keys, key_table, g, t_ref, r0, r1, lr, scaler, eps, FTRL)
        br = np.arange(T * B + 1, dtype=np.int64)
        oracle.ebc_backward_update(B, np.arange(T), ev, np.zeros(T, np.int32), keys, br, row_start,
                                   t_orc, g.reshape(1, -1), optimizer=opt, lr=lr, scaler=scaler,
                                   epsilon=eps, accum=o0, ftrl=FTRL, ftrl_z=o1)
        assert_close(t_orc, t_ref, 1e-6, 1e-7, f"oracle {name} table it{it}")
        assert_close(o0, r0, 1e-6, 1e-7, f"oracle {name} accum / n it{it}")
        assert_close(o1, r1, 1e-6, 1e-7, f"oracle {name} z it{it}")
        if lib is not None:
            idx = (row_start[key_table] + keys).astype(np.uint64)
            code = {0: _lib.OPT_SGD, 1: _lib.OPT_ADAGRAD, 2: _lib.OPT_FTRL}[opt]
            emu.check(lib, lib.hctr_updater_update(upd, T * B, T * B, _p(br), _p(idx), _p(g), _lib.F32,
                                                   code, _lib.UPDATE_LOCAL, lr, 0.9, 0.999, eps, 0.0,
                                                   scaler, it + 1, _p(t_hip), _p(h0), _p(h1), None))
            assert_close(t_hip, t_ref, 2e-6, 2e-7, f"hip {name} table it{it}")
            assert_close(h0, r0, 2e-6, 2e-7, f"hip {name} accum / n it{it}")
            if opt == 2:
                assert_close(h1, r1, 2e-6, 2e-7, f"hip {name} z it{it}")
    if lib is not None:
        lib.hctr_updater_destroy(upd)


def test_static_lookup_addresses_equal_the_reference_kernel():
    """ragged_static_embedding_table_lookup_kernel: the address of every key's vector =
    table + (first row of the key's table + key) * ev, which is what hctr_static_lookup hands out"""
    rows, ev = [5, 40, 2], 8
    ref = RefStatic(rows, ev)
    total = sum(rows)
    table = np.zeros((total, ev), np.float32)
    rng = np.random.default_rng(3)
    per = [rng.integers(0, r, size=6) for r in rows]
    keys = np.concatenate(per).astype(np.int64)
    off = np.concatenate([[0], np.cumsum([len(p) for p in per])])
    got = ref.lookup(keys, off, table)
    base = table.ctypes.data
    start = np.concatenate([[0], np.cumsum(rows)])[:-1]
    want = np.concatenate([base + (start[t] + per[t]) * ev * 4 for t in range(len(rows))])
    assert np.array_equal(got, want.astype(np.uint64))


def test_keys_to_indices_equals_the_reference_kernel(oracle):
    """keys_to_indices_kernel (keys_to_indices.cu:23-43) over several lookups that share tables,
    row-sharded tables included (index = table's first local row + key / num_shards): the oracle's
    map and hctr_ebc_keys_to_indices (HIP source), lookup by lookup"""
    from hugectr_amd import _lib
    L = ctypes.CDLL(LIB)
    L.refstatic_keys_to_indices.argtypes = [ctypes.c_void_p, ctypes.c_size_t, ctypes.c_void_p,
                                            ctypes.c_int, ctypes.c_void_p, ctypes.c_void_p,
                                            ctypes.c_int] + [ctypes.c_void_p] * 2
    rng = np.random.default_rng(5)
    local_tables = np.array([0, 2, 3, 6], np.int32)            # the tables this GPU holds
    rows = np.array([100, 40, 900, 7], np.uint64)
    row_off = np.concatenate([[0], np.cumsum(rows)]).astype(np.uint64)
    num_shards = np.array([1, 1, 2, 4, 1, 1, 1], np.int32)     # by table id
    table_of_lookup = np.array([0, 2, 2, 3, 6, 0], np.int32)
    counts = [17, 0, 33, 64, 5, 300]
    per = [rng.integers(0, 2**31, size=c).astype(np.int64) for c in counts]
    keys = np.concatenate(per)
    off = np.concatenate([[0], np.cumsum(counts)]).astype(np.uint64)
    got = keys.copy()
    L.refstatic_keys_to_indices(_p(got), got.size, _p(off), len(counts), _p(table_of_lookup),
                                _p(local_tables), len(local_tables), _p(row_off), _p(num_shards))
    lib = None
    if emu.available():
        lib = emu.load_under_test()
    for l, k in enumerate(per):
        t = int(table_of_lookup[l])
        start = int(row_off[list(local_tables).index(t)])
        ns = int(num_shards[t])
        want = np.empty(k.size, np.int64)
        oracle.lib().hco_keys_to_indices(k.size, oracle._p(k), start, ns, oracle._p(want))
        seg = got[int(off[l]):int(off[l + 1])]
        assert np.array_equal(seg, want), ("oracle", l)
        if lib is not None and k.size:
            out = np.zeros(k.size, np.uint64)
            emu.check(lib, lib.hctr_ebc_keys_to_indices(_p(k), _lib.KEY_I64, k.size, start, ns,
                                                        _p(out), None))
            assert np.array_equal(out.astype(np.int64), seg), ("hip", l)
