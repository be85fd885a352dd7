"""oracle/hctr_oracle.c's embedding_collection reference (hco_ebc_forward /
hco_ebc_backward_update -- what the GPU tests of hctr_ebc_* compare against) against the
REFERENCE'S OWN CPU reference of embedding_collection: EmbeddingTableCPU + EmbeddingReferenceCPU
(R/test/utest/embedding_collection/embedding_table_cpu.hpp:27-125, reference_embedding.hpp:32-237),
compiled from the reference checkout into oracle/_ref/libref_ebc.so (oracle/Makefile `ref`,
oracle/ref_ebc_shim.cpp; oracle/ref_shims/ebc/ replaces the CUDA-bound headers with declarations).
Same tables, keys and top gradients on both sides: forward for Sum / Average lookups that share
tables, FeatureMajor and BatchMajor outputs, 1 / 2 / 4 GPUs; backward + SGD update over 3 steps."""
import ctypes
import os

import numpy as np
import pytest

from oracle import pyoracle as po

LIB = os.path.join(os.path.dirname(__file__), "..", "oracle", "_ref", "libref_ebc.so")
pytestmark = pytest.mark.skipif(not os.path.exists(LIB),
                                reason="oracle/_ref not built (needs the reference checkout)")


def _ref():
    L = ctypes.CDLL(LIB)
    P, I, F, Z = ctypes.c_void_p, ctypes.c_int, ctypes.c_float, ctypes.c_size_t
    L.ref_ebc_create.restype = P
    L.ref_ebc_create.argtypes = [I, I, I, P, F, F, I, P, P, P, I, P, P, P]
    L.ref_ebc_destroy.argtypes = [P, I]
    L.ref_ebc_forward.argtypes = [P, I, P, Z, P, Z, P, Z]
    L.ref_ebc_backward_update.argtypes = [P, I, P, Z, P, Z, P, Z]
    L.ref_ebc_get.argtypes = [P, I, I, ctypes.c_longlong, P]
    return L


def _p(a):
    return a.ctypes.data_as(ctypes.c_void_p)


@pytest.mark.parametrize("num_gpus", [1, 2, 4])
@pytest.mark.parametrize("batch_major", [0, 1])
def test_ebc_oracle_matches_the_reference_cpu_reference(num_gpus, batch_major):
    L = _ref()
    rng = np.random.default_rng(10 * num_gpus + batch_major)
    ev, batch = 8, 16
    rows = [30, 7, 50]                       # three tables
    lookup_table = [0, 1, 2, 0, 1]           # five lookups, tables 0 and 1 are shared
    combiners = [0, 1, 0, 1, 0]              # Combiner::Sum = 0, Average = 1 on both sides
    hot = [3, 4, 1, 5, 2]
    nl = len(lookup_table)
    row_start = np.concatenate([[0], np.cumsum(rows)[:-1]]).astype(np.int64)
    tables = (rng.standard_normal((sum(rows), ev)) * 0.3).astype(np.float32)
    keys_all = np.concatenate([np.arange(r) for r in rows]).astype(np.int64)
    koffs = np.concatenate([[0], np.cumsum(rows)]).astype(np.uint32)
    lr, scaler = 0.25, 4.0
    h = L.ref_ebc_create(0, num_gpus, len(rows), _p(np.full(len(rows), ev, np.int32)), lr, scaler,
                         nl, _p(np.array(lookup_table, np.int32)), _p(np.array(combiners, np.int32)),
                         _p(np.array(hot, np.int32)), batch_major, _p(keys_all), _p(koffs),
                         _p(tables.copy()))
    assert h
    try:
        per_gpu = nl * ev * (batch // num_gpus)
        for step in range(3):
            lens = np.concatenate([rng.integers(0, hot[l] + 1, batch) for l in range(nl)])
            lens[rng.integers(0, lens.size, 4)] = 0          # some empty buckets
            br = np.concatenate([[0], np.cumsum(lens)]).astype(np.int64)
            keys = np.concatenate([rng.integers(0, rows[lookup_table[l]],
                                                int(lens[l * batch:(l + 1) * batch].sum()))
                                   for l in range(nl)]).astype(np.int64)
            got = np.empty(num_gpus * per_gpu, np.float32)
            br32 = br.astype(np.uint32)
            assert L.ref_ebc_forward(h, 0, _p(keys), keys.size, _p(br32), br32.size, _p(got),
                                     per_gpu) == 0
            want = po.ebc_forward(batch, lookup_table, ev, combiners, keys, br, row_start, tables,
                                  num_gpus=num_gpus, batch_major=bool(batch_major))
            # same additions in the same order (a vector is summed key by key, then divided)
            assert np.array_equal(want.reshape(-1), got), f"forward step {step}"
            g = rng.standard_normal(num_gpus * per_gpu).astype(np.float32)
            # Average with an empty bucket divides 0 / 0 in the reference only for keys of that
            # bucket, of which there are none: no NaN can reach a table
            assert L.ref_ebc_backward_update(h, 0, _p(g), per_gpu, _p(keys), keys.size, _p(br32),
                                             br32.size) == 0
            po.ebc_backward_update(batch, lookup_table, ev, combiners, keys, br, row_start, tables,
                                   g, optimizer=0, lr=lr, scaler=scaler, num_gpus=num_gpus,
                                   batch_major=bool(batch_major))
            ref_tab = np.empty_like(tables)
            for t, r in enumerate(rows):
                for k in range(r):
                    assert L.ref_ebc_get(h, 0, t, k, _p(ref_tab[row_start[t] + k])) == 0
            # the reference sums a key's gradients in hash-map bucket order with Kahan
            # compensation, the oracle in bucket order with the same compensation: equal to rounding
            np.testing.assert_allclose(tables, ref_tab, rtol=1e-6, atol=1e-7,
                                       err_msg=f"tables after step {step}")
    finally:
        L.ref_ebc_destroy(h, 0)
