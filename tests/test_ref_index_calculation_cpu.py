"""hctr_ebc_local_reduce (the HIP source's sort by row / first occurrences / unique rows / per-row
gradient sums, run by the host interpreter of tests/emu) next to the REFERENCE'S OWN DEVICE CODE of
LocalReduce's index calculation -- R/HugeCTR/embedding/operators/index_calculation.cu:
replicate_bucket_range_kernel, cal_table_range_kernel, get_keys_flag, get_unique_key, cut out of
the checkout (oracle/Makefile -> oracle/_ref/libref_index_calculation.so) and executed by the same
interpreter in the order LocalReduceIndexCalculation::cal_for_sparse_input runs them (partition by
table -> segmented sort -> segmented unique).  The unique (table, key) list, its order, and the
key -> unique-key map that LocalReduce sums by must be what the HIP path produces from row ids
(row = first row of the table + key): the same unique rows in the same order, the same gradient
sums when every unique key's gradients are added in the sorted list's order (bit for bit for rows
met at most 32 times)."""
import ctypes
import os
import sys

import numpy as np
import pytest

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "emu"))
import emu  # noqa: E402

LIB = os.path.join(os.path.dirname(__file__), "..", "oracle", "_ref", "libref_index_calculation.so")
pytestmark = pytest.mark.skipif(not (os.path.exists(LIB) and emu.available()),
                                reason="oracle/_ref not built (needs the reference checkout)")


def _p(a):
    return a.ctypes.data_as(ctypes.c_void_p)


@pytest.mark.parametrize("off32", [0, 1], ids=["int64_offsets", "uint32_offsets"])
@pytest.mark.parametrize("B,hot,rows,D", [(37, [1, 3, 7, 2, 5], [5, 40, 9, 300, 17], 8),
                                          (300, [2, 1, 4], [3, 1000, 60], 16),
                                          (5, [9], [4], 4)])
def test_local_reduce_equals_the_reference_index_calculation(B, hot, rows, D, off32):
    from hugectr_amd import _lib
    R = ctypes.CDLL(LIB)
    lib = emu.load_under_test()
    emu.bind(lib)
    rng = np.random.default_rng(B + len(hot) + off32)
    L = len(hot)
    # feature-major buckets (bucket = lookup * B + sample), 0 .. hot[l] keys each, duplicates wanted
    lens = np.concatenate([rng.integers(0, h + 1, size=B) for h in hot])
    br = np.concatenate([[0], np.cumsum(lens)]).astype(np.int64)
    nnz = int(br[-1])
    table_of = np.repeat(np.repeat(np.arange(L), B), lens)
    keys = (rng.integers(0, 1 << 30, size=nnz) % np.array(rows)[table_of]).astype(np.int64)
    row_start = np.concatenate([[0], np.cumsum(rows)[:-1]]).astype(np.int64)
    grad = rng.standard_normal((L * B, D)).astype(np.float32)
    # ---- the reference's kernels ------------------------------------------------------------------
    tids = np.arange(L, dtype=np.int32)               # lookup l reads table l (num_table == num_lookup)
    evs = np.full(L, D, np.int32)
    s_keys = np.full(nnz, -1, np.int64)
    s_src = np.full(nnz, 0xFFFFFFFF, np.uint32)
    t_ids = np.full(nnz + 1, -1, np.int32)
    u_keys = np.full(nnz, -1, np.int64)
    u_tids = np.full(nnz, -1, np.int32)
    dst = np.full(nnz, 0xFFFFFFFF, np.uint32)
    nu = np.zeros(1, np.uint64)
    brr = br.astype(np.uint32) if off32 else br
    assert R.refidx_local_reduce_indices(B, L, off32, _p(keys), _p(brr), _p(tids), _p(evs), _p(s_keys),
                                         _p(s_src), _p(t_ids), _p(u_keys), _p(u_tids), _p(dst),
                                         _p(nu)) == 0
    n_u = int(nu[0])
    # what the kernels say, checked against the definition first: keys sorted per table, source
    # buckets carried along in input order (a stable sort), firsts numbered in order
    assert (t_ids[:nnz] == table_of).all()
    comp = row_start[t_ids[:nnz]] + s_keys
    assert (np.diff(comp) >= 0).all() and n_u == np.unique(comp).size
    assert (np.sort(comp) == np.sort(row_start[table_of] + keys)).all()
    src_bucket_of_key = np.repeat(np.arange(L * B), lens)
    for t in range(L):  # stable: equal keys keep ascending source positions = ascending buckets
        m = t_ids[:nnz] == t
        order = np.argsort(keys[table_of == t], kind="stable")
        assert (s_src[m] == src_bucket_of_key[table_of == t][order]).all()
    ref_rows = row_start[u_tids[:n_u]] + u_keys[:n_u]
    assert (comp == ref_rows[dst]).all()
    # LocalReduce (model_backward.cu): every unique key's gradients added in the sorted list's order
    ref_wgrad = np.zeros((n_u, D), np.float32)
    for i in range(nnz):
        ref_wgrad[dst[i]] += grad[s_src[i]]
    # ---- the HIP source -----------------------------------------------------------------------------
    upd = ctypes.c_void_p()
    emu.check(lib, lib.hctr_updater_create(max(nnz, 1), int(sum(rows)), D, ctypes.byref(upd)))
    row_ids = (row_start[table_of] + keys).astype(np.uint64)
    h_urow = np.full(nnz, -1, np.int64)
    h_ukey = np.full(nnz, -1, np.int64)
    h_wgrad = np.full((nnz, D), np.nan, np.float32)
    n_h = ctypes.c_size_t()
    emu.check(lib, lib.hctr_ebc_local_reduce(upd, L * B, nnz, _p(br), _p(row_ids), int(sum(rows)),
                                             _p(keys), _p(grad), _lib.F32, ctypes.byref(n_h),
                                             _p(h_urow), _p(h_ukey), _p(h_wgrad), None))
    lib.hctr_updater_destroy(upd)
    assert n_h.value == n_u
    assert (h_urow[:n_u] == ref_rows).all(), "unique rows / their order"
    assert (h_ukey[:n_u] == u_keys[:n_u]).all(), "unique keys"
    # rows met at most 32 times are one ascending chain on both sides: the same bits; longer runs
    # are summed tile by tile in the HIP path (fixed order, another association: DESIGN section 3)
    cnt = np.bincount(dst, minlength=n_u)
    few = cnt <= 32
    assert np.array_equal(h_wgrad[:n_u][few].view(np.uint32), ref_wgrad[few].view(np.uint32)), \
        "gradient sums (short runs)"
    scale = np.abs(grad).max() * np.maximum(cnt, 1)[:, None]
    assert (np.abs(h_wgrad[:n_u] - ref_wgrad) <= 1e-6 * scale).all(), "gradient sums (long runs)"
