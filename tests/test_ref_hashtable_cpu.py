"""The oracle's hash table AND the HIP index stage's source against the REFERENCE'S OWN GPU hash
table: HugeCTR::HashTable<KeyType, size_t> (R/HugeCTR/src/hashtable/nv_hashtable.cu:36-345) on the
cuDF-derived concurrent_unordered_map (R/HugeCTR/include/hashtable/cudf/
concurrent_unordered_map.cuh:280-752: bucket claimed with atomicCAS, linear probing from
MurmurHash3_32(key) % size, row number = atomicAdd on the table's counter), compiled from the
checkout as plain C++ and executed by the host interpreter of tests/emu
(oracle/_ref/libref_hashtable.so, oracle/Makefile `ref`).  With thread blocks in block order the keys
of a call are inserted in position order -- the order the oracle (and the HIP kernels, by
construction) reproduce -- so everything observable must be identical: the row of every key, what
get_mark answers for unseen keys, size, value head, the bucket array's size and the (key, row)
pairs of dump() IN BUCKET ORDER (i.e. the same probing, bucket for bucket)."""
import ctypes
import os
import sys

import numpy as np
import pytest

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "emu"))
import emu  # noqa: E402

LIB = os.path.join(os.path.dirname(__file__), "..", "oracle", "_ref", "libref_hashtable.so")
pytestmark = pytest.mark.skipif(not os.path.exists(LIB),
                                reason="oracle/_ref not built (needs the reference checkout)")
INVALID = np.uint64(0xFFFFFFFFFFFFFFFF)


def _p(a):
    return a.ctypes.data_as(ctypes.c_void_p)


class RefTable:
    def __init__(self, capacity, kb):
        L = self.L = ctypes.CDLL(LIB)
        P, Z = ctypes.c_void_p, ctypes.c_size_t
        L.refht_create.restype = P
        L.refht_create.argtypes = [Z, ctypes.c_int]
        L.refht_destroy.argtypes = [P]
        L.refht_schedule.argtypes = [Z]
        for f in ("refht_get_insert", "refht_get_mark", "refht_insert"):
            getattr(L, f).argtypes = [P, P, P, Z]
        for f in ("refht_size", "refht_value_head", "refht_capacity"):
            getattr(L, f).restype = Z
            getattr(L, f).argtypes = [P]
        L.refht_dump.restype = Z
        L.refht_dump.argtypes = [P, P, P]
        self.kdt = np.int64 if kb == 8 else np.uint32
        self.h = ctypes.c_void_p(L.refht_create(capacity, kb))

    def close(self):
        self.L.refht_destroy(self.h)

    def get_insert(self, keys):
        k = np.ascontiguousarray(keys, self.kdt)
        v = np.full(k.size, 7, np.uint64)
        self.L.refht_get_insert(self.h, _p(k), _p(v), k.size)
        return v

    def get_mark(self, keys):
        k = np.ascontiguousarray(keys, self.kdt)
        v = np.full(k.size, 7, np.uint64)
        self.L.refht_get_mark(self.h, _p(k), _p(v), k.size)
        return v

    def size(self):
        return self.L.refht_size(self.h)

    def value_head(self):
        return self.L.refht_value_head(self.h)

    def table_size(self):
        return self.L.refht_capacity(self.h)

    def dump(self):
        n = self.table_size()
        k = np.zeros(n, self.kdt)
        v = np.zeros(n, np.uint64)
        c = self.L.refht_dump(self.h, _p(k), _p(v))
        return k[:c].astype(np.int64), v[:c]


def _batches(rng, kb, capacity, calls):
    """an irregular call sequence: new keys, repeats inside a call, repeats of earlier calls, long
    and empty calls, clustered keys (probe chains); never more distinct keys than `capacity`"""
    hi = 2**31 - 1 if kb == 8 else 2**32 - 2
    seen = np.zeros(0, np.int64)
    for c in range(calls):
        n = int(rng.choice([0, 1, 3, 40, 257, 700]))
        room = capacity - np.unique(seen).size
        fresh = rng.integers(0, hi, size=min(n, max(room, 0)))
        if c % 3 == 1 and fresh.size:  # consecutive keys
            fresh = (int(fresh[0]) + np.arange(fresh.size)) % hi
        old = rng.choice(seen, size=min(n // 2, seen.size)) if seen.size else np.zeros(0, np.int64)
        keys = np.concatenate([fresh, old, fresh[: fresh.size // 3]]).astype(np.int64)
        rng.shuffle(keys)
        while np.unique(np.concatenate([seen, keys])).size > capacity:
            keys = keys[:-1]
        seen = np.concatenate([seen, keys])
        yield keys


@pytest.mark.parametrize("kb", [8, 4])
@pytest.mark.parametrize("capacity,calls,seed", [(64, 12, 1), (1000, 14, 2), (3000, 10, 3)])
def test_oracle_hash_table_equals_the_reference_gpu_table(oracle, kb, capacity, calls, seed):
    rng = np.random.default_rng(seed * 10 + kb)
    ref, orc = RefTable(capacity, kb), oracle.HashTable(capacity, kb)
    try:
        assert ref.table_size() == orc.table_size()
        for keys in _batches(rng, kb, capacity, calls):
            probe = rng.integers(0, 2**31 - 1, size=50).astype(np.int64)
            assert np.array_equal(ref.get_mark(probe), orc.get_mark(probe))
            assert np.array_equal(ref.get_insert(keys), orc.get_insert(keys))
            assert ref.size() == orc.size() and ref.value_head() == orc.value_head()
            assert np.array_equal(ref.get_mark(keys), orc.get_mark(keys))
        rk, rv = ref.dump()
        ok, ov = orc.dump()
        assert np.array_equal(rk, np.asarray(ok, np.int64)) and np.array_equal(rv, ov), \
            "dump: same pairs in the same bucket order"
    finally:
        ref.close()


@pytest.mark.parametrize("kb", [8, 4])
@pytest.mark.parametrize("capacity,calls,seed", [(64, 10, 4), (2000, 12, 5)])
def test_hip_index_stage_source_equals_the_reference_gpu_table(kb, capacity, calls, seed):
    """hctr_ht_* (hugectr_amd/csrc/hashtable.hip under the interpreter: probe + cooperative finish
    kernel) against the reference's table: rows, marks, size, value head, bucket-array size and
    the dumped pairs"""
    if not emu.available():
        pytest.skip("no host clang++ / make")
    from hugectr_amd import _lib
    lib = emu.load_under_test()
    rng = np.random.default_rng(seed * 10 + kb)
    ref = RefTable(capacity, kb)
    hip = emu.HashTable(lib, capacity, _lib.KEY_I64 if kb == 8 else _lib.KEY_U32)
    kdt = np.int64 if kb == 8 else np.uint32
    try:
        assert lib.hctr_ht_table_size(hip.h) == ref.table_size()
        for keys in _batches(rng, kb, capacity, calls):
            if keys.size == 0:
                continue
            k = np.ascontiguousarray(keys, kdt)
            assert np.array_equal(hip.get_insert(k), ref.get_insert(keys))
            assert hip.size() == ref.size() and hip.value_head() == ref.value_head()
            probe = np.ascontiguousarray(
                np.concatenate([keys[:20], rng.integers(0, 2**31 - 1, size=30)]), kdt)
            assert np.array_equal(hip.get_mark(probe), ref.get_mark(probe))
        # dump(): the same (key, row) pairs.  Their ORDER is the bucket order, and which of two new
        # keys of ONE call that probe the same bucket claims it is the atomicCAS race in both
        # implementations (rows do not depend on it: they follow first occurrence); the reference
        # run here inserts in position order, the HIP kernel inserts a call's keys concurrently
        hk, hv = hip.dump()
        rk, rv = ref.dump()
        hk = np.asarray(hk, np.int64) & (0xFFFFFFFF if kb == 4 else -1)
        assert sorted(zip(hk.tolist(), hv.tolist())) == sorted(zip(rk.tolist(), rv.tolist()))
    finally:
        ref.close()
