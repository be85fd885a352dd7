"""The oracle's hash functions against the REFERENCE's own, compiled from
R/HugeCTR/include/hashtable/cudf/hash_functions.cuh and R/gpu_cache/include/hash_functions.cuh into
oracle/_ref/libref_hash.so (oracle/ref_hash_shim.cpp, `make -C oracle ref`): the MurmurHash3_32 of
the embedding hash table (4- and 8-byte keys) -- which fixes the slot every key probes first -- and
the set / slab hash of the embedding cache.  The library is built in the container where the
reference is mounted and travels with the repository snapshot; skipped when it was never built."""
import ctypes
import os
import struct

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
LIB = os.path.join(ROOT, "oracle", "_ref", "libref_hash.so")

pytestmark = pytest.mark.skipif(not os.path.exists(LIB), reason="oracle/_ref not built")


def _ref():
    L = ctypes.CDLL(LIB)
    for n in ("ref_murmur3_u32", "ref_cache_murmur3_u32"):
        getattr(L, n).restype = ctypes.c_uint32
        getattr(L, n).argtypes = [ctypes.c_uint32]
    for n in ("ref_murmur3_i64", "ref_cache_murmur3_i64"):
        getattr(L, n).restype = ctypes.c_uint32
        getattr(L, n).argtypes = [ctypes.c_longlong]
    L.ref_cache_mod_hash_i64.restype = ctypes.c_size_t
    L.ref_cache_mod_hash_i64.argtypes = [ctypes.c_longlong]
    L.ref_murmur3_i64_many.argtypes = [ctypes.c_void_p, ctypes.c_size_t, ctypes.c_void_p]
    L.ref_murmur3_u32_many.argtypes = [ctypes.c_void_p, ctypes.c_size_t, ctypes.c_void_p]
    return L


def test_table_hash_equals_the_reference(oracle):
    L = _ref()
    rng = np.random.default_rng(0)
    k64 = np.concatenate([rng.integers(-2 ** 63, 2 ** 63 - 1, size=200000, dtype=np.int64),
                          np.arange(-3, 1000, dtype=np.int64),
                          np.array([2 ** 63 - 1, -2 ** 63, 2 ** 40 + 7], np.int64)])
    want = np.empty(k64.size, np.uint32)
    L.ref_murmur3_i64_many(k64.ctypes.data, k64.size, want.ctypes.data)
    got = oracle.hash_keys(k64, 8)
    assert (got == want).all()
    k32 = np.concatenate([rng.integers(0, 2 ** 32, size=200000, dtype=np.uint64).astype(np.uint32),
                          np.array([0, 1, 0xFFFFFFFE, 0xFFFFFFFF], np.uint32)])
    want = np.empty(k32.size, np.uint32)
    L.ref_murmur3_u32_many(k32.ctypes.data, k32.size, want.ctypes.data)
    got = oracle.hash_keys(k32.astype(np.int64), 4)
    assert (got == want).all()
    # the byte-string entry point the known-answer tests use
    for k in (0, 1, 123456789, 2 ** 40 + 7, -5):
        assert oracle.murmur3_32(struct.pack("<q", k)) == L.ref_murmur3_i64(k)


def test_cache_hashes_equal_the_reference():
    from oracle.cache_oracle import CacheOracle, murmur3_32
    L = _ref()
    rng = np.random.default_rng(1)
    o = CacheOracle(1000, 4)
    for k in rng.integers(-2 ** 62, 2 ** 62, size=3000).tolist() + [0, 1, -1, 63, 64]:
        h = L.ref_cache_murmur3_i64(k)
        assert murmur3_32(struct.pack("<q", k)) == h
        assert o._set(k) == h % 1000                               # set_hasher % capacity_in_set
        # slab_hasher = Mod_Hash: (size_t)key % SET_ASSOCIATIVITY picks the first slab probed
        assert o._probe_order(k)[0] // 32 == L.ref_cache_mod_hash_i64(k) % 2
    o4 = CacheOracle(77, 4, key_bytes=4)
    for k in rng.integers(0, 2 ** 32, size=2000).tolist():
        assert o4._set(k) == L.ref_cache_murmur3_u32(k) % 77
