"""oracle/cache_oracle.py (the restated oracle of the embedding cache that the GPU tests of
hctr_cache_* compare against) against the REFERENCE'S OWN cache: gpu_cache::gpu_cache
(R/gpu_cache/src/nv_gpu_cache.cu -- get_kernel :392-531, insert_replace_kernel :700-852,
update_kernel :970-1074, dump_kernel :1155-1226, host methods :1279-1624; class
R/gpu_cache/include/nv_gpu_cache.hpp:50-120), which is CUDA source with no CPU mirror and no tests
of its own.  oracle/Makefile `ref` compiles that file from the reference checkout as plain C++
(ref_shims/cuda/: stand-ins for the CUDA headers; oracle/ref_launch_rewrite.py: the <<<>>> launch
syntax) and the host interpreter of tests/emu executes its kernels thread by thread -- CUDA threads
as fibers, the 32-lane cooperative-groups tiles as wavefronts of width 32, real atomics and the
real per-set spin locks -- into oracle/_ref/libref_cache.so.

Two schedules:
* thread blocks one after the other in block order: the per-set mutexes are then taken in
  key-position order, which is the interleaving the oracle restates -- results AND the internal
  state (key of every slot, LRU counter of every slot, vectors, global counter, Dump order) must
  be identical after every call;
* one OS thread per thread block (the locks are really contended, the order in which the keys of
  one call reach a set is whatever the threads make it): with distinct keys per call and Replace
  used as the tiered table uses it (on the keys a Query just missed) the cache as a MAP -- per set
  {key: (counter, vector)} -- and every Query result must still be the oracle's.
"""
import ctypes
import os

import numpy as np
import pytest

from oracle.cache_oracle import SLOTS, CacheOracle

LIB = os.path.join(os.path.dirname(__file__), "..", "oracle", "_ref", "libref_cache.so")
pytestmark = pytest.mark.skipif(not os.path.exists(LIB),
                                reason="oracle/_ref not built (needs the reference checkout)")


def _p(a):
    return a.ctypes.data_as(ctypes.c_void_p)


class RefCache:
    """the reference's gpu_cache<key, uint64_t, max, 2, 32> under the interpreter"""

    def __init__(self, sets, D, key_bytes, workers=1):
        L = self.L = ctypes.CDLL(LIB)
        P, Z = ctypes.c_void_p, ctypes.c_size_t
        L.refcache_create.restype = P
        L.refcache_create.argtypes = [Z, Z, ctypes.c_int]
        L.refcache_destroy.argtypes = [P]
        L.refcache_schedule.argtypes = [Z]
        L.refcache_query.argtypes = [P, P, Z, P, P, P, P]
        L.refcache_replace.argtypes = [P, P, Z, P]
        L.refcache_update.argtypes = [P, P, Z, P]
        L.refcache_dump.argtypes = [P, P, P, Z, Z]
        L.refcache_state.argtypes = [P] * 6
        self.sets, self.D, self.kb = sets, D, key_bytes
        self.kdt = np.int64 if key_bytes == 8 else np.uint32
        self.h = ctypes.c_void_p(L.refcache_create(sets, D, key_bytes))
        L.refcache_schedule(workers)

    def close(self):
        self.L.refcache_destroy(self.h)

    def query(self, keys, fill):
        n = len(keys)
        k = np.ascontiguousarray(keys, self.kdt)
        out = np.full((max(n, 1), self.D), fill, np.float32)
        mi = np.zeros(max(n, 1), np.uint64)
        mk = np.zeros(max(n, 1), self.kdt)
        ml = np.full(1, 12345, np.uint64)
        self.L.refcache_query(self.h, _p(k), n, _p(out), _p(mi), _p(mk), _p(ml))
        m = int(ml[0])
        order = np.argsort(mi[:m], kind="stable")  # (tiles append their misses as they finish)
        return out[:n], mi[:m][order].astype(np.int64), mk[:m][order].astype(np.int64)

    def replace(self, keys, values):
        k = np.ascontiguousarray(keys, self.kdt)
        v = np.ascontiguousarray(values, np.float32)
        self.L.refcache_replace(self.h, _p(k), len(k), _p(v))

    def update(self, keys, values):
        k = np.ascontiguousarray(keys, self.kdt)
        v = np.ascontiguousarray(values, np.float32)
        self.L.refcache_update(self.h, _p(k), len(k), _p(v))

    def dump(self, s0, s1):
        out = np.zeros(self.sets * SLOTS, self.kdt)
        cnt = np.zeros(1, np.uint64)
        self.L.refcache_dump(self.h, _p(out), _p(cnt), s0, s1)
        return out[:int(cnt[0])].astype(np.int64)

    def state(self):
        n = self.sets * SLOTS
        keys = np.zeros(n, np.int64)
        empty = np.zeros(n, np.uint8)
        cnt = np.zeros(n, np.uint64)
        vals = np.zeros((n, self.D), np.float32)
        g = np.zeros(1, np.uint64)
        self.L.refcache_state(self.h, _p(keys), _p(empty), _p(cnt), _p(vals), _p(g))
        return (keys.reshape(self.sets, SLOTS), empty.reshape(self.sets, SLOTS).astype(bool),
                cnt.reshape(self.sets, SLOTS), vals.reshape(self.sets, SLOTS, self.D), int(g[0]))


def _same_state(ref: RefCache, orc: CacheOracle, exact_slots: bool):
    keys, empty, cnt, vals, g = ref.state()
    assert g == orc.global_counter
    for s in range(ref.sets):
        if exact_slots:
            for slot in range(SLOTS):
                ok = orc.keys[s][slot]
                assert empty[s, slot] == (ok is None), (s, slot)
                if ok is not None:
                    assert keys[s, slot] == ok, (s, slot)
                    assert cnt[s, slot] == orc.cnt[s][slot], (s, slot)
                    assert np.array_equal(vals[s, slot], orc.vals[s, slot]), (s, slot)
        else:
            got = {int(keys[s, i]): (int(cnt[s, i]), vals[s, i].tobytes())
                   for i in range(SLOTS) if not empty[s, i]}
            want = {orc.keys[s][i]: (int(orc.cnt[s][i]), orc.vals[s, i].tobytes())
                    for i in range(SLOTS) if orc.keys[s][i] is not None}
            assert got == want, s


def _run(seed, sets, D, key_bytes, workers, calls, key_space, max_len, distinct):
    rng = np.random.default_rng(seed)
    ref = RefCache(sets, D, key_bytes, workers)
    orc = CacheOracle(sets, D, key_bytes)
    exact = workers == 1
    try:
        _same_state(ref, orc, exact)
        for c in range(calls):
            # (contended schedule: Replace only as the tiered table uses it, behind a Query -- the
            #  Query ages everything that is cached, so the keys of ONE call never evict each
            #  other and the outcome does not depend on the order the locks are taken in; a bare
            #  Replace gives its keys the age of older entries, and which of two same-age keys is
            #  evicted IS the lock order, in the reference as in any implementation)
            op = rng.choice(["query", "replace", "update", "dump", "miss_fill"],
                            p=[0.25, 0.3, 0.15, 0.1, 0.2] if exact else [0.3, 0.0, 0.2, 0.1, 0.4])
            n = int(rng.integers(0, max_len + 1))
            if distinct:
                keys = rng.choice(key_space, size=min(n, key_space), replace=False)
            else:
                keys = rng.integers(0, key_space, size=n)
            keys = keys.astype(np.int64)
            vals = rng.standard_normal((len(keys), D)).astype(np.float32)
            if op == "query":
                out, mi, mk = ref.query(keys, -7.0)
                want = np.full((len(keys), D), -7.0, np.float32)
                wmi, wmk = orc.query(keys, want)
                assert np.array_equal(mi, wmi) and np.array_equal(mk, wmk), (c, op)
                assert np.array_equal(out, want), (c, op)
            elif op == "miss_fill":
                # the cache's use in the tiered table: Query, then Replace with the missing keys
                out, mi, mk = ref.query(keys, 0.0)
                want = np.zeros((len(keys), D), np.float32)
                wmi, wmk = orc.query(keys, want)
                assert np.array_equal(mi, wmi) and np.array_equal(mk, wmk), (c, op)
                assert np.array_equal(out, want), (c, op)
                if distinct or workers == 1:
                    ref.replace(mk, vals[mi])
                    orc.replace(wmk, vals[wmi])
            elif op == "replace":
                ref.replace(keys, vals)
                orc.replace(keys, vals)
            elif op == "update":
                if not distinct:  # (two values for one key in one call: last writer wins -- any
                    _, first = np.unique(keys, return_index=True)  # order is allowed; keep one)
                    sel = np.sort(first)
                    keys, vals = keys[sel], vals[sel]
                ref.update(keys, vals)
                orc.update(keys, vals)
            else:
                s0 = int(rng.integers(0, sets))
                s1 = int(rng.integers(s0 + 1, sets + 1))
                got, want = ref.dump(s0, s1), orc.dump(s0, s1)
                if exact:
                    assert np.array_equal(got, want), (c, op)
                else:
                    assert np.array_equal(np.sort(got), np.sort(want)), (c, op)
            _same_state(ref, orc, exact)
    finally:
        ref.close()


@pytest.mark.parametrize("key_bytes", [8, 4])
@pytest.mark.parametrize("seed,sets,D,key_space,max_len", [
    (1, 1, 4, 200, 40),       # one set: every insert beyond 64 keys evicts
    (2, 3, 8, 400, 70),       # a few sets, more keys than slots
    (3, 8, 16, 600, 150),     # several thread blocks per call
    (4, 5, 3, 100, 90),       # mostly hits (key space < capacity): refresh / update paths
])
def test_oracle_equals_reference_cache_in_key_order(seed, sets, D, key_space, max_len, key_bytes):
    """blocks in order = the oracle's interleaving: everything identical, slot by slot"""
    _run(seed * 11 + key_bytes, sets, D, key_bytes, 1, 30, key_space, max_len, distinct=False)


@pytest.mark.parametrize("key_bytes", [8, 4])
@pytest.mark.parametrize("seed,sets,D,key_space,max_len", [
    (5, 2, 4, 500, 60),       # (at most 64 keys of a call per set: hits + inserts fit the set)
    (6, 6, 8, 900, 64),
])
def test_reference_cache_under_contention_keeps_what_the_oracle_keeps(seed, sets, D, key_space,
                                                                       max_len, key_bytes):
    """thread blocks on concurrent OS threads: the order in which the keys of one call take a
    set's lock is whatever the threads make it.  WHICH of several equally old entries leaves then
    depends on that order (tie rule = the probing order of the inserting key), in the reference as
    in any implementation, so slot contents are not compared; what no order may change is: every
    hit returns the vector last written for its key, the keys a Query missed and a Replace brought
    in are hits right afterwards, a write-through reaches exactly the cached keys, and per set the
    number of cached keys and the global counter are the oracle's."""
    rng = np.random.default_rng(seed * 13 + key_bytes)
    ref, orc = RefCache(sets, D, key_bytes, 0), CacheOracle(sets, D, key_bytes)
    truth = {}
    try:
        for c in range(25):
            n = int(rng.integers(1, max_len + 1))
            keys = rng.choice(key_space, size=n, replace=False).astype(np.int64)
            vals = rng.standard_normal((n, D)).astype(np.float32)
            out, mi, mk = ref.query(keys, np.nan)
            wmi, _ = orc.query(keys, np.zeros((n, D), np.float32))
            hit = np.ones(n, bool)
            hit[mi] = False
            assert np.array_equal(mk, keys[mi])
            for i in np.nonzero(hit)[0]:
                assert np.array_equal(out[i], truth[int(keys[i])]), (c, i)
            assert np.isnan(out[~hit]).all()
            if rng.random() < 0.7:  # fill the misses (the tiered table's use of the cache)
                ref.replace(mk, vals[mi])
                # (the oracle replays ITS OWN misses: which of two equally old entries left
                #  earlier may differ, the number of entries per set may not)
                orc.replace(keys[wmi], vals[wmi])
                for i in mi:
                    truth[int(keys[i])] = vals[i]
                out2, mi2, _ = ref.query(keys, np.nan)
                orc.query(keys, np.zeros((n, D), np.float32))
                assert len(mi2) == 0, (c, "keys just brought in must be hits")
                for i in range(n):
                    assert np.array_equal(out2[i], truth[int(keys[i])]), (c, i)
            if rng.random() < 0.4:  # write-through of new vectors for some keys
                sel = rng.random(n) < 0.5
                nv = rng.standard_normal((int(sel.sum()), D)).astype(np.float32)
                ref.update(keys[sel], nv)
                rk, rempty, _, _, _ = ref.state()
                cached = set(rk[~rempty].tolist())
                for k, v in zip(keys[sel], nv):
                    if int(k) in cached:
                        truth[int(k)] = v
            _, empty, cnt, _, g = ref.state()
            assert g == orc.global_counter
            for s in range(sets):
                want = sum(k is not None for k in orc.keys[s])
                assert int((~empty[s]).sum()) == want, (c, s)
    finally:
        ref.close()


def test_lru_victim_and_tie_order_of_the_reference():
    """a full set, known ages: the victim is the slot with the smallest counter, ties go to the
    first slab in probing order (key % 2 first), then to the lowest lane"""
    sets, D = 1, 2
    ref, orc = RefCache(sets, D, 8, 1), CacheOracle(sets, D, 8)
    try:
        keys = np.arange(64, dtype=np.int64)
        vals = np.arange(128, dtype=np.float32).reshape(64, 2)
        for c in (ref, orc):  # all 64 slots filled at age 0 (no Query yet)
            c.replace(keys, vals)
        _same_state(ref, orc, True)
        # touch everything except keys 10, 11, 40 -> those three are the oldest, equal ages
        touched = np.array([k for k in range(64) if k not in (10, 11, 40)], np.int64)
        ref.query(touched, 0.0)
        orc.query(touched, np.zeros((len(touched), D), np.float32))
        for new in (1001, 1002, 1003):
            nk = np.array([new], np.int64)
            nv = np.full((1, D), float(new), np.float32)
            ref.replace(nk, nv)
            orc.replace(nk, nv)
            _same_state(ref, orc, True)
        out, mi, mk = ref.query(np.array([10, 11, 40, 1001, 1002, 1003], np.int64), -1.0)
        assert list(mi) == [0, 1, 2] and list(mk) == [10, 11, 40]
        assert np.array_equal(out[3:, 0], [1001.0, 1002.0, 1003.0])
    finally:
        ref.close()
