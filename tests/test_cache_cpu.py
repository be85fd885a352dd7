"""The cache oracle against hand-worked scenarios of the reference's rules
(R/gpu_cache/src/nv_gpu_cache.cu:541-697): probing starts in slab key % 2, empty slots first,
least-recently-used eviction with ties broken in probing order, Query refreshes, Update never
inserts.  (The reference ships no golden vectors for gpu_cache; tests/test_ref_cache_cpu.py pins
the oracle against the reference's own kernels executed by the host interpreter.)"""
import numpy as np

from oracle.cache_oracle import CacheOracle, TieredOracle, murmur3_32


def test_murmur_known_answers():
    # MurmurHash3_x86_32 reference values (public domain test vectors)
    assert murmur3_32(b"", 0) == 0
    assert murmur3_32(b"", 1) == 0x514E28B7
    assert murmur3_32(b"\xff\xff\xff\xff", 0) == 0x76293B50
    assert murmur3_32(b"\x21\x43\x65\x87", 0) == 0xF55B516B


def test_probing_and_lru_rules():
    o = CacheOracle(1, 2)
    v = lambda k: np.array([k, -k], np.float32)
    even = list(range(0, 64, 2))   # 32 even keys fill slab 0
    o.query(np.array([0]), None)   # counter 1
    o.replace(np.array(even), np.stack([v(k) for k in even]))
    assert o.keys[0][:32] == even and all(k is None for k in o.keys[0][32:])
    # 33rd even key: slab 0 is full, first empty slot of slab 1
    o.replace(np.array([64]), v(64)[None])
    assert o.keys[0][32] == 64
    odd = list(range(1, 62, 2))    # 31 odd keys take the rest of slab 1
    o.replace(np.array(odd), np.stack([v(k) for k in odd]))
    assert o.keys[0][33:] == odd and None not in o.keys[0]
    # all slots carry counter 1; a Query refreshes keys 0 and 1 to counter 2
    out = np.zeros((2, 2), np.float32)
    mi, mk = o.query(np.array([0, 1]), out)
    assert mi.size == 0 and (out == np.stack([v(0), v(1)])).all()
    # an odd newcomer probes slab 1 first: evicts the first counter-1 slot there (slot 32, key 64)
    o.replace(np.array([101]), v(101)[None])
    assert o.keys[0][32] == 101 and 64 not in o.keys[0]
    # an even newcomer probes slab 0 first: slot 0 holds key 0 (counter 2) -> evicts slot 1 (key 2)
    o.replace(np.array([200]), v(200)[None])
    assert o.keys[0][1] == 200 and o.keys[0][0] == 0
    # Update never inserts; it overwrites cached vectors only
    o.update(np.array([999, 0]), np.stack([v(5), v(6)]))
    assert 999 not in o.keys[0] and (o.vals[0, 0] == v(6)).all()
    mi, mk = o.query(np.array([2, 64, 0]), None)
    assert mk.tolist() == [2, 64]
    assert set(o.dump(0, 1).tolist()) == set(k for k in o.keys[0])


def test_tiered_oracle_write_through():
    t = TieredOracle(10, 2, 1)
    t.host[:] = np.arange(20, dtype=np.float32).reshape(10, 2)
    out, nmiss = t.lookup(np.array([3, 3, 11, 4]))
    assert nmiss == 4 and (out[0] == [6, 7]).all() and (out[2] == 0).all()
    out, nmiss = t.lookup(np.array([3, 4, 5]))
    assert nmiss == 1
    t.scatter(np.array([3, 9]), np.ones((2, 2), np.float32), add=True)
    assert (t.host[3] == [7, 8]).all() and (t.host[9] == [19, 20]).all()
    out, _ = t.lookup(np.array([3, 9]))
    assert (out == [[7, 8], [19, 20]]).all()
