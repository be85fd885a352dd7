"""CPU restatement of the embedding cache (TEST INFRASTRUCTURE ONLY, like everything under oracle/):
gpu_cache::gpu_cache Query / Replace / Update / Dump
(R/gpu_cache/src/nv_gpu_cache.cu:247-388 get_kernel, :541-697 insert_replace_kernel, :860-967
update_kernel, :1080-1153 dump_kernel; constants R/gpu_cache/include/nv_gpu_cache.hpp:30-31) executed
sequentially in key-position order -- one of the interleavings the reference's per-set mutexes
allow, and the one the HIP implementation fixes.  The tiered table on top follows the role of
gpu_cache::UvmTable (R/gpu_cache/include/uvm_table.hpp:133-174).
The reference has no tests, callers or golden vectors for these classes, so CacheOracle is pinned
against the reference's OWN kernels instead: oracle/Makefile `ref` compiles nv_gpu_cache.cu from
the checkout as plain C++ and the host interpreter of tests/emu executes it (threads = fibers,
32-lane tiles = wavefronts, thread blocks in block order = key-position order) into
oracle/_ref/libref_cache.so; tests/test_ref_cache_cpu.py compares every Query result, the Dump
order and the internal state (key, LRU counter and vector of every slot, global counter) after
each call of random Query / Replace / Update / Dump sequences, u32 and i64 keys.  TieredOracle
(our tiered table's flow, not UvmTable's internals) has nothing to be pinned against beyond that."""
import struct

import numpy as np

SET_ASSOCIATIVITY = 2
SLAB_SIZE = 32
SLOTS = SET_ASSOCIATIVITY * SLAB_SIZE


def _rotl(x, r):
    return ((x << r) | (x >> (32 - r))) & 0xFFFFFFFF


def murmur3_32(data: bytes, seed: int = 0) -> int:
    """MurmurHash3_x86_32 (R/gpu_cache/include/hash_functions.cuh = the hash of the path's table)"""
    c1, c2 = 0xCC9E2D51, 0x1B873593
    h = seed
    n = len(data) // 4
    for i in range(n):
        k = struct.unpack_from("<I", data, 4 * i)[0]
        k = (k * c1) & 0xFFFFFFFF
        k = _rotl(k, 15)
        k = (k * c2) & 0xFFFFFFFF
        h ^= k
        h = _rotl(h, 13)
        h = (h * 5 + 0xE6546B64) & 0xFFFFFFFF
    h ^= len(data)
    h ^= h >> 16
    h = (h * 0x85EBCA6B) & 0xFFFFFFFF
    h ^= h >> 13
    h = (h * 0xC2B2AE35) & 0xFFFFFFFF
    h ^= h >> 16
    return h


class CacheOracle:
    def __init__(self, capacity_in_set, vec_size, key_bytes=8):
        self.num_sets, self.D, self.kb = capacity_in_set, vec_size, key_bytes
        self.keys = [[None] * SLOTS for _ in range(capacity_in_set)]   # None = empty_key
        self.cnt = [[0] * SLOTS for _ in range(capacity_in_set)]
        self.vals = np.zeros((capacity_in_set, SLOTS, vec_size), np.float32)
        self.global_counter = 0

    def _set(self, key):
        raw = struct.pack("<q" if self.kb == 8 else "<I", int(key))
        return murmur3_32(raw) % self.num_sets

    def _probe_order(self, key):
        first = int(key) % SET_ASSOCIATIVITY                       # Mod_Hash
        return [((first + d) % SET_ASSOCIATIVITY) * SLAB_SIZE + s
                for d in range(SET_ASSOCIATIVITY) for s in range(SLAB_SIZE)]

    def query(self, keys, values):
        """values [len, D] is written for hits only; returns (missing_index, missing_keys)"""
        self.global_counter += 1                                   # update_kernel_overflow_ignore
        mi, mk = [], []
        for i, k in enumerate(keys):
            s = self._set(k)
            if int(k) in self.keys[s]:
                slot = self.keys[s].index(int(k))
                self.cnt[s][slot] = self.global_counter
                if values is not None:
                    values[i] = self.vals[s, slot]
            else:
                mi.append(i)
                mk.append(int(k))
        return np.array(mi, np.int64), np.array(mk, np.int64)

    def replace(self, keys, values):
        for i, k in enumerate(keys):
            s = self._set(k)
            ks = self.keys[s]
            if int(k) in ks:                                       # refresh only
                self.cnt[s][ks.index(int(k))] = self.global_counter
                continue
            order = self._probe_order(k)
            empty = [slot for slot in order if ks[slot] is None]
            if empty:
                slot = empty[0]
            else:                                                  # LRU, ties in probing order
                mn = min(self.cnt[s])
                slot = [x for x in order if self.cnt[s][x] == mn][0]
            ks[slot] = int(k)
            self.cnt[s][slot] = self.global_counter
            self.vals[s, slot] = values[i]

    def update(self, keys, values):
        for i, k in enumerate(keys):
            s = self._set(k)
            if int(k) in self.keys[s]:
                self.vals[s, self.keys[s].index(int(k))] = values[i]

    def dump(self, start_set, end_set):
        return np.array([k for s in range(start_set, end_set) for k in self.keys[s] if k is not None],
                        np.int64)


class TieredOracle:
    def __init__(self, host_rows, vec_size, capacity_in_set):
        self.host = np.zeros((host_rows, vec_size), np.float32)
        self.cache = CacheOracle(capacity_in_set, vec_size)

    def lookup(self, keys):
        out = np.zeros((len(keys), self.host.shape[1]), np.float32)
        mi, mk = self.cache.query(keys, out)
        for i, k in zip(mi, mk):
            if 0 <= k < self.host.shape[0]:
                out[i] = self.host[k]
        self.cache.replace(mk, out[mi])
        return out, len(mi)

    def scatter(self, unique_keys, values, add):
        for i, k in enumerate(unique_keys):
            k = int(k)
            if not (0 <= k < self.host.shape[0]):
                continue
            s = self.cache._set(k)
            cached = k in self.cache.keys[s]
            slot = self.cache.keys[s].index(k) if cached else -1
            old = self.cache.vals[s, slot] if cached else self.host[k]
            new = (old + values[i]).astype(np.float32) if add else values[i].astype(np.float32)
            self.host[k] = new
            if cached:
                self.cache.vals[s, slot] = new


class UvmOracle:
    """gpu_cache::UvmTable restated (R/gpu_cache/include/uvm_table.hpp:127-174): key -> vector, unknown
    keys read the default value; rows of the host store are handed out in order of first occurrence
    (add and the training lookup alike), the cache above is TieredOracle's, keyed by row."""

    def __init__(self, device_table_capacity, host_table_capacity, vec_size, default_value=0.0,
                 max_batch_size=1 << 30):
        self.max_batch = max_batch_size  # longer key lists run as that many separate lookups
        self.tier = TieredOracle(host_table_capacity, vec_size, -(-device_table_capacity // 64))
        self.row = {}
        self.default = np.float32(default_value)
        self.capacity = host_table_capacity

    def _rows(self, keys, insert):
        out = np.empty(len(keys), np.int64)
        for i, k in enumerate(keys):
            k = int(k)
            if k not in self.row and insert and len(self.row) < self.capacity:
                self.row[k] = len(self.row)
            out[i] = self.row.get(k, -1)
        return out

    def add(self, keys, vectors):
        last = {int(k): i for i, k in enumerate(keys)}
        pick = [i for i, k in enumerate(keys) if last[int(k)] == i]
        rows = self._rows([keys[i] for i in pick], True)
        self.tier.scatter(rows, np.asarray(vectors, np.float32)[pick], add=False)

    def _lookup(self, keys, insert):
        outs, rws, nmiss = [], [], 0
        for b in range(0, len(keys), self.max_batch):
            rows = self._rows(keys[b:b + self.max_batch], insert)
            out, nmiss = self.tier.lookup(rows)  # (the count of the last piece, as the device reports)
            out[rows < 0] = self.default
            outs.append(out)
            rws.append(rows)
        return np.concatenate(outs), np.concatenate(rws), nmiss

    def query(self, keys):
        return self._lookup(keys, False)[0]

    def lookup(self, keys):
        return self._lookup(keys, True)
