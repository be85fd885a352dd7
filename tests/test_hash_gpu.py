"""GPU parity: hash / index stage, bit-exact against the sequential CPU restatement."""
import ctypes
import os

import numpy as np
import pytest

pytestmark = pytest.mark.gpu


def _mk(torch, arr, dtype):
    return torch.from_numpy(np.ascontiguousarray(arr)).to("cuda").to(dtype)


class GpuHT:
    def __init__(self, capacity, key_type):
        from hugectr_amd import _lib
        self._lib = _lib
        self.h = ctypes.c_void_p()
        _lib.check(_lib.lib.hctr_ht_create(capacity, key_type, ctypes.byref(self.h)))

    def __del__(self):
        if self.h:
            self._lib.lib.hctr_ht_destroy(self.h)
            self.h = None

    def get_insert(self, keys_t):
        import torch
        out = torch.empty(keys_t.numel(), dtype=torch.int64, device="cuda")
        self._lib.check(self._lib.lib.hctr_ht_get_insert(self.h, self._lib.ptr(keys_t), keys_t.numel(), None,
                                                         self._lib.ptr(out), self._lib.stream_ptr()))
        torch.cuda.synchronize()
        return out.cpu().numpy().view(np.uint64)

    def get_mark(self, keys_t):
        import torch
        out = torch.empty(keys_t.numel(), dtype=torch.int64, device="cuda")
        self._lib.check(self._lib.lib.hctr_ht_get_mark(self.h, self._lib.ptr(keys_t), keys_t.numel(), None,
                                                       self._lib.ptr(out), self._lib.stream_ptr()))
        torch.cuda.synchronize()
        return out.cpu().numpy().view(np.uint64)

    def size(self):
        n = ctypes.c_size_t()
        self._lib.check(self._lib.lib.hctr_ht_size(self.h, self._lib.stream_ptr(), ctypes.byref(n)))
        return n.value

    def value_head(self):
        n = ctypes.c_size_t()
        self._lib.check(self._lib.lib.hctr_ht_value_head(self.h, self._lib.stream_ptr(), ctypes.byref(n)))
        return n.value

    def table_size(self):
        return self._lib.lib.hctr_ht_table_size(self.h)

    def dump(self):
        import torch
        n = self.table_size()
        k = torch.empty(n, dtype=torch.int64, device="cuda")
        v = torch.empty(n, dtype=torch.int64, device="cuda")
        c = ctypes.c_size_t()
        self._lib.check(self._lib.lib.hctr_ht_dump(self.h, self._lib.ptr(k), self._lib.ptr(v), ctypes.byref(c),
                                                   self._lib.stream_ptr()))
        return k[:c.value].cpu().numpy(), v[:c.value].cpu().numpy().view(np.uint64)


@pytest.mark.parametrize("key_bytes", [4, 8])
def test_murmur_hash_bit_exact(oracle, key_bytes):
    import torch
    from hugectr_amd import _lib
    rng = np.random.default_rng(0)
    if key_bytes == 4:
        keys = rng.integers(0, 2**32 - 1, size=5000, dtype=np.uint64).astype(np.int64)
        kt, ktype = _mk(torch, keys.astype(np.uint32).view(np.int32), torch.int32), _lib.KEY_U32
    else:
        keys = rng.integers(-2**62, 2**62, size=5000, dtype=np.int64)
        kt, ktype = _mk(torch, keys, torch.int64), _lib.KEY_I64
    out = torch.empty(keys.size, dtype=torch.int32, device="cuda")
    _lib.check(_lib.lib.hctr_hash_keys(_lib.ptr(kt), ktype, keys.size, _lib.ptr(out), _lib.stream_ptr()))
    got = out.cpu().numpy().view(np.uint32)
    assert (got == oracle.hash_keys(keys, key_bytes)).all()
    # ... and the reference's own functor, compiled from its header (oracle/_ref, built where the
    # reference is mounted; it travels with the repository snapshot)
    ref = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "oracle", "_ref",
                       "libref_hash.so")
    if os.path.exists(ref):
        import ctypes
        L = ctypes.CDLL(ref)
        want = np.empty(keys.size, np.uint32)
        if key_bytes == 4:
            k = keys.astype(np.uint32)
            L.ref_murmur3_u32_many(ctypes.c_void_p(k.ctypes.data), ctypes.c_size_t(k.size),
                                   ctypes.c_void_p(want.ctypes.data))
        else:
            L.ref_murmur3_i64_many(ctypes.c_void_p(keys.ctypes.data), ctypes.c_size_t(keys.size),
                                   ctypes.c_void_p(want.ctypes.data))
        assert (got == want).all()
    # published vector: 4 zero bytes, seed 0
    if key_bytes == 4:
        z = torch.zeros(1, dtype=torch.int32, device="cuda")
        _lib.check(_lib.lib.hctr_hash_keys(_lib.ptr(z), ktype, 1, _lib.ptr(out), _lib.stream_ptr()))
        assert int(out[:1].cpu().numpy().view(np.uint32)[0]) == 0x2362F9DE


@pytest.mark.parametrize("key_bytes,capacity,vocab,n", [
    (8, 5000, 3000, 20000),    # heavy duplication, many batches
    (4, 5000, 4000, 20000),
    (8, 300, 290, 5000),       # nearly full table: long probe chains
    (8, 200000, 150000, 300000),  # multi-tile compaction
])
def test_get_insert_matches_sequential_oracle(oracle, key_bytes, capacity, vocab, n):
    import torch
    from hugectr_amd import _lib
    rng = np.random.default_rng(42)
    ht_o = oracle.HashTable(capacity, key_bytes)
    ht_g = GpuHT(capacity, _lib.KEY_U32 if key_bytes == 4 else _lib.KEY_I64)
    assert ht_g.table_size() == ht_o.table_size()
    for batch in range(4):
        # power-law-ish duplicates + a spread of big keys
        keys = (rng.zipf(1.3, size=n) % vocab).astype(np.int64) * 7919 % (2**31)
        kt = _mk(torch, keys.astype(np.uint32).view(np.int32) if key_bytes == 4 else keys,
                 torch.int32 if key_bytes == 4 else torch.int64)
        got = ht_g.get_insert(kt)
        want = ht_o.get_insert(keys)
        assert (got == want).all(), f"batch {batch}: first-occurrence row assignment differs"
        assert ht_g.value_head() == ht_o.value_head()
    assert ht_g.size() == ht_o.size()
    # same key -> value map (physical slot order may differ under concurrent probing)
    gk, gv = ht_g.dump()
    ok, ov = ht_o.dump()
    assert dict(zip(gk.tolist(), gv.tolist())) == dict(zip(ok.tolist(), ov.tolist()))
    # get_mark: hits return the row, misses SIZE_MAX, nothing is inserted
    probe = np.concatenate([keys[:100], np.arange(10**9, 10**9 + 50)]).astype(np.int64)
    pt = _mk(torch, probe.astype(np.uint32).view(np.int32) if key_bytes == 4 else probe,
             torch.int32 if key_bytes == 4 else torch.int64)
    assert (ht_g.get_mark(pt) == ht_o.get_mark(probe)).all()
    assert ht_g.size() == ht_o.size()


def test_get_insert_empty_and_single(oracle):
    import torch
    from hugectr_amd import _lib
    ht = GpuHT(16, _lib.KEY_I64)
    e = torch.empty(0, dtype=torch.int64, device="cuda")
    assert ht.get_insert(e).size == 0
    one = torch.tensor([77], dtype=torch.int64, device="cuda")
    assert ht.get_insert(one).tolist() == [0]
    assert ht.get_insert(one).tolist() == [0]
    assert ht.value_head() == 1


@pytest.mark.parametrize("n,distinct_new", [(100_000, 40), (100_000, 5_000), (3_000, 700)])
def test_unseen_and_erased_keys_repeated_many_times_inside_one_batch(n, distinct_new):
    """The unseen keys' protocol under contention: a batch in which FEW unseen keys fill most
    positions (the thread that claims a key's slot and the many other occurrences of the key meet
    within nanoseconds: claim, store, atomic min and the mins deferred to the finish kernel all
    occur), next to erased keys -- entries whose key is in the table with "no row" as a dynamic
    table's remove() leaves them, which nobody claims -- and known keys.  Rows must be what a
    sequential insert in position order hands out (first occurrence of an unseen OR erased key takes
    the next row)."""
    import torch
    from hugectr_amd import _lib
    rng = np.random.default_rng(int(os.environ.get("HCTR_TEST_SEED", "0")) + 11)
    cap = 60_000
    ht = GpuHT(cap, _lib.KEY_I64)
    known = (rng.permutation(40_000)[:8_000].astype(np.int64) * 7919 + 3)
    r0 = ht.get_insert(_mk(torch, known, torch.int64))
    assert (r0 == np.arange(known.size, dtype=np.uint64)).all()
    erased = np.arange(1, 301, dtype=np.int64) * 1_000_003 + 10**12
    ev = torch.full((erased.size,), -1, dtype=torch.int64, device="cuda")  # SIZE_MAX: "no row"
    et = _mk(torch, erased, torch.int64)
    _lib.check(_lib.lib.hctr_ht_insert(ht.h, _lib.ptr(et), _lib.ptr(ev), erased.size, _lib.stream_ptr()))
    torch.cuda.synchronize()
    state = {int(k): i for i, k in enumerate(known.tolist())}
    head = known.size
    for batch in range(3):
        fresh = rng.integers(1, 2**40, size=distinct_new).astype(np.int64) + 10**13 * (batch + 1)
        # most positions: the few fresh keys; the rest: erased and known keys
        pick = rng.random(n)
        keys = np.where(pick < 0.7, fresh[rng.integers(0, fresh.size, n)],
                        np.where(pick < 0.85, erased[rng.integers(0, erased.size, n)],
                                 known[rng.integers(0, known.size, n)]))
        want = np.empty(n, dtype=np.uint64)
        for i, k in enumerate(keys.tolist()):
            r = state.get(k)
            if r is None:
                r = state[k] = head
                head += 1
            want[i] = r
        got = ht.get_insert(_mk(torch, keys, torch.int64))
        assert (got == want).all(), f"batch {batch}: {int((got != want).sum())} positions differ"
        assert ht.value_head() == head
    # everything is a plain hit now
    allk = np.array(list(state.keys()), dtype=np.int64)
    assert (ht.get_mark(_mk(torch, allk, torch.int64)) ==
            np.array(list(state.values()), dtype=np.uint64)).all()


def test_get_insert_mostly_unseen_batches_at_the_bench_batch_size(oracle):
    """A first epoch at the bench's batch size: 1.7 M keys per call, most of them unseen, a fifth
    of the positions repeating a key of the same call -- from the second call on the probe kernel
    takes its store form (the previous call inserted more than an eighth of a batch: claimers store,
    later occurrences lower with an atomic min or defer it to the finish kernel, two grid barriers).
    Rows must be the sequential oracle's, bit for bit, in every call."""
    import torch
    from hugectr_amd import _lib
    rng = np.random.default_rng(int(os.environ.get("HCTR_TEST_SEED", "0")) + 5)
    cap, n = 6_000_000, 1_703_936
    ht_o = oracle.HashTable(cap, 8)
    ht_g = GpuHT(cap, _lib.KEY_I64)
    for batch in range(3):
        keys = (rng.integers(0, 3_000_000, size=n).astype(np.int64) * 2_654_435_761) % (1 << 45)
        got = ht_g.get_insert(_mk(torch, keys, torch.int64))
        want = ht_o.get_insert(keys)
        assert (got == want).all(), f"call {batch}: {int((got != want).sum())} rows differ"
        assert ht_g.value_head() == ht_o.value_head()
    assert ht_g.size() == ht_o.size()


def test_overflow_is_flagged():
    """more distinct keys than max_vocabulary_size_per_gpu -> check_overflow raises
    (R/HugeCTR/include/embeddings/localized_slot_sparse_embedding_hash.hpp:552-569)"""
    import torch
    import hugectr_amd as ha
    from hugectr_amd import _lib
    emb = ha.SparseEmbeddingHash(_lib.EMB_LOCALIZED, 4, 0, 8, 4, 4, 4, 0, ha.OptParams())
    ro = torch.arange(0, 17, dtype=torch.int64, device="cuda")
    keys = torch.arange(0, 16, dtype=torch.int64, device="cuda")
    emb.forward(True, ro, keys)
    with pytest.raises(ha.HugeCTRAmdError):
        emb.check_overflow()


@pytest.mark.parametrize("D", [16, 11])
def test_overflow_keys_get_no_row_and_nothing_is_touched_out_of_bounds(D):
    """A table with 8 rows that meets 40 distinct keys: 8 of them get the rows 0..7 (which ones is
    decided by who claims one of the floor(8 / 0.75) = 10 physical slots first -- the reference's
    full table is a race too), every other key resolves to "no row" -- pooled as zeros, skipped by
    the update, absent from the dump -- the row counter stops at the capacity, and the overflow
    is reported by the blocking and by the polled check.  (Before: rows >= capacity were handed out
    and read / written past the table.)"""
    import numpy as np
    import torch
    import hugectr_amd as ha
    from hugectr_amd import _lib
    B, S, V = 10, 4, 8
    opt = ha.OptParams(optimizer=_lib.OPT_SGD, lr=1.0, atomic_update=False)
    emb = ha.SparseEmbeddingHash(_lib.EMB_LOCALIZED, B, B, V, D, S, S, 0, opt)
    emb.table().fill_(1.0)
    ro = torch.arange(0, B * S + 1, dtype=torch.int64, device="cuda")
    keys = torch.arange(100, 100 + B * S, dtype=torch.int64, device="cuda")
    for it in range(2):  # second pass: the overflowed keys try again and still get nothing
        out = emb.forward(True, ro, keys)
        torch.cuda.synchronize()
        vi = emb.value_index(B * S).cpu().numpy().view(np.uint64)
        live = vi != np.uint64(0xFFFFFFFFFFFFFFFF)
        assert live.sum() == V and sorted(vi[live].tolist()) == list(range(V))
        if it == 0:
            first = vi.copy()
        assert (vi == first).all(), "a key changed its row"
        o = out.float().cpu().numpy().reshape(B * S, D)
        assert (o[live] == 1.0).all() and (o[~live] == 0.0).all()
        assert emb.get_vocabulary_size() == V
    g = torch.ones(B, S, D, device="cuda")
    emb.backward(g)
    emb.update_params()
    torch.cuda.synchronize()
    assert (emb.table().cpu().numpy() == 0.0).all()  # 1 - lr * 1 on the 8 live rows, no more
    k, sid, vec = emb.dump_parameters()
    assert sorted(k.cpu().tolist()) == sorted((100 + np.nonzero(live)[0]).tolist())
    with pytest.raises(ha.HugeCTRAmdError):
        emb.check_overflow()
    with pytest.raises(ha.HugeCTRAmdError):
        for _ in range(3):  # the polled form may lag by two calls
            emb.poll_overflow()
            torch.cuda.synchronize()
    # eval on the same keys: get_mark of a key that holds no row is a miss
    out = emb.forward(False, ro, keys).float().cpu().numpy().reshape(B * S, D)
    assert (out == 0.0).all()  # live rows were stepped to 1 - lr * 1 = 0, the rest are misses


@pytest.mark.parametrize("key_bytes", [4, 8])
def test_get_insert_many_batches_of_every_shape(oracle, key_bytes):
    """The two-launch index stage over a long, irregular sequence on ONE table: batches of 1 to
    300 000 keys (one workgroup / the full cooperative grid of the finish kernel), none / a few /
    all keys unseen (the finish kernel's early exit alternating with its barrier path, so its
    double-buffered masks and region counts change hands at every possible moment), keys repeated
    hundreds of times inside a batch, get_mark in between, the live-count form (d_n < n) -- rows
    bit-equal to the sequential oracle after every call."""
    import torch
    from hugectr_amd import _lib
    rng = np.random.default_rng(17 + key_bytes)
    cap = 1_500_000
    ktype = _lib.KEY_U32 if key_bytes == 4 else _lib.KEY_I64
    tdt = torch.int32 if key_bytes == 4 else torch.int64
    g, ref = GpuHT(cap, ktype), oracle.HashTable(cap, key_bytes)
    universe = rng.permutation(6_000_000)[:2_500_000].astype(np.int64) * (3 if key_bytes == 8 else 1)
    seen = 0
    sizes = [1, 63, 64, 65, 4095, 4096, 4097, 300_000, 5, 131_072, 200_000, 1000, 257, 65_536,
             250_000, 31, 100_000, 8191, 3, 150_000]
    for it, n in enumerate(sizes * 2):
        kind = it % 4
        if kind == 0:      # all unseen
            new = n
        elif kind == 1:    # none unseen (steady state: the finish kernel exits at once)
            new = 0
        elif kind == 2:    # a few per cent unseen
            new = max(1, n // 25)
        else:              # half unseen, heavy repetition inside the batch
            new = n // 2
        new = min(new, universe.size - seen) if seen < universe.size else 0
        if seen == 0 and new == 0:
            new = n
        fresh = universe[seen:seen + new]
        seen += new
        old = universe[rng.integers(0, max(seen - new, 1), size=n - new)] if seen - new > 0 \
            else np.repeat(fresh[:1], n - new)
        keys = np.concatenate([fresh, old])
        if kind == 3 and n > 8:  # hundreds of copies of a handful of keys, new ones included
            hot = keys[rng.integers(0, keys.size, size=6)]
            keys[rng.integers(0, n, size=n // 3)] = hot[rng.integers(0, 6, size=n // 3)]
        rng.shuffle(keys)
        kt = _mk(torch, keys if key_bytes == 8 else keys.astype(np.uint32).view(np.int32), tdt)
        got = g.get_insert(kt)
        want = ref.get_insert(keys)
        if not (got == want).all():
            bad = np.nonzero(got != want)[0]
            first_seen = {int(k): j for j, k in enumerate(universe[:seen])}
            raise AssertionError((it, n, kind, bad.size, [
                (int(keys[b]), int(got[b]), int(want[b]), first_seen.get(int(keys[b])))
                for b in bad[:6]]))
        if it % 5 == 4:  # eval in between: misses read as invalid, nothing is inserted
            probe = np.concatenate([universe[:min(seen, 1000)], universe[-500:]])
            pt = _mk(torch, probe if key_bytes == 8 else probe.astype(np.uint32).view(np.int32), tdt)
            assert (g.get_mark(pt) == ref.get_mark(probe)).all()
            assert g.value_head() == ref.size()
    # live count on the device smaller than the host bound: positions beyond it are not touched
    n, live = 50_000, 31_337
    keys = universe[rng.integers(0, universe.size, size=n)]
    kt = _mk(torch, keys if key_bytes == 8 else keys.astype(np.uint32).view(np.int32), tdt)
    d_n = torch.tensor([live], dtype=torch.int64, device="cuda")
    out = torch.full((n,), -7, dtype=torch.int64, device="cuda")
    _lib.check(_lib.lib.hctr_ht_get_insert(g.h, _lib.ptr(kt), n, _lib.ptr(d_n), _lib.ptr(out),
                                           _lib.stream_ptr()))
    torch.cuda.synchronize()
    res = out.cpu().numpy()
    assert (res[:live].view(np.uint64) == ref.get_insert(keys[:live])).all()
    assert (res[live:] == -7).all()
    assert g.size() == ref.size()


def test_a_grid_barrier_that_gives_up_is_all_or_nothing_sticky_and_recoverable(oracle, monkeypatch):
    """HCTR_HT_SPIN_LIMIT=0: a workgroup of the cooperative finish kernel that has to wait at the
    grid barrier gives up at its first poll.  Then (1) NO unseen key of the batch has a row (the
    known keys keep theirs), error bit 2 (value 4) is set; (2) the next inserting batch fails the
    same way instead of reading the slots the first one left pending as its own; (3) after
    hctr_ht_recover over the keys of both batches the same two batches resolve to exactly the rows
    of the sequential oracle (the row counter never moved)."""
    import torch
    from hugectr_amd import _lib
    rng = np.random.default_rng(7)
    cap = 1 << 17
    g, ref = GpuHT(cap, _lib.KEY_I64), oracle.HashTable(cap, 8)
    warm = rng.integers(0, 1 << 40, size=3000).astype(np.int64)
    assert (g.get_insert(_mk(torch, warm, torch.int64)) == ref.get_insert(warm)).all()
    # > 4096 positions per finish workgroup: 40 000 keys -> 10 workgroups at the barrier
    b1 = np.concatenate([rng.integers(0, 1 << 40, size=37000), warm[:3000]]).astype(np.int64)
    rng.shuffle(b1)
    b2 = np.concatenate([rng.integers(0, 1 << 40, size=9000), b1[:1000]]).astype(np.int64)

    def flags():
        e = ctypes.c_uint32()
        _lib.check(_lib.lib.hctr_ht_error_flags(g.h, _lib.stream_ptr(), ctypes.byref(e)))
        return e.value

    head0 = g.value_head()
    monkeypatch.setenv("HCTR_HT_SPIN_LIMIT", "0")
    r1 = g.get_insert(_mk(torch, b1, torch.int64))
    assert flags() == 4
    known = np.isin(b1, warm)
    want_known = ref.get_mark(b1)  # (find only: the oracle's table is not touched)
    assert (r1[known] == want_known[known]).all()
    assert (r1[~known] == np.uint64(0xFFFFFFFFFFFFFFFF)).all(), "a position got a row from a barrier that gave up"
    r2 = g.get_insert(_mk(torch, b2, torch.int64))
    assert flags() == 4 and g.value_head() == head0
    k2 = np.isin(b2, warm)
    assert (r2[~k2] == np.uint64(0xFFFFFFFFFFFFFFFF)).all()
    monkeypatch.delenv("HCTR_HT_SPIN_LIMIT")
    both = _mk(torch, np.concatenate([b1, b2]), torch.int64)
    _lib.check(_lib.lib.hctr_ht_recover(g.h, _lib.ptr(both), both.numel(), _lib.stream_ptr()))
    assert flags() == 0 and g.size() == ref.size()
    assert (g.get_insert(_mk(torch, b1, torch.int64)) == ref.get_insert(b1)).all()
    assert (g.get_insert(_mk(torch, b2, torch.int64)) == ref.get_insert(b2)).all()
    assert flags() == 0 and g.size() == ref.size() and g.value_head() == ref.value_head()
