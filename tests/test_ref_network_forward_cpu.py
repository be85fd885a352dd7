"""The REFERENCE's NetworkForward of a model-parallel embedding_collection group with ITS OWN address
lambdas (R/HugeCTR/embedding/operators/network_forward.cu:272-321 batch-major, :353-404
feature-major: which partial vectors of the received buffers add up to an output vector, the
Average divisor, the output address) over the network layout ITS OWN host code produces
(NetworkIndices::init, :23-62) and generic_lookup.cuh's kernels -- all cut out of the checkout
(oracle/_ref/libref_network_forward.so, libref_network_indices.so) and executed by the host
interpreter of tests/emu -- next to this repo's hctr_ebc_network_forward (csrc/ebc.hip, the
kernel source stepped through by the same interpreter) on the same received buffers, for random
shardings, both output layouts, fp32 and binary16 vectors, Sum and Average lookups: the outputs
agree BIT FOR BIT (the shards of a lookup are added in the same order: the reference's list is a
sort of at most 16 entries here, which leaves equal lookups in rank order)."""
import ctypes
import os
import sys

import numpy as np
import pytest

HERE = os.path.dirname(os.path.abspath(__file__))
REF = os.path.join(HERE, "..", "oracle", "_ref")
pytestmark = pytest.mark.skipif(
    not (os.path.exists(os.path.join(REF, "libref_network_forward.so")) and
         os.path.exists(os.path.join(REF, "libref_network_indices.so"))),
    reason="oracle/_ref not built (needs the reference checkout)")
sys.path.insert(0, os.path.join(HERE, "emu"))
import emu  # noqa: E402

P, I = ctypes.c_void_p, ctypes.c_int


def _p(a):
    return a.ctypes.data_as(P)


@pytest.fixture(scope="module")
def libs():
    ni = ctypes.CDLL(os.path.join(REF, "libref_network_indices.so"))
    ni.refnet_indices.argtypes = [I, P, P, P, P, P, P]
    nf = ctypes.CDLL(os.path.join(REF, "libref_network_forward.so"))
    nf.refnet_forward.argtypes = [I, I, I, I, I, I, P, P, P, P, P, P, P, P, P, P, P]
    return ni, nf, emu.load_under_test()


@pytest.mark.parametrize("seed", range(12))
def test_network_forward_equals_the_reference_operator(libs, seed):
    from hugectr_amd import _lib
    ni, nf, elib = libs
    rng = np.random.default_rng(seed)
    half = seed % 2
    batch_major = (seed // 2) % 2
    dt = np.float16 if half else np.float32
    world = int(rng.choice([1, 2, 3, 4]))
    T = int(rng.integers(1, 4))
    L = int(rng.integers(T, T + 2))
    lookup_table = list(range(T)) + [int(rng.integers(0, T)) for _ in range(L - T)]
    ev = int(rng.choice([4, 6, 16, 32, 128, 200]))
    bpg = int(rng.choice([1, 3, 8]))
    owners = []
    for t in range(T):  # every table on one to three ranks, in ascending rank order
        k = int(rng.integers(1, min(world, 3) + 1))
        owners.append(sorted(int(g) for g in rng.choice(world, size=k, replace=False)))
    local = [[l for l in range(L) if r in owners[lookup_table[l]]] for r in range(world)]
    total = sum(len(x) for x in local)
    assert total <= 16
    combiner = rng.integers(0, 2, size=L).astype(np.int32)  # 0 Sum, 1 Average
    counts = rng.integers(0, 5, size=L * bpg).astype(np.int64)  # keys of (lookup, local sample)
    # ---- the reference's layout of the network buffer ---------------------------------------------
    offs = np.concatenate([[0], np.cumsum([len(x) for x in local])]).astype(np.int32)
    flat = np.array([l for x in local for l in x] + [0], dtype=np.int32)
    ids, gpus = np.full(total, -1, np.int32), np.full(total, -1, np.int32)
    noff, dst = np.full(L + 1, -1, np.int32), np.full(L, -1, np.int32)
    n_dst = ni.refnet_indices(world, _p(offs), _p(flat), _p(ids), _p(gpus), _p(noff), _p(dst))
    assert n_dst == L
    # ---- received buffers: from rank g its local lookups' [bpg][ev] blocks back to back -----------
    bufs = [(rng.standard_normal((max(len(local[g]), 1), bpg, ev)) * 0.5).astype(dt) for g in range(world)]
    ev_sizes = [np.full(max(len(local[g]), 1), ev, np.int32) for g in range(world)]
    ev_offs = [(np.arange(max(len(local[g]), 1) + 1) * ev).astype(np.int32) for g in range(world)]
    arr = lambda xs: (P * len(xs))(*[x.ctypes.data for x in xs])  # noqa: E731
    dst_start = (np.arange(L + 1) * ev).astype(np.int32)
    comb_c = combiner.astype(np.int8)
    want = np.full(bpg * L * ev, np.nan, dt)
    nf.refnet_forward(batch_major, half, bpg, L, n_dst, ev, _p(counts), _p(ids), _p(gpus), _p(noff), _p(dst),
                      arr(ev_sizes), arr(ev_offs), arr(bufs), _p(dst_start), _p(comb_c), _p(want))
    assert not np.isnan(want.astype(np.float32)).any()
    # ---- this repo: the block table of embedding_collection.py and the HIP source ------------------
    base = np.concatenate([[0], np.cumsum([len(x) for x in local])])
    max_shards = max(len(o) for o in owners)
    src = np.full(L * max_shards, -1, np.int32)
    for l in range(L):
        for s, r in enumerate(owners[lookup_table[l]]):
            src[l * max_shards + s] = base[r] + local[r].index(l)
    recv = np.concatenate([bufs[g][:len(local[g])].reshape(-1) for g in range(world)] + [np.zeros(0, dt)])
    got = np.full(bpg * L * ev, np.nan, dt)
    emu.check(elib, elib.hctr_ebc_network_forward(bpg, L, ev, max_shards, _p(src), _p(combiner), _p(counts),
                                                  batch_major, _p(recv), _p(got), _lib.F16 if half else _lib.F32,
                                                  None))
    np.testing.assert_array_equal(got.view(np.uint16 if half else np.uint32),
                                  want.view(np.uint16 if half else np.uint32))


@pytest.mark.parametrize("seed", range(12))
def test_model_forward_equals_the_reference_operator(libs, seed):
    """ModelForward::sparse_forward with its own lambdas (model_forward.cu:182-206): buckets ordered
    (local lookup, GLOBAL sample), rows reached through the float** of ILookup::lookup, sums written
    to the buffer of the GPU that owns the sample.  Here the routing pass orders the buckets
    (destination GPU, local lookup, local sample) so that the pooled vectors ARE the send buffer:
    hctr_forward_pool_ptrs (dynamic tables) and hctr_forward_pool (row indices into a flat table) on
    the re-ordered buckets give the reference's per-GPU buffers bit for bit."""
    from hugectr_amd import _lib
    _, nf, elib = libs
    nf.refmodel_forward.argtypes = [I, I, I, I, I, P, P, P, P, P]
    rng = np.random.default_rng(100 + seed)
    half = seed % 2
    dt = np.float16 if half else np.float32
    world = int(rng.choice([1, 2, 4]))
    bpg = int(rng.choice([1, 2, 5]))
    B = world * bpg
    nl = int(rng.integers(1, 4))  # local lookups
    ev = int(rng.choice([4, 6, 16, 32, 128, 200]))
    rows = int(rng.choice([3, 50, 400]))
    table = (rng.standard_normal((rows, ev)) * 0.5).astype(np.float32)
    max_hot = int(rng.choice([1, 3, 9, 40]))
    lens = rng.integers(0, max_hot + 1, size=nl * B).astype(np.int64)  # bucket (l, global b)
    lens[rng.random(nl * B) < 0.2] = 0
    br = np.zeros(nl * B + 1, np.int64)
    np.cumsum(lens, out=br[1:])
    idx = rng.integers(0, rows, size=int(br[-1])).astype(np.int64)
    # ---- reference -----------------------------------------------------------------------------------
    ptrs = np.array([table.ctypes.data + int(r) * ev * 4 for r in idx] + [0], dtype=np.uint64)
    ev_size = np.full(nl, ev, np.int32)
    ev_start = (np.arange(nl + 1) * ev).astype(np.int32)
    bufs = [np.full(nl * bpg * ev, np.nan, dt) for _ in range(world)]
    arr = (P * world)(*[b.ctypes.data for b in bufs])
    nf.refmodel_forward(half, B, bpg, nl, ev, _p(br), _p(ev_size), _p(ev_start), _p(ptrs), arr)
    want = np.concatenate(bufs)
    assert not np.isnan(want.astype(np.float32)).any()
    # ---- this repo: buckets re-ordered (peer, lookup, local sample) ------------------------------------
    order = [l * B + p * bpg + b for p in range(world) for l in range(nl) for b in range(bpg)]
    lens2 = lens[order]
    br2 = np.zeros(nl * B + 1, np.int64)
    np.cumsum(lens2, out=br2[1:])
    idx2 = np.concatenate([idx[br[o]:br[o + 1]] for o in order] + [np.zeros(0, np.int64)]).astype(np.int64)
    ptrs2 = np.array([table.ctypes.data + int(r) * ev * 4 for r in idx2] + [0], dtype=np.uint64)
    code = _lib.F16 if half else _lib.F32
    got = np.full(world * nl * bpg * ev, np.nan, dt)
    emu.check(elib, elib.hctr_forward_pool_ptrs(nl * B, ev, 0, _p(br2), _p(ptrs2), _p(got), code, None))
    view = np.uint16 if half else np.uint32
    np.testing.assert_array_equal(got.view(view), want.view(view))
    got2 = np.full(world * nl * bpg * ev, np.nan, dt)
    idx2u = np.concatenate([idx2, np.zeros(1, np.int64)]).astype(np.uint64)
    emu.check(elib, elib.hctr_forward_pool(nl * B, ev, 0, _p(br2), _lib.KEY_I64, _p(idx2u), _p(table), _p(got2),
                                           code, None))
    np.testing.assert_array_equal(got2.view(view), want.view(view))


@pytest.mark.parametrize("seed", range(12))
def test_network_backward_equals_the_reference_operator(libs, seed):
    """NetworkBackward with its own lambdas (network_backward.cu:55-98, :132-174): the gradient of
    every output vector, divided by the bucket's key count for Average lookups, written to the
    block of each shard of its lookup -- next to hctr_ebc_network_backward (HIP source)."""
    from hugectr_amd import _lib
    ni, nf, elib = libs
    nf.refnet_backward.argtypes = [I, I, I, I, I, P, P, P, P, P, P, P, P, P, P, P]
    rng = np.random.default_rng(200 + seed)
    half = seed % 2
    batch_major = (seed // 2) % 2
    dt = np.float16 if half else np.float32
    world = int(rng.choice([1, 2, 3, 4]))
    T = int(rng.integers(1, 4))
    L = int(rng.integers(T, T + 2))
    lookup_table = list(range(T)) + [int(rng.integers(0, T)) for _ in range(L - T)]
    ev = int(rng.choice([4, 6, 16, 32, 128, 200]))
    bpg = int(rng.choice([1, 3, 8]))
    owners = []
    for t in range(T):
        k = int(rng.integers(1, min(world, 3) + 1))
        owners.append(sorted(int(g) for g in rng.choice(world, size=k, replace=False)))
    local = [[l for l in range(L) if r in owners[lookup_table[l]]] for r in range(world)]
    total = sum(len(x) for x in local)
    combiner = rng.integers(0, 2, size=L).astype(np.int32)
    counts = rng.integers(0, 5, size=L * bpg).astype(np.int64)
    offs = np.concatenate([[0], np.cumsum([len(x) for x in local])]).astype(np.int32)
    flat = np.array([l for x in local for l in x] + [0], dtype=np.int32)
    ids, gpus = np.full(total, -1, np.int32), np.full(total, -1, np.int32)
    noff, dst = np.full(L + 1, -1, np.int32), np.full(L, -1, np.int32)
    assert ni.refnet_indices(world, _p(offs), _p(flat), _p(ids), _p(gpus), _p(noff), _p(dst)) == L
    grad = (rng.standard_normal(bpg * L * ev) * 0.5).astype(dt)  # [b][l][ev] or [l][b][ev]
    bufs = [np.full(max(len(local[g]), 1) * bpg * ev, np.nan, dt) for g in range(world)]
    ev_sizes = [np.full(max(len(local[g]), 1), ev, np.int32) for g in range(world)]
    ev_offs = [(np.arange(max(len(local[g]), 1) + 1) * ev).astype(np.int32) for g in range(world)]
    arr = lambda xs: (P * len(xs))(*[x.ctypes.data for x in xs])  # noqa: E731
    dst_start = (np.arange(L + 1) * ev).astype(np.int32)
    comb_c = combiner.astype(np.int8)
    nf.refnet_backward(batch_major, half, bpg, L, ev, _p(counts), _p(ids), _p(gpus), _p(noff), _p(dst),
                       arr(ev_sizes), arr(ev_offs), _p(dst_start), _p(grad), arr(bufs), _p(comb_c))
    want = np.concatenate([bufs[g][:len(local[g]) * bpg * ev] for g in range(world)] + [np.zeros(0, dt)])
    assert not np.isnan(want.astype(np.float32)).any()
    base = np.concatenate([[0], np.cumsum([len(x) for x in local])])
    max_shards = max(len(o) for o in owners)
    src = np.full(L * max_shards, -1, np.int32)
    for l in range(L):
        for s, r in enumerate(owners[lookup_table[l]]):
            src[l * max_shards + s] = base[r] + local[r].index(l)
    got = np.full(total * bpg * ev, np.nan, dt)
    emu.check(elib, elib.hctr_ebc_network_backward(bpg, L, ev, max_shards, _p(src), _p(combiner), _p(counts),
                                                   batch_major, _p(grad), _p(got), _lib.F16 if half else _lib.F32,
                                                   None))
    view = np.uint16 if half else np.uint32
    np.testing.assert_array_equal(got.view(view), want.view(view))
