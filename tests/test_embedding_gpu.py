"""GPU parity: sparse embedding forward / backward / update through the C ABI vs the CPU oracle.
Forward and index stage: bit-exact.  Optimizer state/weights: rel 1e-5 (north star: 1e-3)."""

import numpy as np
import pytest

from util import assert_close, make_csr

pytestmark = pytest.mark.gpu


def _t(torch, a, dtype=None):
    t = torch.from_numpy(np.ascontiguousarray(a)).to("cuda")
    return t if dtype is None else t.to(dtype)


def _oracle_opt(oracle, opt, times):
    from hugectr_amd import _lib
    m = {_lib.OPT_ADAM: oracle.OPT_ADAM, _lib.OPT_ADAGRAD: oracle.OPT_ADAGRAD,
         _lib.OPT_MOMENTUM_SGD: oracle.OPT_MOMENTUM, _lib.OPT_NESTEROV: oracle.OPT_NESTEROV,
         _lib.OPT_SGD: oracle.OPT_SGD}
    o = oracle.OptParamsC()
    o.optimizer, o.update_type, o.lr = m[opt.optimizer], opt.update_type, opt.lr
    o.beta1, o.beta2, o.epsilon = opt.beta1, opt.beta2, opt.epsilon
    o.momentum_factor, o.scaler, o.times = opt.momentum_factor, opt.scaler, times
    return o


@pytest.mark.parametrize("D,combiner,one_hot,key_bytes", [
    (128, 0, True, 8),    # DLRM Criteo-1TB shape: one-hot, sum
    (16, 0, True, 4),     # DCN / DeepFM shape, u32 keys
    (16, 1, False, 8),    # ragged multi-hot with empty buckets, mean
    (64, 0, False, 8),
    (128, 1, False, 4),
    (11, 1, False, 8),    # samples/deepfm: embedding_vec_size 11 -> generic (non-vec4) path
    (256, 0, False, 8),
    (4, 1, False, 8),
])
def test_forward_bit_exact(oracle, D, combiner, one_hot, key_bytes):
    import torch
    import hugectr_amd as ha
    from hugectr_amd import _lib
    rng = np.random.default_rng(D * 10 + combiner)
    B, S, hot, vps = 96, 7, 5, 50
    ro, keys = make_csr(rng, B, S, hot, vps, one_hot=one_hot)
    kd = torch.int64 if key_bytes == 8 else torch.int32
    emb = ha.SparseEmbeddingHash(_lib.EMB_LOCALIZED, B, B, S * vps, D, S * hot, S, combiner,
                                 ha.OptParams(), key_dtype=torch.int64 if key_bytes == 8 else torch.uint32)
    emb.init_params()
    table = emb.table().cpu().numpy().copy()
    ht = oracle.HashTable(S * vps, key_bytes)
    for it in range(2):  # second pass: all keys known (steady state)
        out = emb.forward(True, _t(torch, ro, kd), _t(torch, keys, kd))
        torch.cuda.synchronize()
        vi = ht.get_insert(keys)
        got_vi = emb.value_index(keys.size).cpu().numpy().view(np.uint64)
        assert (got_vi == vi).all(), "row indices differ from the sequential oracle"
        want = oracle.forward(ro, vi, table, D, combiner)
        got = out.cpu().numpy().reshape(-1, D)
        assert (got.view(np.uint32) == want.view(np.uint32)).all(), "forward not bit-exact"
    # eval: unseen keys contribute 0 but still count in the mean
    ro_e, keys_e = make_csr(rng, B, S, hot, vps * 2, one_hot=one_hot)
    out = emb.forward(False, _t(torch, ro_e, kd), _t(torch, keys_e, kd))
    vi = ht.get_mark(keys_e)
    assert (vi == oracle.INVALID).any()
    want = oracle.forward(ro_e, vi, table, D, combiner)
    assert (out.cpu().numpy().reshape(-1, D).view(np.uint32) == want.view(np.uint32)).all()
    assert emb.get_vocabulary_size() == ht.size()


@pytest.mark.parametrize("D,combiner", [(128, 0), (16, 1)])
def test_one_hot_and_multi_hot_batches_alternate_on_one_handle(oracle, D, combiner):
    """The index stage keeps TWO one-hot flags and presets the next batch's from its finish kernel
    (no memset launch): a handle that meets one-hot, ragged multi-hot, one-hot, an evaluation batch
    and one-hot again must take the right gather loop every time -- rows and pooled vectors
    bit-equal to the oracle at every step, new keys arriving at every step."""
    import torch
    import hugectr_amd as ha
    from hugectr_amd import _lib
    rng = np.random.default_rng(5 + D)
    B, S, hot, vps = 96, 7, 4, 300
    emb = ha.SparseEmbeddingHash(_lib.EMB_LOCALIZED, B, B, S * vps, D, S * hot, S, combiner,
                                 ha.OptParams())
    emb.init_params()
    table = emb.table().cpu().numpy().copy()
    ht = oracle.HashTable(S * vps, 8)
    for step, (train, one_hot) in enumerate([(True, True), (True, False), (True, True),
                                             (False, False), (True, True), (True, False),
                                             (True, False), (False, True), (True, True)]):
        ro, keys = make_csr(rng, B, S, hot, vps, one_hot=one_hot)
        out = emb.forward(train, _t(torch, ro, torch.int64), _t(torch, keys, torch.int64))
        vi = ht.get_insert(keys) if train else ht.get_mark(keys)
        if train:
            got_vi = emb.value_index(keys.size).cpu().numpy().view(np.uint64)
            assert (got_vi == vi).all(), (step, "row indices differ from the sequential oracle")
        want = oracle.forward(ro, vi, table, D, combiner)
        got = out.cpu().numpy().reshape(-1, D)
        assert (got.view(np.uint32) == want.view(np.uint32)).all(), (step, train, one_hot)
    assert emb.get_vocabulary_size() == ht.size()
    emb.poll_overflow()


def test_forward_empty_batch_of_keys(oracle):
    import torch
    import hugectr_amd as ha
    from hugectr_amd import _lib
    B, S, D = 8, 3, 16
    emb = ha.SparseEmbeddingHash(_lib.EMB_LOCALIZED, B, 0, 100, D, 4, S, 1, ha.OptParams())
    ro = torch.zeros(B * S + 1, dtype=torch.int64, device="cuda")
    keys = torch.empty(0, dtype=torch.int64, device="cuda")
    out = emb.forward(True, ro, keys)
    assert out.shape == (B, S, D) and float(out.abs().max()) == 0.0


OPTS = [
    ("sgd", dict(optimizer=6, atomic_update=False)),
    ("adam_local", dict(optimizer=1, update_type=0)),
    ("adam_global", dict(optimizer=1, update_type=1)),
    ("adam_lazy", dict(optimizer=1, update_type=2)),
    ("adagrad", dict(optimizer=3)),
    ("momentum_local", dict(optimizer=5, update_type=0, momentum_factor=0.9)),
    ("momentum_global", dict(optimizer=5, update_type=1, momentum_factor=0.9)),
    ("nesterov_local", dict(optimizer=4, update_type=0, momentum_factor=0.9)),
    ("nesterov_global", dict(optimizer=4, update_type=1, momentum_factor=0.9)),
]


@pytest.mark.parametrize("name,kw", OPTS, ids=[o[0] for o in OPTS])
@pytest.mark.parametrize("D,combiner", [(16, 1), (128, 0)])
def test_train_steps_match_oracle(oracle, name, kw, D, combiner):
    """The reference's own test recipe (localized_slot_sparse_embedding_hash_test.cu:181-519):
    several train batches; compare forward, wgrad and the whole table after every update."""
    import torch
    import hugectr_amd as ha
    from hugectr_amd import _lib
    rng = np.random.default_rng(7)
    B, S, hot, vps = 64, 6, 4, 40
    V = S * vps + 16  # a few never-used padding rows (they matter for Global updates)
    opt = ha.OptParams(lr=0.05, scaler=4.0, **kw)
    emb = ha.SparseEmbeddingHash(_lib.EMB_LOCALIZED, B, 0, V, D, S * hot, S, combiner, opt)
    emb.init_params()
    torch.cuda.synchronize()
    table = emb.table().cpu().numpy().copy()
    ns = {1: 2, 3: 1, 5: 1, 4: 1, 6: 0}[opt.optimizer]
    s0 = np.zeros_like(table) if ns >= 1 else None
    s1 = np.zeros_like(table) if ns >= 2 else None
    pt = np.ones(table.shape, dtype=np.uint64) if name == "adam_lazy" else None
    ht = oracle.HashTable(V, 8)
    for it in range(4):
        ro, keys = make_csr(rng, B, S, hot, vps, one_hot=(combiner == 0 and it % 2 == 0))
        out = emb.forward(True, _t(torch, ro), _t(torch, keys))
        vi = ht.get_insert(keys)
        want = oracle.forward(ro, vi, table, D, combiner)
        assert_close(out.cpu().numpy().reshape(-1, D), want, 1e-5, 1e-6, f"{name} fwd it{it}")
        g = rng.standard_normal((B * S, D)).astype(np.float32)
        gt = _t(torch, g).view(B, S, D).contiguous()
        emb.backward(gt)
        wg = emb.get_wgrad().cpu().numpy().reshape(-1, D)
        want_wg = oracle.backward(ro, g, D, combiner)
        assert (wg.view(np.uint32) == want_wg.view(np.uint32)).all(), "wgrad not bit-exact"
        emb.update_params()
        torch.cuda.synchronize()
        oracle.update_params(ro, vi, want_wg, _oracle_opt(oracle, opt, it + 1), table, s0, s1, pt)
        assert_close(emb.table().cpu().numpy(), table, 1e-5, 1e-6, f"{name} table it{it}")
        if s0 is not None:
            assert_close(emb.opt_state(0).cpu().numpy(), s0, 1e-5, 1e-6, f"{name} state0 it{it}")
        if s1 is not None:
            assert_close(emb.opt_state(1).cpu().numpy(), s1, 1e-5, 1e-7, f"{name} state1 it{it}")


def test_sgd_atomic_update_matches_sorted_within_tolerance(oracle):
    """Python default atomic_update=True (optimizer_wrapper.hpp:40): fp32 atomicAdd, order not
    deterministic -> compare with tolerance."""
    import torch
    import hugectr_amd as ha
    from hugectr_amd import _lib
    rng = np.random.default_rng(3)
    B, S, D, vps = 128, 5, 32, 20
    opt = ha.OptParams(optimizer=_lib.OPT_SGD, lr=0.1, atomic_update=True, scaler=2.0)
    emb = ha.SparseEmbeddingHash(_lib.EMB_LOCALIZED, B, 0, S * vps, D, S * 3, S, 0, opt)
    emb.init_params()
    table = emb.table().cpu().numpy().copy()
    ht = oracle.HashTable(S * vps, 8)
    ro, keys = make_csr(rng, B, S, 3, vps)
    emb.forward(True, _t(torch, ro), _t(torch, keys))
    vi = ht.get_insert(keys)
    g = rng.standard_normal((B * S, D)).astype(np.float32)
    emb.backward(_t(torch, g).view(B, S, D))
    emb.update_params()
    torch.cuda.synchronize()
    o = _oracle_opt(oracle, opt, 1)
    oracle.update_params(ro, vi, g, o, table)
    assert_close(emb.table().cpu().numpy(), table, 1e-4, 1e-5, "atomic sgd")


@pytest.mark.parametrize("world", [2, 4])
@pytest.mark.parametrize("emb_type", ["localized", "distributed"])
def test_multi_rank_partition_single_process(oracle, world, emb_type):
    """All ranks' shards driven from one process on one GPU (the all-to-all / reduce-scatter is
    emulated with tensor copies): filter + hash + pool + exchange + reorder must reproduce the
    world=1 result, and the per-rank CSR must equal the oracle's filter bit-exactly."""
    import torch
    import hugectr_amd as ha
    from hugectr_amd import _lib
    rng = np.random.default_rng(world)
    B, S, D, hot, vps = 32, 7, 16, 3, 30
    V = S * vps
    ro, keys = make_csr(rng, B, S, hot, vps)
    localized = emb_type == "localized"
    et = _lib.EMB_LOCALIZED if localized else _lib.EMB_DISTRIBUTED
    # one shared "logical" table: key k lives in row k of a dense [V, D] matrix
    dense = rng.standard_normal((V, D)).astype(np.float32)
    full = oracle.forward(ro, keys.astype(np.uint64), dense, D, 0).reshape(B, S, D)
    shards = []
    for r in range(world):
        e = ha.SparseEmbeddingHash(et, B, 0, V, D, S * hot, S, 0, ha.OptParams(), rank=r, world=world)
        kk = np.arange(V, dtype=np.int64)
        own = kk[(kk // vps) % world == r] if localized else kk[kk % world == r]
        e.load_parameters(torch.from_numpy(own), torch.from_numpy(own // vps), torch.from_numpy(dense[own]))
        shards.append(e)
    outs = [e.forward(True, _t(torch, ro), _t(torch, keys)) for e in shards]
    torch.cuda.synchronize()
    bpg = B // world
    if localized:
        for r, e in enumerate(shards):
            fro, fkeys = oracle.localized_filter(ro, keys, B, S, r, world)
            assert e.slots_on_rank == oracle.slots_on_gpu(S, r, world)
            vi = e.value_index(fkeys.size).cpu().numpy()
            # rows were loaded in ascending key order: row = rank of the key within `own`
            kk = np.arange(V, dtype=np.int64)
            own = kk[(kk // vps) % world == r]
            assert (own[vi] == fkeys).all(), "filtered key stream differs from the oracle filter"
        for dst in range(world):  # emulate the all-to-all: dst receives its sample slice from all
            recv = torch.cat([o[dst * bpg:(dst + 1) * bpg].reshape(-1) for o in outs])
            got = ha.forward_reorder(recv, bpg, S, D, world).cpu().numpy()
            want = oracle.forward_reorder(recv.cpu().numpy(), bpg, S, D, world)
            assert (got == want).all()
            assert (got == full[dst * bpg:(dst + 1) * bpg]).all()
            back = ha.backward_reorder(torch.from_numpy(got).cuda(), bpg, S, D, world)
            assert (back.cpu() == recv.cpu()).all()
    else:
        # reduce-scatter(sum) of the partial sums
        total = torch.stack(outs).sum(0).cpu().numpy()
        assert_close(total, full, 1e-5, 1e-5, "distributed partial sums")
        for r, e in enumerate(shards):
            fro, fkeys = oracle.distributed_filter(ro, keys, B, S, r, world)
            assert e.slots_on_rank == S


def _seq_sum(parts, dt):
    """rank-order sum of the per-rank partial tensors in the embedding type (what the emulated
    reduce-scatter of this test does: every add rounds to the type)"""
    import torch
    acc = parts[0].clone()
    for q in parts[1:]:
        acc = (acc + q).to(acc.dtype)
    return acc


@pytest.mark.parametrize("opt_kw", [dict(optimizer=6, atomic_update=False), dict(optimizer=6, atomic_update=True),
                                    dict(optimizer=3)])
@pytest.mark.parametrize("D", [16, 11])
@pytest.mark.parametrize("dt", ["f32", "f16"])
@pytest.mark.parametrize("world", [2, 4])
def test_distributed_mean_divides_by_the_global_count(oracle, world, dt, D, opt_kw):
    """DistributedSlotSparseEmbeddingHash, combiner mean, N > 1 GPUs
    (R/HugeCTR/include/embeddings/distributed_slot_sparse_embedding_hash.hpp:152-221): every GPU
    pools partial SUMS of the keys it owns (key % N), the reduce-scatter adds them, and only then
    forward_scale divides by the bucket's key count over all GPUs; backward divides the gathered top
    gradient by the same global count.  Ragged multi-hot with empty and single-key buckets, all
    shards driven from one process (collectives emulated by tensor arithmetic).  The result must be
    the world = 1 result: fp32 within 1e-6, fp16 bit-exact against the align2 rule."""
    import torch
    import hugectr_amd as ha
    from hugectr_amd import _lib
    rng = np.random.default_rng(100 * world + D)
    B, S, hot, vps = 32, 5, 6, 40
    V = S * vps
    ro, keys = make_csr(rng, B, S, hot, vps)
    tdt = torch.float32 if dt == "f32" else torch.float16
    dense = (rng.standard_normal((V, D)) * 0.5).astype(np.float32)
    opt = ha.OptParams(lr=0.05, scaler=1.0 if dt == "f32" else 128.0, **opt_kw)
    shards, owns = [], []
    for r in range(world):
        e = ha.SparseEmbeddingHash(_lib.EMB_DISTRIBUTED, B, B, V, D, S * hot, S, 1, opt,
                                   out_dtype=tdt, rank=r, world=world)
        kk = np.arange(V, dtype=np.int64)
        own = kk[kk % world == r]
        e.load_parameters(torch.from_numpy(own), None, torch.from_numpy(dense[own]))
        shards.append(e)
        owns.append(own)
    n = (ro[1:] - ro[:-1]).astype(np.float32)
    sc = np.where(n > 1, np.float32(1) / np.maximum(n, 1), np.float32(1)).astype(np.float32)
    bpg = B // world
    for is_train in (True, False):
        parts = [e.forward(is_train, _t(torch, ro), _t(torch, keys)) for e in shards]
        torch.cuda.synchronize()
        for r, e in enumerate(shards):  # partial SUMS over the rank's keys, in key order
            fro, fkeys = oracle.distributed_filter(ro, keys, B, S, r, world)
            want = oracle.round_to(oracle.forward(fro, fkeys.astype(np.uint64), dense, D, 0), dt)
            got = parts[r].float().cpu().numpy().reshape(-1, D)
            assert (got.view(np.uint32) == want.view(np.uint32)).all(), "partial sums"
        summed = _seq_sum(parts, dt)  # reduce-scatter(sum): rank r keeps sample slice r
        for r, e in enumerate(shards):
            loc = summed[r * bpg:(r + 1) * bpg].contiguous()
            before = loc.float().cpu().numpy().reshape(-1, D)
            e.forward_scale(is_train, loc)
            got = loc.float().cpu().numpy().reshape(-1, D)
            scr = sc[r * bpg * S:(r + 1) * bpg * S, None]
            if dt == "f32":
                want = before * scr
            elif D % 2 == 0:
                want = oracle.round_to(before * oracle.round_to(scr, dt), dt)
            else:
                want = oracle.round_to(before * scr, dt)
            assert (got.view(np.uint32) == want.view(np.uint32)).all(), "forward_scale"
            one = oracle.forward(ro, keys.astype(np.uint64), dense, D, 1)[r * bpg * S:(r + 1) * bpg * S]
            assert_close(got, one, 1e-6 if dt == "f32" else 2e-3, 1e-6 if dt == "f32" else 2e-3,
                         "distributed mean vs world = 1")
    # backward: the all-gathered top gradient [B, S, D] reaches every rank
    g = oracle.round_to(rng.standard_normal((B * S, D)).astype(np.float32), dt)
    wg = oracle.backward_mixed(ro, g, D, 1, dt)  # divides by the GLOBAL count
    table = dense.copy()
    o = _oracle_opt(oracle, opt, 1)
    o.state_half = 1 if dt == "f16" else 0  # q6: fp16 embeddings keep fp16-valued state
    st = np.zeros_like(table)
    if opt.optimizer == _lib.OPT_ADAGRAD:
        oracle.update_params(ro, keys.astype(np.uint64), wg, o, table, st)
    else:
        oracle.update_params(ro, keys.astype(np.uint64), wg, o, table)
    for r, e in enumerate(shards):
        top = _t(torch, g).to(tdt).view(B, S, D)
        e.backward(top)
        got_wg = e.get_wgrad().float().cpu().numpy().reshape(-1, D)
        assert (got_wg.view(np.uint32) == wg.view(np.uint32)).all(), "wgrad uses the global count"
        e.update_params()
        torch.cuda.synchronize()
        got = e.table().cpu().numpy()[:owns[r].size]
        tol = 1e-5 if not opt.atomic_update else 1e-4
        assert_close(got, table[owns[r]], tol, tol, f"rank {r} table after the update")


def test_dump_load_roundtrip(oracle):
    import torch
    import hugectr_amd as ha
    from hugectr_amd import _lib
    rng = np.random.default_rng(5)
    B, S, D, vps = 32, 4, 16, 25
    ro, keys = make_csr(rng, B, S, 3, vps)
    a = ha.SparseEmbeddingHash(_lib.EMB_LOCALIZED, B, B, S * vps, D, S * 3, S, 0, ha.OptParams(),
                               slot_size_array=[vps] * S)
    a.init_params()
    out_a = a.forward(True, _t(torch, ro), _t(torch, keys))
    k, sid, vec = a.dump_parameters()
    assert k.numel() == len(np.unique(keys))
    # slot id of a key is key // vps (keys carry cumulative slot offsets)
    assert (sid.cpu().numpy() == k.cpu().numpy() // vps).all()
    b = ha.SparseEmbeddingHash(_lib.EMB_LOCALIZED, B, B, S * vps, D, S * 3, S, 0, ha.OptParams())
    b.load_parameters(k, sid, vec)
    out_b = b.forward(False, _t(torch, ro), _t(torch, keys))
    assert (out_a.cpu() == out_b.cpu()).all()
    assert b.get_vocabulary_size() == a.get_vocabulary_size()


@pytest.mark.parametrize("D,opt_kw", [(128, dict(optimizer=6, atomic_update=False)),
                                      (16, dict(optimizer=3)),
                                      (64, dict(optimizer=1, update_type=0))])
def test_update_power_law_duplicates(oracle, D, opt_kw):
    """Criteo-like skew: tiny tables (3, 4, 10 rows) next to big ones -> a row collects thousands
    of gradients; exercises the long-run (tail/head partial + combine) path of the segmented
    update as well as runs crossing exactly one tile border."""
    import torch
    import hugectr_amd as ha
    from hugectr_amd import _lib
    rng = np.random.default_rng(11)
    B = 2048
    sizes = [3, 4, 10, 36, 1000, 50000, 1, 97]
    S = len(sizes)
    offs = np.concatenate([[0], np.cumsum(sizes)[:-1]])
    keys = np.stack([np.minimum((rng.pareto(1.1, size=B)).astype(np.int64), v - 1) + o
                     for v, o in zip(sizes, offs)], axis=1).reshape(-1)
    ro = np.arange(B * S + 1, dtype=np.int64)
    V = int(sum(sizes))
    opt = ha.OptParams(lr=0.01, scaler=1.0, **opt_kw)
    emb = ha.SparseEmbeddingHash(_lib.EMB_LOCALIZED, B, 0, V, D, S, S, 0, opt)
    emb.init_params()
    torch.cuda.synchronize()
    table = emb.table().cpu().numpy().copy()
    ns = {1: 2, 3: 1, 6: 0}[opt.optimizer]
    s0 = np.zeros_like(table) if ns >= 1 else None
    s1 = np.zeros_like(table) if ns >= 2 else None
    ht = oracle.HashTable(V, 8)
    for it in range(2):
        emb.forward(True, _t(torch, ro), _t(torch, keys))
        vi = ht.get_insert(keys)
        g = (rng.standard_normal((B * S, D)) * 0.1).astype(np.float32)
        emb.backward(_t(torch, g).view(B, S, D))
        emb.update_params()
        torch.cuda.synchronize()
        oracle.update_params(ro, vi, g, _oracle_opt(oracle, opt, it + 1), table, s0, s1, None)
        # long runs are summed tile-wise (different association than the sequential oracle)
        assert_close(emb.table().cpu().numpy(), table, 2e-4, 2e-5, f"table it{it}")
        keys = np.roll(keys, 7)  # different run/tile alignment in the second step


@pytest.mark.parametrize("combiner", [0, 1])
@pytest.mark.parametrize("key32", [False, True])
def test_update_walks_giant_and_empty_buckets(oracle, combiner, key32):
    """the key-parallel CSR walk of the (row, bucket) expansion (for_each_key_wave): a bucket of
    5000 keys (many trips of one wavefront chunk, shared by several wavefronts), runs of empty
    buckets across chunk borders, a bucket count that is no multiple of 64 -- forward and two SGD
    steps against the oracle"""
    import torch
    import hugectr_amd as ha
    from hugectr_amd import _lib
    rng = np.random.default_rng(41)
    B, S, D = 37, 3, 16
    lens = rng.integers(0, 4, size=B * S)
    lens[5] = 5000
    lens[60:70] = 0
    lens[-1] = 700
    ro = np.concatenate([[0], np.cumsum(lens)]).astype(np.int64)
    V = 900
    slot_of = np.repeat(np.tile(np.arange(S), B), lens)
    keys = (rng.integers(0, V // S, size=slot_of.size) + slot_of * (V // S)).astype(np.int64)
    opt = ha.OptParams(lr=0.05, scaler=1.0, optimizer=_lib.OPT_SGD, atomic_update=False)
    kdt = torch.int32 if key32 else torch.int64  # (u32 keys AND u32 row offsets, as the reader hands them)
    emb = ha.SparseEmbeddingHash(_lib.EMB_LOCALIZED, B, 0, V, D, int(lens.reshape(B, S).sum(1).max()), S,
                                 combiner, opt, key_dtype=kdt)
    emb.init_params()
    torch.cuda.synchronize()
    table = emb.table().cpu().numpy().copy()
    ht = oracle.HashTable(V, 4 if key32 else 8)
    for it in range(2):
        out = emb.forward(True, _t(torch, ro).to(kdt), _t(torch, keys).to(kdt))
        vi = ht.get_insert(keys)
        want = oracle.forward(ro, vi, table, D, combiner)
        assert_close(out.cpu().numpy().reshape(-1, D), want, 1e-5, 1e-6, f"forward it{it}")
        g = (rng.standard_normal((B * S, D)) * 0.1).astype(np.float32)
        emb.backward(_t(torch, g).view(B, S, D))
        emb.update_params()
        torch.cuda.synchronize()
        wg = oracle.backward(ro, g, D, combiner)
        oracle.update_params(ro, vi, wg, _oracle_opt(oracle, opt, it + 1), table, None, None, None)
        assert_close(emb.table().cpu().numpy(), table, 2e-4, 2e-5, f"table it{it}")


@pytest.mark.parametrize("D,dt", [(128, "fp16"), (16, "fp32"), (64, "bf16")])
@pytest.mark.parametrize("combiner,hot", [(0, 1), (1, 5)])
@pytest.mark.parametrize("opt_name", ["sgd", "adagrad"])
def test_sgd_apply_folded_into_the_reduce_is_bit_equal(monkeypatch, D, dt, combiner, hot, opt_name):
    """SGD and AdaGrad apply each unique row where its gradient sum completes (seg_reduce_kernel
    <.., kFuseSgd / kFuseAdaGrad>); HCTR_SGD_FUSED=0 parks the sums and applies them in a second
    pass (seg_apply).  Same arithmetic -> same bits, on skewed keys with long runs and ragged buckets."""
    import torch
    import hugectr_amd as ha
    from hugectr_amd import _lib
    rng = np.random.default_rng(23)
    B = 4096
    sizes = [3, 10, 1000, 50000, 1, 97, 200000]
    S = len(sizes)
    offs = np.concatenate([[0], np.cumsum(sizes)[:-1]])
    lens = rng.integers(0 if hot > 1 else 1, hot + 1, size=B * S)
    ro = np.concatenate([[0], np.cumsum(lens)]).astype(np.int64)
    slot_of = np.repeat(np.tile(np.arange(S), B), lens)
    keys = (np.minimum(rng.pareto(1.1, size=slot_of.size).astype(np.int64),
                       np.array(sizes)[slot_of] - 1) + offs[slot_of]).astype(np.int64)
    tdt = {"fp32": torch.float32, "fp16": torch.float16, "bf16": torch.bfloat16}[dt]
    tables = []
    for fused in ("1", "0"):
        monkeypatch.setenv("HCTR_SGD_FUSED", fused)
        opt = ha.OptParams(lr=0.05, scaler=128.0, atomic_update=False, initial_accu_value=0.0,
                           optimizer=_lib.OPT_SGD if opt_name == "sgd" else _lib.OPT_ADAGRAD)
        emb = ha.SparseEmbeddingHash(_lib.EMB_LOCALIZED, B, 0, int(sum(sizes)), D, S * hot, S, combiner,
                                     opt, out_dtype=tdt, seed=7)
        emb.init_params()
        g_rng = np.random.default_rng(5)
        for _ in range(3):
            emb.forward(True, _t(torch, ro), _t(torch, keys))
            g = torch.from_numpy(g_rng.standard_normal((B, S, D)).astype(np.float32)).cuda().to(tdt)
            emb.backward(g)
            emb.update_params()
        torch.cuda.synchronize()
        tables.append(emb.table().clone())
    assert torch.equal(tables[0], tables[1])


@pytest.mark.parametrize("presort", ["1", "0"])
def test_rank_shard_updates_with_presort_guess(oracle, presort, monkeypatch):
    """world = 2, rank 1 of a localized embedding over several train steps whose per-rank nnz
    jumps (so the side-stream sort's size guess -- previous exact nnz + 1/8 -- is once too short
    and update_params must re-sort in line): table after every step == oracle on the filtered CSR.
    HCTR_PRESORT=0 runs the same steps with the sort on the caller's stream."""
    import torch
    import hugectr_amd as ha
    from hugectr_amd import _lib
    monkeypatch.setenv("HCTR_PRESORT", presort)
    rng = np.random.default_rng(11)
    B, S, D, vps, world, rank = 256, 6, 16, 300, 2, 1
    V = S * vps
    opt = ha.OptParams(optimizer=_lib.OPT_ADAGRAD, lr=0.05, epsilon=1e-6, scaler=1.0)
    emb = ha.SparseEmbeddingHash(_lib.EMB_LOCALIZED, B, 0, V, D, S * 12, S, 1, opt, rank=rank,
                                 world=world)
    emb.init_params()
    torch.cuda.synchronize()
    table = emb.table().cpu().numpy().copy()
    acc = np.zeros_like(table)
    ht = oracle.HashTable(V, 8)
    s_r = oracle.slots_on_gpu(S, rank, world)
    for it, hot in enumerate([2, 2, 12, 3, 12]):  # nnz: ~1.2k, 1.2k, 7.4k (guess short), 1.8k, 7.4k
        ro, keys = make_csr(rng, B, S, hot, vps, empty_frac=0.1)
        fro, fkeys = oracle.localized_filter(ro, keys, B, S, rank, world)
        out = emb.forward(True, _t(torch, ro), _t(torch, keys))
        vi = ht.get_insert(fkeys)
        want = oracle.forward(fro, vi, table, D, 1)
        assert_close(out.cpu().numpy().reshape(-1, D), want, 1e-5, 1e-6, f"fwd it{it}")
        g = rng.standard_normal((B * s_r, D)).astype(np.float32)
        emb.backward(_t(torch, g).view(B, s_r, D).contiguous())
        emb.update_params()
        torch.cuda.synchronize()
        wg = oracle.backward(fro, g, D, 1)
        oracle.update_params(fro, vi, wg, _oracle_opt(oracle, opt, it + 1), table, acc)
        assert_close(emb.table().cpu().numpy(), table, 1e-5, 1e-6, f"table it{it}")
        assert_close(emb.opt_state(0).cpu().numpy(), acc, 1e-5, 1e-6, f"accum it{it}")


@pytest.mark.parametrize("D", [4, 16, 64, 128, 256])
@pytest.mark.parametrize("combiner", [0, 1])
def test_stateless_pool_kernels_agree_bit_exact(oracle, D, combiner):
    """hctr_forward_pool (bucket-major) and hctr_forward_pool_multihot (flat key walk) on ragged
    buckets with empty ones, very long ones and missing rows: identical bits, equal to the oracle."""
    import torch
    from hugectr_amd import _lib
    rng = np.random.default_rng(D + combiner)
    nb, V = 1003, 500
    lens = rng.integers(0, 7, size=nb)
    lens[rng.random(nb) < 0.3] = 0
    lens[[5, 400, 1002]] = [300, 77, 19]        # long buckets, one of them the very last
    lens[:3] = 0                                  # leading empties
    ro = np.zeros(nb + 1, dtype=np.int64)
    np.cumsum(lens, out=ro[1:])
    vi = rng.integers(0, V, size=int(ro[-1])).astype(np.uint64)
    vi[rng.random(vi.size) < 0.05] = oracle.INVALID   # eval misses contribute 0 but count (q3)
    table = rng.standard_normal((V, D)).astype(np.float32)
    want = oracle.forward(ro, vi, table, D, combiner)
    rot, vit, tt = _t(torch, ro), _t(torch, vi.view(np.int64)), _t(torch, table)
    outs = []
    for fn in (_lib.lib.hctr_forward_pool, _lib.lib.hctr_forward_pool_multihot):
        out = torch.full((nb, D), 7.0, device="cuda")
        _lib.check(fn(nb, D, combiner, _lib.ptr(rot), _lib.KEY_I64, _lib.ptr(vit), _lib.ptr(tt),
                      _lib.ptr(out), _lib.F32, _lib.stream_ptr()))
        outs.append(out.cpu().numpy())
        assert (outs[-1].view(np.uint32) == want.view(np.uint32)).all()
    assert (outs[0].view(np.uint32) == outs[1].view(np.uint32)).all()


@pytest.mark.parametrize("dtype", ["f16", "bf16"])
@pytest.mark.parametrize("D,combiner,one_hot", [(16, 1, False), (128, 0, True), (128, 1, False),
                                                (11, 1, False), (6, 1, False)])
def test_mixed_precision_forward_wgrad_update(oracle, dtype, D, combiner, one_hot):
    """16-bit pooled vectors and top gradients (the reference's use_mixed_precision mode, fp16;
    bf16 is the same rule with bf16): forward and wgrad BIT-exact against the restated align2 /
    generic kernels (half-precision multiply by half(1/n) for even sizes), table after
    the update at rel 1e-5 (fp32 accumulation of the 16-bit wgrads in ascending bucket order)."""
    import torch
    import hugectr_amd as ha
    from hugectr_amd import _lib
    tdt = torch.float16 if dtype == "f16" else torch.bfloat16
    rng = np.random.default_rng(D * 3 + combiner + (dtype == "f16"))
    B, S, hot, vps = 64, 5, 6, 30
    V = S * vps
    opt = ha.OptParams(optimizer=_lib.OPT_SGD, lr=0.1, scaler=2.0, atomic_update=False)
    emb = ha.SparseEmbeddingHash(_lib.EMB_LOCALIZED, B, 0, V, D, S * hot, S, combiner, opt,
                                 out_dtype=tdt)
    emb.init_params()
    torch.cuda.synchronize()
    table = emb.table().cpu().numpy().copy()
    ht = oracle.HashTable(V, 8)
    for it in range(2):
        ro, keys = make_csr(rng, B, S, hot, vps, one_hot=one_hot)
        out = emb.forward(True, _t(torch, ro), _t(torch, keys))
        assert out.dtype == tdt
        vi = ht.get_insert(keys)
        want = oracle.forward_mixed(ro, vi, table, D, combiner, dtype)
        got = out.float().cpu().numpy().reshape(-1, D)
        assert (got.view(np.uint32) == want.view(np.uint32)).all(), f"forward {dtype} it{it}"
        g = (rng.standard_normal((B * S, D)) * 3).astype(np.float32)
        gt = _t(torch, g).to(tdt).view(B, S, D).contiguous()
        emb.backward(gt)
        wg = emb.get_wgrad().float().cpu().numpy().reshape(-1, D)
        want_wg = oracle.backward_mixed(ro, g, D, combiner, dtype)
        assert (wg.view(np.uint32) == want_wg.view(np.uint32)).all(), f"wgrad {dtype} it{it}"
        emb.update_params()
        torch.cuda.synchronize()
        oracle.update_params(ro, vi, want_wg, _oracle_opt(oracle, opt, it + 1), table)
        assert_close(emb.table().cpu().numpy(), table, 1e-5, 1e-6, f"table {dtype} it{it}")


@pytest.mark.parametrize("name,kw", [o for o in OPTS if o[0] != "sgd"],
                         ids=[o[0] for o in OPTS if o[0] != "sgd"])
def test_fp16_embedding_keeps_fp16_valued_optimizer_state(oracle, name, kw):
    """with fp16 embeddings the reference's optimizer state is OptimizerTensor<__half>
    (optimizer.hpp:284-296): read as float, stored back rounded to fp16, the weight step of the
    same launch uses the unrounded value.  The state arrays here hold exactly those fp16 values."""
    import torch
    import hugectr_amd as ha
    from hugectr_amd import _lib
    rng = np.random.default_rng(11)
    # every row occurs at most once per batch: the per-row gradient is a single fp16 value, so GPU
    # and oracle feed the SAME float into the fp16 rounding of the state (with summed gradients a
    # last-bit difference of the fp32 sum can flip an fp16 ulp of v, which Adam amplifies)
    B, S, hot, vps, D, combiner = 32, 6, 1, 64, 16, 0
    V = S * vps + 16
    opt = ha.OptParams(lr=0.05, scaler=4.0, **kw)
    emb = ha.SparseEmbeddingHash(_lib.EMB_LOCALIZED, B, 0, V, D, S * hot, S, combiner, opt,
                                 out_dtype=torch.float16)
    emb.init_params()
    torch.cuda.synchronize()
    table = emb.table().cpu().numpy().copy()
    ns = {1: 2, 3: 1, 5: 1, 4: 1}[opt.optimizer]
    s0 = np.zeros_like(table)
    s1 = np.zeros_like(table) if ns >= 2 else None
    pt = np.ones(table.shape, dtype=np.uint64) if name == "adam_lazy" else None
    ht = oracle.HashTable(V, 8)
    for it in range(4):
        ro = np.arange(B * S + 1, dtype=np.int64)
        keys = np.stack([s_ * vps + rng.permutation(vps)[:B] for s_ in range(S)], 1).reshape(-1)
        keys = keys.astype(np.int64)
        emb.forward(True, _t(torch, ro), _t(torch, keys))
        vi = ht.get_insert(keys)
        g = (rng.standard_normal((B * S, D)) * 8).astype(np.float32)
        emb.backward(_t(torch, g).to(torch.float16).view(B, S, D).contiguous())
        want_wg = oracle.backward_mixed(ro, g, D, combiner, "f16")
        emb.update_params()
        torch.cuda.synchronize()
        oo = _oracle_opt(oracle, opt, it + 1)
        oo.state_half = 1
        oracle.update_params(ro, vi, want_wg, oo, table, s0, s1, pt)
        assert_close(emb.table().cpu().numpy(), table, 1e-5, 1e-6, f"{name} table it{it}")
        for k, want in ((0, s0), (1, s1)):
            if want is None:
                continue
            assert emb.opt_state(k).dtype == torch.float16, "the state of fp16 embeddings is stored in fp16"
            got = emb.opt_state(k).float().cpu().numpy()
            if name == "adam_lazy":  # powf is not correctly rounded: allow an fp16 ulp
                assert_close(got, want, 1e-3, 1e-7, f"{name} state{k} it{it}")
            else:  # float multiply / add, then ONE conversion: the same bits as the oracle
                assert (got.view(np.uint32) == want.view(np.uint32)).all(), f"{name} state{k} it{it}"
    assert np.abs(s0).max() > 0


@pytest.mark.parametrize("D", [16, 128])
def test_row_with_a_quarter_million_gradients(D):
    """one row collects 250 000 gradients (7 800 tile partials): seg_combine_big_kernel adds them as
    four chunks on four workgroups, the last one to finish adds the chunk sums in chunk order --
    right to float64 within summation error, and the same bits every time"""
    import ctypes
    import torch
    from hugectr_amd import _lib
    lib, ptr = _lib.lib, _lib.ptr
    rng = np.random.default_rng(D)
    V, nb = 64, 300_000
    rows = rng.integers(1, V, size=nb)
    rows[rng.random(nb) < 0.83] = 0          # ~250 k positions of row 0
    rows[1000:1000 + 70_000] = 5             # and one run of ~2 200 partials (more than one chunk)
    ro = torch.arange(nb + 1, dtype=torch.int64, device="cuda")
    idx = torch.from_numpy(rows.astype(np.int64)).cuda()
    g = (rng.standard_normal((nb, D)) * 0.01).astype(np.float32)
    gt = torch.from_numpy(g).cuda()
    t0 = rng.standard_normal((V, D)).astype(np.float32)
    want = t0.astype(np.float64)
    np.subtract.at(want, rows, 0.5 * g.astype(np.float64) / 2.0)
    u = ctypes.c_void_p()
    _lib.check(lib.hctr_updater_create(nb, V, D, ctypes.byref(u)))
    outs = []
    for _ in range(3):
        tab = torch.from_numpy(t0).cuda()
        _lib.check(lib.hctr_updater_update(u, nb, nb, ptr(ro), ptr(idx), ptr(gt), _lib.F32, _lib.OPT_SGD,
                                           _lib.UPDATE_LOCAL, 0.5, 0.9, 0.999, 1e-7, 0.0, 2.0, 1, ptr(tab),
                                           None, None, _lib.stream_ptr()))
        torch.cuda.synchronize()
        outs.append(tab.clone())
    lib.hctr_updater_destroy(u)
    assert torch.equal(outs[0], outs[1]) and torch.equal(outs[0], outs[2])
    err = np.abs(outs[0].cpu().numpy().astype(np.float64) - want).max()
    assert err < 2e-3, err  # row 0 moved by ~ sqrt(250k) * 0.01 * 0.25: fp32 partial sums


@pytest.mark.parametrize("opt_kw", [dict(optimizer=6, atomic_update=False), dict(optimizer=3),
                                    dict(optimizer=1, update_type=0)],
                         ids=["sgd", "adagrad", "adam"])
@pytest.mark.parametrize("B,D,dt,hot_rows", [(16384, 128, "f16", 8192), (12000, 16, "f32", 500),
                                             (4100, 64, "bf16", 16384)])
def test_update_hot_rows_of_one_hot_batches(oracle, monkeypatch, opt_kw, B, D, dt, hot_rows):
    """the hot-row path of the sparse update (hot_chunk_kernel + hot_apply_kernel, the cold pairs
    sorted on the side stream by a first pass that leaves the hot rows out) at sizes where it is on
    by default in the bench: Criteo-like skew (tables of 3 / 4 / 10 rows whose rows fill whole
    streams and cross chunk borders, power-law tables, a nearly unique one), rows on both sides of
    the bound, a ragged batch in between (both kernels exit on the device flag).  Table and state
    against the oracle within the re-association of long sums; the same bits on a second handle;
    next to the plain path (HCTR_HOT_ROWS=0)."""
    import torch
    import hugectr_amd as ha
    from hugectr_amd import _lib
    sizes = [3, 4, 10, 36, 1000, 50000, 200000, 97]
    S = len(sizes)
    offs = np.concatenate([[0], np.cumsum(sizes)[:-1]])
    V = int(sum(sizes))
    tdt = {"f32": torch.float32, "f16": torch.float16, "bf16": torch.bfloat16}[dt]
    opt = ha.OptParams(lr=0.01, scaler=2.0, **opt_kw)
    ns = {1: 2, 3: 1, 6: 0}[opt.optimizer]

    def run(rows_env):
        monkeypatch.setenv("HCTR_HOT_MIN", "0")
        monkeypatch.setenv("HCTR_HOT_ROWS", str(rows_env))
        rng = np.random.default_rng(B + D)
        emb = ha.SparseEmbeddingHash(_lib.EMB_LOCALIZED, B, 0, V, D, 2 * S, S, 0, opt, out_dtype=tdt)
        emb.init_params()
        torch.cuda.synchronize()
        table = emb.table().cpu().numpy().copy()
        s0 = np.zeros_like(table) if ns >= 1 else None
        s1 = np.zeros_like(table) if ns >= 2 else None
        ht = oracle.HashTable(V, 8)
        tol_cum = np.zeros(V)
        for it in range(4):
            if it == 2:
                lens = rng.integers(0, 3, size=B * S)
                ro = np.concatenate([[0], np.cumsum(lens)]).astype(np.int64)
                slot_of = np.repeat(np.tile(np.arange(S), B), lens)
                keys = (np.minimum(rng.pareto(1.1, size=slot_of.size).astype(np.int64),
                                   np.array(sizes)[slot_of] - 1) + offs[slot_of]).astype(np.int64)
            else:
                ro = np.arange(B * S + 1, dtype=np.int64)
                keys = np.stack([np.minimum((rng.pareto(1.1, size=B)).astype(np.int64), v - 1) + o
                                 for v, o in zip(sizes, offs)], axis=1).reshape(-1)
            emb.forward(True, _t(torch, ro), _t(torch, keys))
            vi = ht.get_insert(keys)
            g = (rng.standard_normal((B * S, D)) * 0.1).astype(np.float32)
            gt = _t(torch, g).to(tdt).view(B, S, D).contiguous()
            emb.backward(gt)
            torch.cuda.synchronize()
            emb_before = emb.table().cpu().numpy().copy()
            emb.update_params()
            torch.cuda.synchronize()
            wg = oracle.backward(ro, gt.float().cpu().numpy().reshape(-1, D), D, 0)
            oo = _oracle_opt(oracle, opt, it + 1)
            oo.state_half = 1 if dt == "f16" else 0
            before = table.copy()
            oracle.update_params(ro, vi, wg, oo, table, s0, s1, None)
            got = emb.table().cpu().numpy()
            assert_close(got, table, 1e-3, 1e-4, f"table it{it} H={rows_env}")
            if opt.optimizer == 6:
                # plain SGD is linear in the gradient sum, so the bound the re-association implies
                # can be written down: a length-n fp32 sum taken in another order differs by about
                # eps * n * rms(g) (rounding errors of partial sums that grow like sqrt(k)); the
                # weight moves by lr / scaler times that.  8 x for the tails, + 2 ulp of the weight.
                ok = vi != np.uint64(0xFFFFFFFFFFFFFFFF)
                rows = vi[ok].astype(np.int64)
                gpos = wg[np.repeat(np.arange(ro.size - 1), np.diff(ro))][ok]
                n_r = np.bincount(rows, minlength=V).astype(np.float64)
                ms = np.bincount(rows, weights=(gpos.astype(np.float64) ** 2).mean(1), minlength=V)
                rms = np.sqrt(ms / np.maximum(n_r, 1))
                tol_cum += (opt.lr / opt.scaler) * 8 * 2.0 ** -24 * n_r * rms  # (steps add up)
                err = np.abs(got.astype(np.float64) - table)
                bound = (tol_cum[:, None] + 2 * (it + 1) * np.spacing(np.abs(table).astype(np.float32))
                         + 1e-12)
                assert (err <= bound).all(), (it, rows_env, float((err / bound).max()))
                assert (got[n_r == 0] == emb_before[n_r == 0]).all()
            if s0 is not None:
                assert_close(emb.opt_state(0).cpu().numpy(), s0, 2e-3, 1e-4, f"state0 it{it}")
        return emb.table().clone()

    a = run(hot_rows)
    b = run(hot_rows)
    assert torch.equal(a, b), "the hot path is not deterministic"
    c = run(0)
    assert_close(a.cpu().numpy(), c.cpu().numpy(), 1e-3, 1e-4, "hot path vs plain path")


@pytest.mark.parametrize("opt_kw", [dict(optimizer=6, atomic_update=False), dict(optimizer=3),
                                    dict(optimizer=1, update_type=0)],
                         ids=["sgd", "adagrad", "adam"])
@pytest.mark.parametrize("B,D,dt", [(8192, 128, "f16"), (6000, 16, "f32"), (4100, 8, "bf16")])
def test_update_cold_rows_counted_per_row(oracle, monkeypatch, opt_kw, B, D, dt):
    """the cold rows' chain of the sparse update (cold_count / base / scatter / reduce: rows counted
    per row instead of sorted) with nearly every row cold (HCTR_HOT_ROWS=2): rows met once, short
    runs (2 .. 32 positions, summed in ascending position order = the reference's stable-sort
    order), long runs sorted inside LDS (a 50-row table: ~ B / 50 positions a row) and a run longer
    than the LDS list (the third row of a 3-row table: ~ B / 3 positions, re-derived in order by a
    scan of the batch).  Batch 1 is ragged WITH as many keys as buckets (the host cannot tell: the
    device flag sends every row, hot ones included, through the chain and the gradient row of a
    position comes from a search of the offsets; mean combiner scaling is checked there too).
    Table / state against the oracle: rows with at most 32 positions to 1e-6 (same order of
    additions), the others within the re-association of pieces of 32; the same bits from a second
    handle; close to the sorting path (HCTR_COLD_COUNT=0)."""
    import torch
    import hugectr_amd as ha
    from hugectr_amd import _lib
    sizes = [3, 50, 3000, 100000]
    S = len(sizes)
    offs = np.concatenate([[0], np.cumsum(sizes)[:-1]])
    V = int(sum(sizes))
    tdt = {"f32": torch.float32, "f16": torch.float16, "bf16": torch.bfloat16}[dt]
    opt = ha.OptParams(lr=0.01, scaler=2.0, **opt_kw)
    ns = {1: 2, 3: 1, 6: 0}[opt.optimizer]

    def run(cold_env, combiner):
        monkeypatch.setenv("HCTR_HOT_MIN", "0")
        monkeypatch.setenv("HCTR_HOT_ROWS", "2")
        monkeypatch.setenv("HCTR_COLD_COUNT", cold_env)
        rng = np.random.default_rng(B + D)
        emb = ha.SparseEmbeddingHash(_lib.EMB_LOCALIZED, B, 0, V, D, 2 * S, S, combiner, opt,
                                     out_dtype=tdt)
        emb.init_params()
        torch.cuda.synchronize()
        table = emb.table().cpu().numpy().copy()
        s0 = np.zeros_like(table) if ns >= 1 else None
        s1 = np.zeros_like(table) if ns >= 2 else None
        ht = oracle.HashTable(V, 8)
        for it in range(3):
            if it == 1:  # ragged, but nnz == buckets: pairs of buckets hold (0, 2) / (2, 0) / (1, 1) keys
                kind = rng.integers(0, 3, size=B * S // 2)
                lens = np.stack([np.array([0, 2, 1])[kind], np.array([2, 0, 1])[kind]], 1).reshape(-1)
                assert lens.sum() == B * S
                ro = np.concatenate([[0], np.cumsum(lens)]).astype(np.int64)
                slot_of = np.repeat(np.tile(np.arange(S), B), lens)
                keys = (rng.integers(0, 1 << 30, size=slot_of.size) % np.array(sizes)[slot_of] +
                        offs[slot_of]).astype(np.int64)
            else:
                ro = np.arange(B * S + 1, dtype=np.int64)
                keys = np.stack([rng.integers(0, v, size=B) + o for v, o in zip(sizes, offs)],
                                axis=1).reshape(-1).astype(np.int64)
            emb.forward(True, _t(torch, ro), _t(torch, keys))
            vi = ht.get_insert(keys)
            g = (rng.standard_normal((B * S, D)) * 0.1).astype(np.float32)
            gt = _t(torch, g).to(tdt).view(B, S, D).contiguous()
            emb.backward(gt)
            torch.cuda.synchronize()
            before = emb.table().cpu().numpy().copy()
            emb.update_params()
            torch.cuda.synchronize()
            wg = oracle.backward_mixed(ro, gt.float().cpu().numpy().reshape(-1, D), D, combiner, dt)
            oo = _oracle_opt(oracle, opt, it + 1)
            oo.state_half = 1 if dt == "f16" else 0
            oracle.update_params(ro, vi, wg, oo, table, s0, s1, None)
            got = emb.table().cpu().numpy()
            if opt.optimizer == 3:
                # AdaGrad steps by lr * g / sqrt(accum): an element whose first gradient sum is
                # nearly zero moves by ~ lr whatever the sum's size, so sums that differ in their
                # last bits (pieces of 32 against one chain) can sit up to 2 lr apart there
                err = np.abs(got.astype(np.float64) - table)
                bad = err > 1e-4 + 1e-3 * np.abs(table)
                assert bad.sum() <= 1e-5 * bad.size and err.max() <= 2.5 * opt.lr, (bad.sum(), err.max())
            else:
                assert_close(got, table, 1e-3, 1e-4, f"table it{it} cold={cold_env}")
            if s0 is not None:
                assert_close(emb.opt_state(0).cpu().numpy(), s0, 2e-3, 1e-4, f"state0 it{it}")
            cnt = np.bincount(vi.astype(np.int64), minlength=V)
            few = (cnt > 0) & (cnt <= 32)
            assert few.sum() > 1000 and ((cnt > 2048).any() or B < 8192) and ((cnt > 32) & (cnt <= 2048)).any()
            if cold_env == "1":
                assert_close(got[few], table[few], 1e-6, 1e-7, f"short runs it{it}")
            assert (got[cnt == 0] == before[cnt == 0]).all(), "a row without a key moved"
        return emb.table().clone()

    for combiner in (0, 1):
        a = run("1", combiner)
        b = run("1", combiner)
        assert torch.equal(a, b), "the cold rows' chain is not deterministic"
        c = run("0", combiner)
        if opt.optimizer != 3:
            assert_close(a.cpu().numpy(), c.cpu().numpy(), 1e-3, 1e-4, "counted vs sorted cold rows")


@pytest.mark.parametrize("ahead", [False, True], ids=["cooperative_finish", "index_ahead_two_launches"])
def test_index_stage_beside_a_device_full_of_gemms(oracle, ahead):
    """the index stage with unseen keys on a side stream while the default stream keeps the device
    full of GEMMs -- the inter-iteration-overlap situation.  In-line form: the cooperative finish
    kernel's grid barrier must open (no timeout flag, error bit 4) however its workgroups are
    scheduled between GEMM workgroups; index_ahead form: the two-launch finish kernel.  Rows
    bit-exact against the sequential oracle in both."""
    import torch
    import hugectr_amd as ha
    from hugectr_amd import _lib
    rng = np.random.default_rng(31)
    B, S, D, vps = 32768, 8, 16, 60000
    V = S * vps
    opt = ha.OptParams(optimizer=_lib.OPT_SGD, lr=0.1, atomic_update=False)
    emb = ha.SparseEmbeddingHash(_lib.EMB_LOCALIZED, B, 0, V, D, S, S, 0, opt)
    emb.init_params()
    ht = oracle.HashTable(V, 8)
    ro = torch.arange(B * S + 1, dtype=torch.int64, device="cuda")
    a = torch.randn(8192, 8192, device="cuda", dtype=torch.float16)
    side = torch.cuda.Stream()

    def keys_of():
        k = np.stack([rng.integers(0, vps, size=B) + s * vps for s in range(S)], 1).reshape(-1)
        return k.astype(np.int64)
    k0 = keys_of()
    emb.index(True, ro, _t(torch, k0))
    assert (emb.value_index(k0.size).cpu().numpy().view(np.uint64) == ht.get_insert(k0)).all()
    for it in range(3):
        k = keys_of()  # ~ 40 % unseen keys: the finish kernel has real work
        kt = _t(torch, k)
        torch.cuda.synchronize()
        for _ in range(12):  # ~ 10 ms of GEMMs queued on the default stream
            a @ a
        with torch.cuda.stream(side):
            if ahead:
                emb.index_ahead(ro, kt)
            else:
                emb.index(True, ro, kt)
        for _ in range(12):
            a @ a
        torch.cuda.current_stream().wait_stream(side)
        torch.cuda.synchronize()
        if ahead:
            emb.index_adopt()
        emb.check_overflow()  # (a barrier that never opened raises here: error bit 4)
        assert (emb.value_index(k.size).cpu().numpy().view(np.uint64) == ht.get_insert(k)).all()


@pytest.mark.parametrize("name,kw", [OPTS[0], OPTS[1], OPTS[4]], ids=["sgd", "adam_local", "adagrad"])
@pytest.mark.parametrize("D,dt,combiner", [(1, "f32", 0), (1, "f16", 0), (3, "f32", 1), (6, "f16", 0),
                                           (20, "f32", 0)])
def test_long_runs_of_short_vectors_match_oracle(oracle, name, kw, D, dt, combiner):
    """Vectors whose length is no multiple of 4 take the generic update (one wavefront per distinct
    row); a row met more than 64 times in the batch -- the wide tables of Wide & Deep, D = 1, whose
    hot rows collect thousands of gradients -- is summed by 64 / L lane groups side by side and a
    fixed butterfly instead of one lane walking the run.  Tables of 2 / 5 / 300 rows: runs of ~ 1000,
    ~ 400 and of a handful (the plain ascending sum), multi-hot and mean; the oracle adds in
    ascending bucket order, so the bound is the re-association's: n eps per element."""
    import torch
    import hugectr_amd as ha
    from hugectr_amd import _lib
    rng = np.random.default_rng(D)
    tdt = torch.float16 if dt == "f16" else torch.float32
    B, sizes, hot = 2048, [2, 5, 300], 2
    S = len(sizes)
    offs = np.concatenate([[0], np.cumsum(sizes)[:-1]])
    V = int(sum(sizes))
    opt = ha.OptParams(lr=0.05, scaler=4.0, **kw)
    emb = ha.SparseEmbeddingHash(_lib.EMB_LOCALIZED, B, 0, V, D, S * hot, S, combiner, opt,
                                 out_dtype=tdt)
    emb.init_params()
    torch.cuda.synchronize()
    table = emb.table().cpu().numpy().copy()
    ns = {1: 2, 3: 1, 6: 0}[opt.optimizer]
    s0 = np.zeros_like(table) if ns >= 1 else None
    s1 = np.zeros_like(table) if ns >= 2 else None
    ht = oracle.HashTable(V, 8)
    for it in range(3):
        lens = rng.integers(1, hot + 1, size=B * S)
        ro = np.concatenate([[0], np.cumsum(lens)]).astype(np.int64)
        slot_of = np.repeat(np.tile(np.arange(S), B), lens)
        keys = (rng.integers(0, 1 << 30, size=slot_of.size) % np.array(sizes)[slot_of] + offs[slot_of]).astype(np.int64)
        emb.forward(True, _t(torch, ro), _t(torch, keys))
        vi = ht.get_insert(keys)
        g = (rng.standard_normal((B * S, D)) * 0.1).astype(np.float32)
        gt = _t(torch, g).to(tdt).view(B, S, D).contiguous()
        emb.backward(gt)
        emb.update_params()
        torch.cuda.synchronize()
        wg = oracle.backward(ro, gt.float().cpu().numpy().reshape(-1, D), D, combiner)
        oo = _oracle_opt(oracle, opt, it + 1)
        oo.state_half = 1 if dt == "f16" else 0
        oracle.update_params(ro, vi, wg, oo, table, s0, s1, None)
        # ~ 1000 addends of size 0.1: 1000 eps32 x 0.1 x sqrt-ish growth, / scaler, through lr (SGD) or the
        # optimizer's normalisation (Adam / AdaGrad move by ~ lr whatever the gradient's size)
        assert_close(emb.table().cpu().numpy(), table, 2e-4, 2e-5, f"{name} table it{it}")
        if s0 is not None:
            assert_close(emb.opt_state(0).float().cpu().numpy(), s0, 2e-3, 1e-5, f"{name} state0 it{it}")
