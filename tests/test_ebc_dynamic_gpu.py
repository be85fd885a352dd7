"""GPU parity: embedding_collection on DYNAMIC tables (BASELINE config 5,
R/HugeCTR/embedding_storage/dynamic_embedding.cu) -- raw-key routing, hctr_det_lookup_rows
(ILookup::lookup -> float**), hctr_forward_pool_ptrs, hctr_ebc_local_reduce (Wgrad of unique keys)
and the table's fused optimizer step -- against the same CPU restatements as the static path
(EmbeddingReferenceCPU, R/test/utest/embedding_collection/reference_embedding.hpp:32-237) and the
dict-based dynamic-table oracle for the optimizers static tables do not have."""
import ctypes

import numpy as np
import pytest

from util import assert_close

pytestmark = pytest.mark.gpu


def _make_inputs(rng, B, vocabs, lookup_table, max_hot):
    L = len(lookup_table)
    lens = rng.integers(0, max_hot + 1, size=L * B).astype(np.int64)
    lens[rng.random(L * B) < 0.15] = 0
    br = np.zeros(L * B + 1, np.int64)
    np.cumsum(lens, out=br[1:])
    keys = np.concatenate([rng.integers(0, vocabs[lookup_table[l]], size=int(lens[l * B:(l + 1) * B].sum()))
                           for l in range(L)]).astype(np.int64)
    return keys, br


@pytest.mark.parametrize("D", [16, 128, 10, 7])
@pytest.mark.parametrize("dtype", ["f32", "f16", "bf16"])
def test_pool_ptrs_equals_pool_by_index(D, dtype):
    """reading rows through addresses gives the same bits as reading them through row indices"""
    import torch
    from hugectr_amd import _lib
    from hugectr_amd._lib import check, lib, ptr, stream_ptr
    rng = np.random.default_rng(D)
    rows, buckets = 500, 777
    table = torch.from_numpy(rng.standard_normal((rows, D)).astype(np.float32)).cuda()
    lens = rng.integers(0, 6, size=buckets)
    ro = torch.from_numpy(np.concatenate([[0], np.cumsum(lens)]).astype(np.int64)).cuda()
    nnz = int(lens.sum())
    idx = rng.integers(0, rows, size=nnz).astype(np.int64)
    missing = rng.random(nnz) < 0.1
    idx_t = torch.from_numpy(np.where(missing, -1, idx)).cuda()           # SIZE_MAX = missing
    ptrs = torch.from_numpy(np.where(missing, 0, table.data_ptr() + idx * D * 4)).cuda()
    tdt = {"f32": torch.float32, "f16": torch.float16, "bf16": torch.bfloat16}[dtype]
    cdt = {"f32": _lib.F32, "f16": _lib.F16, "bf16": _lib.BF16}[dtype]
    for comb in (0, 1):
        a = torch.empty((buckets, D), dtype=tdt, device="cuda")
        b = torch.empty_like(a)
        check(lib.hctr_forward_pool(buckets, D, comb, ptr(ro), _lib.KEY_I64, ptr(idx_t), ptr(table),
                                    ptr(a), cdt, stream_ptr()))
        check(lib.hctr_forward_pool_ptrs(buckets, D, comb, ptr(ro), ptr(ptrs), ptr(b), cdt,
                                         stream_ptr()))
        torch.cuda.synchronize()
        assert torch.equal(a.view(torch.int16 if tdt != torch.float32 else torch.int32),
                           b.view(torch.int16 if tdt != torch.float32 else torch.int32)), (D, dtype, comb)


@pytest.mark.parametrize("D,hot", [(16, 3), (128, 1), (10, 4), (32, 40)])
def test_local_reduce(D, hot):
    """unique rows ascending, first-occurrence keys, per-row gradient sums"""
    import torch
    from hugectr_amd import _lib
    from hugectr_amd._lib import check, lib, ptr, stream_ptr
    rng = np.random.default_rng(D + hot)
    buckets = 3000
    lens = rng.integers(0, hot + 1, size=buckets)
    br = np.concatenate([[0], np.cumsum(lens)]).astype(np.int64)
    nnz = int(br[-1])
    # skewed row ids out of a sparse 40-bit space: long runs and singletons
    pool = np.unique(rng.integers(0, 1 << 40, size=400))
    row_ids = pool[np.minimum((rng.pareto(1.1, size=nnz)).astype(np.int64), pool.size - 1)]
    keys = row_ids * 7 + 3                                   # any function of the row
    grad = rng.standard_normal((buckets, D)).astype(np.float32)
    upd = ctypes.c_void_p()
    check(lib.hctr_updater_create(nnz + 5, nnz + 5, D, ctypes.byref(upd)))
    d = lambda a: torch.from_numpy(a).cuda()
    urow = torch.empty(nnz, dtype=torch.int64, device="cuda")
    ukey = torch.empty(nnz, dtype=torch.int64, device="cuda")
    wg = torch.empty((nnz, D), dtype=torch.float32, device="cuda")
    nu = ctypes.c_size_t()
    t_br, t_rows, t_keys, t_grad = d(br), d(row_ids), d(keys), d(grad)
    check(lib.hctr_ebc_local_reduce(upd, buckets, nnz, ptr(t_br), ptr(t_rows), int(pool.max()),
                                    ptr(t_keys), ptr(t_grad), _lib.F32, ctypes.byref(nu), ptr(urow),
                                    ptr(ukey), ptr(wg), stream_ptr()))
    torch.cuda.synchronize()
    want_rows = np.unique(row_ids)
    n = nu.value
    assert n == want_rows.size
    assert (urow[:n].cpu().numpy() == want_rows).all()
    assert (ukey[:n].cpu().numpy() == want_rows * 7 + 3).all()
    bucket_of = np.repeat(np.arange(buckets), lens)
    want = np.zeros((n, D), np.float64)
    pos = np.searchsorted(want_rows, row_ids)
    np.add.at(want, pos, grad[bucket_of].astype(np.float64))
    mag = np.zeros((n, D), np.float64)       # fp32 summation error scales with sum |g| of the run
    np.add.at(mag, pos, np.abs(grad[bucket_of]).astype(np.float64))
    err = np.abs(wg[:n].cpu().numpy().astype(np.float64) - want)
    assert (err <= 2e-6 * mag + 1e-6).all(), f"local_reduce sums: max err {err.max():.3e}"
    lib.hctr_updater_destroy(upd)


def _rows_at(torch, addr, rows, ev):
    """fp32 rows `rows` of the flat [.][ev] array at device address addr (the library's own gather
    with one key per bucket)"""
    from hugectr_amd import _lib
    from hugectr_amd._lib import check, lib, ptr, stream_ptr
    n = rows.numel()
    out = torch.empty((n, ev), dtype=torch.float32, device="cuda")
    rng = torch.arange(n + 1, dtype=torch.int64, device="cuda")
    check(lib.hctr_forward_pool(n, ev, 0, ptr(rng), _lib.KEY_I64, ptr(rows), addr, ptr(out),
                                _lib.F32, stream_ptr()))
    torch.cuda.synchronize()
    return out


@pytest.mark.parametrize("world,shard", [(1, "table"), (2, "table"), (4, "row"), (2, "mixed")])
@pytest.mark.parametrize("opt_name", ["sgd", "adagrad", "ftrl"])
@pytest.mark.parametrize("preload", [True, False])
def test_ebc_dynamic_forward_backward_update(oracle, world, shard, opt_name, preload):
    import torch
    import hugectr_amd as ha
    from hugectr_amd import _lib
    rng = np.random.default_rng(world * 11 + (1 if preload else 0))
    B, ev = 32, 16
    vocabs = [50, 7, 300, 12]
    lookup_table = [0, 1, 2, 3, 2]
    combiners = ["sum", "mean", "sum", "mean", "mean"]
    T, L = len(vocabs), len(lookup_table)
    tcfg = [ha.EmbeddingTableConfig(f"t{i}", v, ev) for i, v in enumerate(vocabs)]
    cfg = ha.EmbeddingCollectionConfig()
    for l in range(L):
        cfg.embedding_lookup(tcfg[lookup_table[l]], f"in{l}", f"out{l}", combiners[l])
    if shard == "table":
        sm = [[1 if t % world == g else 0 for t in range(T)] for g in range(world)]
    elif shard == "row":
        sm = [[1] * T for _ in range(world)]
    else:
        sm = [[1 if g == 0 else 0, 1, 1, 1 if g == world - 1 else 0] for g in range(world)]
    cfg.shard(sm)
    opt = {"sgd": _lib.OPT_SGD, "adagrad": _lib.OPT_ADAGRAD, "ftrl": _lib.OPT_FTRL}[opt_name]
    ftrl = (0.02, 0.05, 0.3)
    # tiny initial capacity: the maps have to grow while training
    ranks = [ha.EmbeddingCollection.for_rank(r, world, cfg, B, lr=0.1, optimizer=opt, scaler=2.0,
                                             epsilon=1e-6, max_hotness=4, ftrl=ftrl,
                                             storage="dynamic", initializer="" if preload else "0.5",
                                             init_capacity=8)
             for r in range(world)]
    row_start = np.concatenate([[0], np.cumsum(vocabs)[:-1]]).astype(np.int64)
    if preload:  # known random vectors for every key, pushed through the table's own verbs
        dense = rng.standard_normal((sum(vocabs), ev)).astype(np.float32)
        for t in range(T):
            owners = ranks[0].owners[t]
            for sid, g in enumerate(owners):
                e = ranks[g]
                ks = np.arange(sid, vocabs[t], len(owners)).astype(np.int64)
                c = e.class_of_table[t]
                tk = torch.from_numpy(ks).cuda()
                e.det.lookup(tk, [c], [0, ks.size])
                e.det.scatter_update(tk, torch.from_numpy(dense[row_start[t] + ks]).cuda().view(-1),
                                     [c], [0, ks.size])
    else:
        dense = np.full((sum(vocabs), ev), 0.5, np.float32)
    accum = np.zeros_like(dense)
    ftrl_z = np.zeros_like(dense)
    seen = [set() for _ in range(T)]
    bpg = B // world
    comb = [0 if c == "sum" else 1 for c in combiners]
    for it in range(3):
        keys, br = _make_inputs(rng, B, vocabs, lookup_table, 4)
        for l in range(L):
            seen[lookup_table[l]].update(keys[br[l * B]:br[(l + 1) * B]].tolist())
        gk, gbr = torch.from_numpy(keys).cuda(), torch.from_numpy(br).cuda()
        sends = [e.route_and_pool(gk, gbr) for e in ranks]
        torch.cuda.synchronize()
        outs = []
        for d_, e in enumerate(ranks):
            blocks = []
            for s, es in enumerate(ranks):
                if es.n_local:
                    blocks.append(sends[s].view(world, es.n_local, bpg, ev)[d_].reshape(-1, ev))
            recv = torch.cat(blocks) if blocks else torch.empty((0, ev), device="cuda")
            outs.append(e.network_forward(recv.contiguous()))
        want = oracle.ebc_forward(B, lookup_table, ev, comb, keys, br, row_start, dense, num_gpus=world)
        for d_ in range(world):
            assert_close(outs[d_].cpu().numpy().reshape(-1), want[d_], 1e-5, 1e-6, f"fwd rank{d_} it{it}")
        grads = [rng.standard_normal(outs[d_].shape).astype(np.float32) for d_ in range(world)]
        bsends = [ranks[d_].network_backward(torch.from_numpy(grads[d_]).cuda()) for d_ in range(world)]
        torch.cuda.synchronize()
        for s, es in enumerate(ranks):
            if es.n_local == 0:
                continue
            base = sum(ranks[0].n_local_of[:s])
            tops = [bsends[d_].view(-1, bpg, ev)[base:base + es.n_local] for d_ in range(world)]
            es.apply_gradients(torch.stack(tops).contiguous())
        torch.cuda.synchronize()
        oracle.ebc_backward_update(B, lookup_table, ev, comb, keys, br, row_start, dense,
                                   np.stack([g.reshape(-1) for g in grads]),
                                   optimizer={"sgd": 0, "adagrad": 1, "ftrl": 2}[opt_name], lr=0.1,
                                   scaler=2.0, epsilon=1e-6, accum=accum, num_gpus=world, ftrl=ftrl,
                                   ftrl_z=ftrl_z)
        for t in range(T):
            owners = ranks[0].owners[t]
            for sid, g in enumerate(owners):
                e = ranks[g]
                k, v = e.det.export(e.class_of_table[t])
                k, v = k.cpu().numpy(), v.cpu().numpy()
                assert np.unique(k).size == k.size
                if preload:
                    assert k.size == np.arange(sid, vocabs[t], len(owners)).size
                else:  # exactly the keys this shard has been asked for so far
                    assert set(k.tolist()) == {x for x in seen[t] if x % len(owners) == sid}
                assert (k % len(owners) == sid).all()
                assert_close(v, dense[row_start[t] + k], 1e-5, 1e-6, f"table {t} shard {sid} it{it}")


@pytest.mark.parametrize("opt_name", ["adam", "momentum", "nesterov", "rmsprop"])
def test_ebc_dynamic_other_optimizers(opt_name):
    """the optimizers only dynamic tables have, against the dict oracle (one rank)"""
    import torch
    import hugectr_amd as ha
    from hugectr_amd import _lib
    from oracle import det_oracle as do
    rng = np.random.default_rng(5)
    B, ev = 16, 8
    vocabs = [40, 9]
    lookup_table = [0, 1, 0]
    combiners = ["sum", "mean", "mean"]
    tcfg = [ha.EmbeddingTableConfig(f"t{i}", v, ev) for i, v in enumerate(vocabs)]
    cfg = ha.EmbeddingCollectionConfig()
    for l in range(3):
        cfg.embedding_lookup(tcfg[lookup_table[l]], f"in{l}", f"out{l}", combiners[l])
    cfg.shard([[1, 1]])
    code = {"adam": _lib.OPT_ADAM, "momentum": _lib.OPT_MOMENTUM_SGD, "nesterov": _lib.OPT_NESTEROV,
            "rmsprop": _lib.OPT_RMSPROP}[opt_name]
    ocode = {"adam": do.ADAM, "momentum": do.MOMENTUM, "nesterov": do.NESTEROV,
             "rmsprop": do.RMSPROP}[opt_name]
    e = ha.EmbeddingCollection.for_rank(0, 1, cfg, B, lr=0.05, optimizer=code, scaler=1.0,
                                        epsilon=1e-6, max_hotness=3, storage="dynamic",
                                        initializer="0.25", init_capacity=4)
    w = do.DetOracle([ev, ev], 0.25)
    st = do.DetOracle([ev * (2 if opt_name == "adam" else 1)] * 2, 0.0)
    for it in range(3):
        keys, br = _make_inputs(rng, B, vocabs, lookup_table, 3)
        out = e.forward(torch.from_numpy(keys).cuda(), torch.from_numpy(br).cuda())
        # forward + per-key gradients on the host
        g = rng.standard_normal((3, B, ev)).astype(np.float32)
        want = np.zeros((3, B, ev), np.float32)
        kg = {}
        order = []
        for l in range(3):
            c = lookup_table[l]
            for b in range(B):
                ks = keys[br[l * B + b]:br[l * B + b + 1]]
                n = ks.size
                acc = np.zeros(ev, np.float32)
                for k in ks:
                    acc = acc + w.lookup([k], [c], [0, 1])
                if combiners[l] == "mean" and n > 1:
                    acc = acc * np.float32(1.0 / n)
                want[l, b] = acc
                gi = g[l, b] / np.float32(n) if (combiners[l] == "mean" and n > 0) else g[l, b]
                for k in ks:
                    if (c, int(k)) not in kg:
                        kg[(c, int(k))] = np.zeros(ev, np.float32)
                        order.append((c, int(k)))
                    kg[(c, int(k))] = kg[(c, int(k))] + gi
        assert_close(out.cpu().numpy(), want, 1e-5, 1e-6, f"fwd it{it}")
        e.backward_and_update(torch.from_numpy(g).cuda())
        torch.cuda.synchronize()
        order.sort()
        uk = np.array([k for _, k in order], np.int64)
        cls = [c for c, _ in order]
        n0 = cls.count(0)
        wg = np.concatenate([kg[x] for x in order]) if order else np.zeros(0, np.float32)
        do.update(w, st, ocode, uk, [0, 1], [0, n0, len(order)],
                  np.arange(len(order)) * ev, wg, lr=0.05, eps=1e-6, times=it + 1)
        for c in range(2):
            k, v = e.det.export(c)
            k, v = k.cpu().numpy(), v.cpu().numpy()
            assert set(k.tolist()) == set(w.maps[c].keys())
            ref = np.stack([w.maps[c][int(x)] for x in k]) if k.size else np.zeros((0, ev), np.float32)
            assert_close(v, ref, 2e-5, 2e-6, f"{opt_name} class {c} it{it}")


@pytest.mark.parametrize("batch_major", [False, True])
@pytest.mark.parametrize("opt_name,dtype", [("sgd", "float32"), ("adam", "float16"), ("adagrad", "bfloat16")])
def test_one_gpu_direct_path_on_dynamic_tables_equals_staged(monkeypatch, batch_major, opt_name, dtype):
    """One GPU, dynamic tables: pooling through the row pointers straight into the output
    (transposed store for batch-major) and the local reduce reading the output's gradient in place
    give the bits of the staged route -> pool -> network_forward / network_backward path."""
    import torch
    import hugectr_amd as ha
    from hugectr_amd import _lib
    rng = np.random.default_rng(31)
    B, ev = 64, 32
    vocabs = [70, 9, 400]
    lookup_table = [0, 1, 2, 2, 0]
    tcfg = [ha.EmbeddingTableConfig(f"t{i}", -1, ev) for i in range(len(vocabs))]
    cfg = ha.EmbeddingCollectionConfig()
    for l, t in enumerate(lookup_table):
        cfg.embedding_lookup(tcfg[t], f"in{l}", f"out{l}", "sum")
    opt = {"sgd": _lib.OPT_SGD, "adam": _lib.OPT_ADAM, "adagrad": _lib.OPT_ADAGRAD}[opt_name]
    kw = dict(lr=0.05, optimizer=opt, scaler=4.0, epsilon=1e-6, batch_major=batch_major,
              max_hotness=4, out_dtype=getattr(torch, dtype), seed=3, storage="dynamic",
              initializer="", init_capacity=16)
    monkeypatch.setenv("HCTR_EBC_DIRECT", "0")
    staged = ha.EmbeddingCollection.for_rank(0, 1, cfg, B, **kw)
    monkeypatch.setenv("HCTR_EBC_DIRECT", "1")
    direct = ha.EmbeddingCollection.for_rank(0, 1, cfg, B, **kw)
    assert direct._direct and direct.dynamic and not staged._direct
    L = len(lookup_table)
    for step in range(4):
        lens = rng.integers(0, 5, size=L * B).astype(np.int64)
        br = np.zeros(L * B + 1, np.int64)
        np.cumsum(lens, out=br[1:])
        keys = np.concatenate([rng.integers(0, vocabs[lookup_table[l]], size=int(lens[l * B:(l + 1) * B].sum()))
                               for l in range(L)]).astype(np.int64)
        kt, brt = torch.from_numpy(keys).cuda(), torch.from_numpy(br).cuda()
        a, b = staged.forward(kt, brt), direct.forward(kt, brt)
        assert a.shape == b.shape and torch.equal(a, b), step
        g = torch.randn(a.shape, device="cuda").to(a.dtype)
        staged.backward_and_update(g)
        direct.backward_and_update(g)
    assert staged.det.size() == direct.det.size() > 0


@pytest.mark.gpu
@pytest.mark.parametrize("direct", ["0", "1"])
@pytest.mark.parametrize("dtype", ["float32", "float16"])
def test_flat_row_store_gives_the_bits_of_the_pointer_and_unique_key_path(monkeypatch, direct, dtype):
    """One ev_size per group: the classes' rows are one flat table (hctr_det_row_store), so the
    static tables' gather runs on the row numbers and -- SGD -- their sort + segmented reduce
    applies the step in place.  Same pooled vectors and the same table, bit for bit, as the
    pointer-per-key gather and the unique keys -> wgrad -> optimizer kernel -> scatter_add flow of
    the reference (dynamic_embedding.cu:130-330) that HCTR_DYNAMIC_FLAT=0 keeps; the classes grow
    (and the store moves) several times on the way."""
    import torch
    import hugectr_amd as ha
    from hugectr_amd import _lib
    rng = np.random.default_rng(77)
    B, ev = 96, 32
    vocabs = [3000, 9, 400, 50000]
    lookup_table = [0, 1, 2, 2, 3, 0]
    tcfg = [ha.EmbeddingTableConfig(f"t{i}", -1, ev) for i in range(len(vocabs))]
    cfg = ha.EmbeddingCollectionConfig()
    for l, t in enumerate(lookup_table):
        cfg.embedding_lookup(tcfg[t], f"in{l}", f"out{l}", "sum")
    kw = dict(lr=0.05, optimizer=_lib.OPT_SGD, scaler=8.0, batch_major=True, max_hotness=6,
              out_dtype=getattr(torch, dtype), seed=5, storage="dynamic", initializer="",
              init_capacity=16)
    monkeypatch.setenv("HCTR_EBC_DIRECT", direct)
    monkeypatch.setenv("HCTR_DYNAMIC_FLAT", "0")
    ptrs = ha.EmbeddingCollection.for_rank(0, 1, cfg, B, **kw)
    monkeypatch.setenv("HCTR_DYNAMIC_FLAT", "1")
    flat = ha.EmbeddingCollection.for_rank(0, 1, cfg, B, **kw)
    assert flat._dyn_flat and not ptrs._dyn_flat and flat.det.row_store()[0]
    L = len(lookup_table)
    caps = [flat.det.capacity_per_class()]
    for step in range(6):
        lens = rng.integers(0, 7, size=L * B).astype(np.int64)
        br = np.zeros(L * B + 1, np.int64)
        np.cumsum(lens, out=br[1:])
        keys = np.concatenate([rng.integers(0, vocabs[lookup_table[l]], size=int(lens[l * B:(l + 1) * B].sum()))
                               for l in range(L)]).astype(np.int64)
        kt, brt = torch.from_numpy(keys).cuda(), torch.from_numpy(br).cuda()
        a, b = ptrs.forward(kt, brt), flat.forward(kt, brt)
        assert torch.equal(a, b), step
        g = torch.randn(a.shape, device="cuda").to(a.dtype)
        ptrs.backward_and_update(g)
        flat.backward_and_update(g)
        caps.append(flat.det.capacity_per_class())
    assert caps[-1] != caps[0], "the test is meant to cross several growth steps"
    assert ptrs.det.size() == flat.det.size() > 0
    for c in range(len(flat.det.dims)):
        (ka, va), (kb, vb) = ptrs.det.export(c), flat.det.export(c)
        oa, ob = torch.argsort(ka), torch.argsort(kb)
        assert torch.equal(ka[oa], kb[ob]) and torch.equal(va[oa], vb[ob]), c
    # eval: unseen keys are not inserted and pool as zeros on both paths
    ptrs.training = flat.training = False
    keys = torch.arange(10 ** 9, 10 ** 9 + L * B, dtype=torch.int64).cuda()
    br = torch.arange(L * B + 1, dtype=torch.int64).cuda()
    a, b = ptrs.forward(keys, br), flat.forward(keys, br)
    assert torch.equal(a, b) and float(b.abs().max()) == 0.0


@pytest.mark.gpu
@pytest.mark.parametrize("direct", ["0", "1"])
@pytest.mark.parametrize("opt_name,dtype", [("adagrad", "float32"), ("adam", "float32"),
                                            ("momentum", "float32"), ("adagrad", "float16"),
                                            ("adam", "bfloat16")])
def test_flat_row_store_stateful_optimizers_give_the_bits_of_the_unique_key_path(
        monkeypatch, direct, opt_name, dtype):
    """AdaGrad / Adam / MomentumSGD on the flat row store: the optimizer state lies at the weights'
    row numbers (hctr_det_state_store) and the static tables' sparse update applies the step in
    place -- one probe per key (the forward's).  Same pooled vectors and the same table, bit for
    bit, as the reference's flow that HCTR_DYNAMIC_FLAT=0 keeps: unique keys -> wgrad -> probe of
    the weights + inserting probe of a state table keyed like them -> *_update_grad_kernel ->
    scatter_add (dynamic_embedding.cu:227-330); the classes grow several times on the way (memory
    is mapped behind rows AND state, zero state for the new rows), and keys come back after steps in
    which they were absent (their state must still be there)."""
    import torch
    import hugectr_amd as ha
    from hugectr_amd import _lib
    rng = np.random.default_rng(78)
    B, ev = 96, 32
    vocabs = [3000, 9, 400, 50000]
    lookup_table = [0, 1, 2, 2, 3, 0]
    tcfg = [ha.EmbeddingTableConfig(f"t{i}", -1, ev) for i in range(len(vocabs))]
    cfg = ha.EmbeddingCollectionConfig()
    for l, t in enumerate(lookup_table):
        cfg.embedding_lookup(tcfg[t], f"in{l}", f"out{l}", "sum" if l % 2 == 0 else "mean")
    opt = {"adagrad": _lib.OPT_ADAGRAD, "adam": _lib.OPT_ADAM, "momentum": _lib.OPT_MOMENTUM_SGD}[opt_name]
    kw = dict(lr=0.05, optimizer=opt, scaler=8.0, epsilon=1e-6, batch_major=True, max_hotness=6,
              out_dtype=getattr(torch, dtype), seed=5, storage="dynamic", initializer="",
              init_capacity=16, beta1=0.8, beta2=0.95, momentum_factor=0.7)
    monkeypatch.setenv("HCTR_EBC_DIRECT", direct)
    monkeypatch.setenv("HCTR_DYNAMIC_FLAT", "0")
    ptrs = ha.EmbeddingCollection.for_rank(0, 1, cfg, B, **kw)
    monkeypatch.setenv("HCTR_DYNAMIC_FLAT", "1")
    flat = ha.EmbeddingCollection.for_rank(0, 1, cfg, B, **kw)
    assert flat._dyn_flat and not ptrs._dyn_flat
    L = len(lookup_table)
    caps = [flat.det.capacity_per_class()]
    for step in range(7):
        lens = rng.integers(0, 7, size=L * B).astype(np.int64)
        if step == 3:  # a batch without keys: no step is counted (Adam's bias correction)
            lens[:] = 0
        br = np.zeros(L * B + 1, np.int64)
        np.cumsum(lens, out=br[1:])
        keys = np.concatenate([rng.integers(0, vocabs[lookup_table[l]], size=int(lens[l * B:(l + 1) * B].sum()))
                               for l in range(L)]).astype(np.int64)
        kt, brt = torch.from_numpy(keys).cuda(), torch.from_numpy(br).cuda()
        a, b = ptrs.forward(kt, brt), flat.forward(kt, brt)
        assert torch.equal(a, b), step
        g = torch.randn(a.shape, device="cuda").to(a.dtype)
        ptrs.backward_and_update(g)
        flat.backward_and_update(g)
        caps.append(flat.det.capacity_per_class())
    assert caps[-1] != caps[0], "the test is meant to cross several growth steps"
    assert ptrs.det.size() == flat.det.size() > 0
    nst = 2 if opt_name == "adam" else 1
    s0, s1 = flat.det.state_store(nst)
    store, total = flat.det.row_store()
    assert s0 and (s1 if nst == 2 else True)
    # (row numbers: one power-of-two range of addresses per class, backed as far as the class grew)
    assert total % len(caps[-1]) == 0 and total // len(caps[-1]) >= max(caps[-1]) and total < 2**32 - 16
    for c in range(len(flat.det.dims)):
        (ka, va), (kb, vb) = ptrs.det.export(c), flat.det.export(c)
        oa, ob = torch.argsort(ka), torch.argsort(kb)
        assert torch.equal(ka[oa], kb[ob]) and torch.equal(va[oa], vb[ob]), c
        # the state itself: the state table's vectors of the unique-key flow, keyed like the
        # weights ([m | v] per key for Adam), against the arrays at the weights' rows
        sk, sv = ptrs.det_opt.states.export(c)
        if sk.numel() == 0:
            continue
        _, rows, _ = flat.det.lookup_rows(sk, [c], [0, sk.numel()], insert=False, want_ptrs=False)
        for j, addr in enumerate((s0, s1)[:nst]):
            assert torch.equal(_rows_at(torch, addr, rows, ev), sv.view(-1, nst, ev)[:, j]), (c, j)
