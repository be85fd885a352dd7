"""GPU parity: embedding_collection path (key routing -> keys_to_indices -> pooled lookup ->
network forward / backward -> static-table optimizer) vs the CPU restatement of the reference's
EmbeddingReferenceCPU (R/test/utest/embedding_collection/reference_embedding.hpp:32-237)."""
import os

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


def test_keys_to_indices_bit_exact(oracle):
    import torch
    from hugectr_amd import _lib
    rng = np.random.default_rng(0)
    keys = rng.integers(0, 10**9, size=10000).astype(np.int64)
    out = torch.empty(keys.size, dtype=torch.int64, device="cuda")
    kd = torch.from_numpy(keys).cuda()
    for start, ns in ((0, 1), (12345, 8), (7, 3)):
        _lib.check(_lib.lib.hctr_ebc_keys_to_indices(_lib.ptr(kd), _lib.KEY_I64, keys.size, start, ns,
                                                     _lib.ptr(out), _lib.stream_ptr()))
        want = np.empty(keys.size, np.int64)
        oracle.lib().hco_keys_to_indices(keys.size, oracle._p(keys), start, ns, oracle._p(want))
        assert (out.cpu().numpy() == want).all()


@pytest.mark.parametrize("world,shard", [(1, "table"), (2, "table"), (4, "row"), (2, "mixed")])
@pytest.mark.parametrize("batch_major", [False, True])
@pytest.mark.parametrize("opt_name", ["sgd", "adagrad", "ftrl"])
def test_ebc_forward_backward_update(oracle, world, shard, batch_major, opt_name):
    import torch
    import hugectr_amd as ha
    from hugectr_amd import _lib
    rng = np.random.default_rng(world * 7 + (1 if batch_major else 0))
    B, ev = 32, 16
    vocabs = [50, 7, 300, 12]
    lookup_table = [0, 1, 2, 3, 2]       # two lookups share table 2
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
    else:  # table 0 and 3 table-wise, 1 and 2 row-wise over all ranks
        sm = [[1 if g == 0 else 0, 1, 1, 1 if g == world - 1 else 0] for g in range(world)]
    cfg.shard(sm)
    opt = {"sgd": _lib.OPT_SGD, "adagrad": _lib.OPT_ADAGRAD, "ftrl": _lib.OPT_FTRL}[opt_name]
    ftrl = (0.02, 0.05, 0.3)  # lambda1, lambda2, beta
    ranks = [ha.EmbeddingCollection.for_rank(r, world, cfg, B, lr=0.1, optimizer=opt, scaler=2.0,
                                             epsilon=1e-6, batch_major=batch_major, max_hotness=4,
                                             ftrl=ftrl)
             for r in range(world)]
    # dense "logical" tables for the oracle, assembled from the shards (row = key)
    row_start = np.concatenate([[0], np.cumsum(vocabs)[:-1]]).astype(np.int64)
    dense = np.zeros((sum(vocabs), ev), np.float32)
    for t in range(T):
        owners = ranks[0].owners[t]
        for sid, g in enumerate(owners):
            e = ranks[g]
            s0 = e.row_start_of_table[t]
            keys_of_shard = np.arange(sid, vocabs[t], len(owners))
            dense[row_start[t] + keys_of_shard] = e.table[s0:s0 + keys_of_shard.size].cpu().numpy()
    accum = np.zeros_like(dense)
    ftrl_z = np.zeros_like(dense)
    bpg = B // world
    for it in range(2):
        keys, br = _make_inputs(rng, B, vocabs, lookup_table, 4)
        gk, gbr = torch.from_numpy(keys).cuda(), torch.from_numpy(br).cuda()
        sends = [e.route_and_pool(gk, gbr) for e in ranks]
        torch.cuda.synchronize()
        outs = []
        for d, e in enumerate(ranks):  # emulate the all-to-all: d receives its sample slice
            blocks = []
            for s, es in enumerate(ranks):
                v = sends[s].view(world, max(es.n_local, 0), bpg, ev) if es.n_local else None
                if v is not None:
                    blocks.append(v[d].reshape(-1, ev))
            recv = torch.cat(blocks) if blocks else torch.empty((0, ev), device="cuda")
            outs.append(e.network_forward(recv.contiguous()))
        want = oracle.ebc_forward(B, lookup_table, ev, [0 if c == "sum" else 1 for c in combiners],
                                  keys, br, row_start, dense, num_gpus=world, batch_major=batch_major)
        for d in range(world):
            assert_close(outs[d].cpu().numpy().reshape(-1), want[d], 1e-5, 1e-6, f"ebc fwd rank{d}")
        # backward + update
        grads = [rng.standard_normal(outs[d].shape).astype(np.float32) for d in range(world)]
        bsends = [ranks[d].network_backward(torch.from_numpy(grads[d]).cuda()) for d in range(world)]
        torch.cuda.synchronize()
        for s, es in enumerate(ranks):  # mirror all-to-all: owner s collects its blocks from all d
            if es.n_local == 0:
                continue
            base = sum(ranks[0].n_local_of[:s])
            tops = [bsends[d].view(-1, bpg, ev)[base:base + es.n_local] for d in range(world)]
            es.apply_gradients(torch.stack(tops).contiguous())
        torch.cuda.synchronize()
        oracle.ebc_backward_update(B, lookup_table, ev, [0 if c == "sum" else 1 for c in combiners],
                                   keys, br, row_start, dense, np.stack([g.reshape(-1) for g in grads]),
                                   optimizer={"sgd": 0, "adagrad": 1, "ftrl": 2}[opt_name], lr=0.1,
                                   scaler=2.0, epsilon=1e-6, accum=accum, num_gpus=world,
                                   batch_major=batch_major, ftrl=ftrl, ftrl_z=ftrl_z)
        for t in range(T):
            owners = ranks[0].owners[t]
            for sid, g in enumerate(owners):
                e = ranks[g]
                s0 = e.row_start_of_table[t]
                ks = np.arange(sid, vocabs[t], len(owners))
                assert_close(e.table[s0:s0 + ks.size].cpu().numpy(), dense[row_start[t] + ks],
                             1e-5, 1e-6, f"table {t} shard {sid} it{it}")


@pytest.mark.parametrize("world,shard", [(2, "table"), (4, "row"), (2, "mixed")])
@pytest.mark.parametrize("storage", ["static", "dynamic"])
def test_key_route_all_to_all_equals_the_gathered_route(world, shard, storage):
    """the reference's DataDistributor route for data-parallel input (bucket lengths a2a, keys a2a,
    sparse_data_distribution_op_impl.cu:215-395) must hand every owner exactly the CSR it finds by
    filtering the gathered global batch: same bucket ranges, same row indices, same pooled vectors"""
    import torch
    import hugectr_amd as ha
    rng = np.random.default_rng(world)
    B, ev = 32, 16
    vocabs = [50, 7, 300, 12]
    lookup_table = [0, 1, 2, 3, 2]
    T, L = len(vocabs), len(lookup_table)
    tcfg = [ha.EmbeddingTableConfig(f"t{i}", v, ev) for i, v in enumerate(vocabs)]
    cfg = ha.EmbeddingCollectionConfig()
    for l in range(L):
        cfg.embedding_lookup(tcfg[lookup_table[l]], f"in{l}", f"out{l}", "sum" if l % 2 else "mean")
    if shard == "table":
        sm = [[1 if t % world == g else 0 for t in range(T)] for g in range(world)]
    elif shard == "row":
        sm = [[1] * T for _ in range(world)]
    else:
        sm = [[1 if g == 0 else 0, 1, 1, 1 if g == world - 1 else 0] for g in range(world)]
    cfg.shard(sm)
    kw = dict(max_hotness=4, storage=storage, initializer="0.5", init_capacity=16)
    a = [ha.EmbeddingCollection.for_rank(r, world, cfg, B, **kw) for r in range(world)]  # a2a route
    g = [ha.EmbeddingCollection.for_rank(r, world, cfg, B, **kw) for r in range(world)]  # gathered
    if storage == "static":
        for x, y in zip(a, g):
            x.table.copy_(y.table)
    bpg = B // world
    for it in range(2):
        keys, br = _make_inputs(rng, B, vocabs, lookup_table, 4)
        gk, gbr = torch.from_numpy(keys).cuda(), torch.from_numpy(br).cuda()
        want = [e.route_and_pool(gk, gbr) for e in g]
        # every rank's own share, feature-major: bucket = lookup * bpg + b_local
        sends = []
        for r in range(world):
            lens, ks = [], []
            for l in range(L):
                for b in range(r * bpg, (r + 1) * bpg):
                    q0, q1 = br[l * B + b], br[l * B + b + 1]
                    lens.append(q1 - q0)
                    ks.append(keys[q0:q1])
            lbr = np.concatenate([[0], np.cumsum(lens)]).astype(np.int64)
            lk = np.concatenate(ks).astype(np.int64) if ks else np.zeros(0, np.int64)
            sends.append(a[r].route_send(torch.from_numpy(lk).cuda(), torch.from_numpy(lbr).cuda()))
        for p in range(world):  # the two all-to-alls
            lens_all = torch.cat([sends[s][0][p] for s in range(world)])
            keys_all = torch.cat([sends[s][1][p] for s in range(world)])
            a[p].route_recv(lens_all, keys_all)
            got = a[p].pool_routed()
            nb, n = a[p].nb, g[p]._nnz_host if storage == "dynamic" else int(g[p].out_range[g[p].nb])
            assert torch.equal(a[p].out_range[:nb + 1], g[p].out_range[:nb + 1]), f"ranges rank{p}"
            assert a[p]._nnz_host == n
            assert torch.equal(a[p].indices[:n], g[p].indices[:n]), f"indices rank{p}"
            assert torch.equal(a[p].counts, g[p].counts), f"bucket counts rank{p}"
            assert torch.equal(got, want[p]), f"pooled vectors rank{p}"


def _route_worker(rank, world, port, ret):
    import torch
    import torch.distributed as dist
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        torch.cuda.set_device(0)
        import hugectr_amd as ha
        rng = np.random.default_rng(5)                      # same stream on both ranks
        B, ev = 64, 16
        vocabs = [50, 7, 300, 12]
        lookup_table = [0, 1, 2, 3, 2]
        L = len(lookup_table)
        tcfg = [ha.EmbeddingTableConfig(f"t{i}", v, ev) for i, v in enumerate(vocabs)]
        cfg = ha.EmbeddingCollectionConfig()
        for l in range(L):
            cfg.embedding_lookup(tcfg[lookup_table[l]], f"in{l}", f"out{l}", "sum" if l % 2 else "mean")
        cfg.shard([[1, 1, 1, 0], [0, 1, 1, 1]])             # tables 1, 2 row-sharded over both
        ea = ha.EmbeddingCollection(cfg, B, lr=0.1, max_hotness=4, key_route="a2a", seed=3)
        eg = ha.EmbeddingCollection(cfg, B, lr=0.1, max_hotness=4, key_route="allgather", seed=3)
        assert torch.equal(ea.table, eg.table)
        bpg = B // world
        for it in range(3):
            keys, br = _make_inputs(rng, B, vocabs, lookup_table, 4)
            lens, ks = [], []
            for l in range(L):
                for b in range(rank * bpg, (rank + 1) * bpg):
                    q0, q1 = br[l * B + b], br[l * B + b + 1]
                    lens.append(q1 - q0)
                    ks.append(keys[q0:q1])
            lbr = torch.from_numpy(np.concatenate([[0], np.cumsum(lens)]).astype(np.int64)).cuda()
            lk = torch.from_numpy(np.concatenate(ks).astype(np.int64)).cuda()
            oa, og = ea.forward(lk, lbr), eg.forward(lk, lbr)
            assert torch.equal(oa, og), f"forward it{it}"
            grad = torch.from_numpy(rng.standard_normal((world,) + tuple(oa.shape)).astype(np.float32))[rank].cuda()
            ea.backward_and_update(grad)
            eg.backward_and_update(grad)
            assert torch.equal(ea.table, eg.table), f"tables it{it}"
        ret[rank] = "ok"
    except Exception as ex:
        import traceback
        ret[rank] = "".join(traceback.format_exception(type(ex), ex, ex.__traceback__))
    finally:
        dist.destroy_process_group()


def test_key_route_all_to_all_two_ranks_on_one_gpu():
    """EmbeddingCollection.forward / backward_and_update with the key-route all-to-all on 2
    processes (gloo, both on this GPU): identical outputs and tables to the all-gather route"""
    import torch.multiprocessing as mp
    ctx = mp.get_context("spawn")
    ret = ctx.Manager().dict()
    port = 29500 + os.getpid() % 2000
    procs = [ctx.Process(target=_route_worker, args=(r, 2, port, ret)) for r in range(2)]
    for p in procs:
        p.start()
    for p in procs:
        p.join(300)
    for r in range(2):
        if ret.get(r) != "ok":
            print(f"--- rank {r} ---\n{ret.get(r)}")
    assert ret.get(0) == "ok" and ret.get(1) == "ok"


@pytest.mark.parametrize("world", [1, 2, 4])
@pytest.mark.parametrize("opt_name", ["sgd", "adagrad", "ftrl"])
def test_data_parallel_tables(oracle, world, opt_name):
    """replicated ("dp") tables: every rank resolves its own samples, the per-row gradient sums are
    all-reduced (emulated here by adding the ranks' operands), every replica takes the same step;
    forward and tables against the same EBC oracle as the model-parallel tables"""
    import torch
    import hugectr_amd as ha
    from hugectr_amd import _lib
    from hugectr_amd.embedding_collection import DataParallelCollection
    rng = np.random.default_rng(world + 50)
    B, ev = 32, 16
    vocabs = [50, 7, 30]
    lookup_table = [0, 1, 2, 1]
    combiners = ["sum", "mean", "mean", "sum"]
    T, L = len(vocabs), len(lookup_table)
    tcfg = [ha.EmbeddingTableConfig(f"t{i}", v, ev) for i, v in enumerate(vocabs)]
    cfg = ha.EmbeddingCollectionConfig()
    for l in range(L):
        cfg.embedding_lookup(tcfg[lookup_table[l]], f"in{l}", f"out{l}", combiners[l])
    opt = {"sgd": _lib.OPT_SGD, "adagrad": _lib.OPT_ADAGRAD, "ftrl": _lib.OPT_FTRL}[opt_name]
    ftrl = (0.02, 0.05, 0.3)
    ranks = [DataParallelCollection(cfg, B, lr=0.1, optimizer=opt, scaler=2.0, epsilon=1e-6,
                                    max_hotness=4, ftrl=ftrl, rank=r, world=world, seed=4)
             for r in range(world)]
    assert all(torch.equal(ranks[0].table, e.table) for e in ranks)      # replica-uniform init
    row_start = np.concatenate([[0], np.cumsum(vocabs)[:-1]]).astype(np.int64)
    dense = ranks[0].table.cpu().numpy().copy()
    accum, ftrl_z = np.zeros_like(dense), np.zeros_like(dense)
    comb = [0 if c == "sum" else 1 for c in combiners]
    bpg = B // world
    for it in range(3):
        keys, br = _make_inputs(rng, B, vocabs, lookup_table, 4)
        gk, gbr = torch.from_numpy(keys).cuda(), torch.from_numpy(br).cuda()
        outs = [e.forward(gk, gbr) for e in ranks]                        # [bpg, L, ev]
        want = oracle.ebc_forward(B, lookup_table, ev, comb, keys, br, row_start, dense,
                                  num_gpus=world, batch_major=True)
        for d in range(world):
            assert_close(outs[d].cpu().numpy().reshape(-1), want[d], 1e-5, 1e-6, f"dp fwd rank{d}")
        grads = [rng.standard_normal(outs[d].shape).astype(np.float32) for d in range(world)]
        parts = [ranks[d].backward_local(torch.from_numpy(grads[d]).cuda()) for d in range(world)]
        total = sum(p[0] for p in parts)
        touched = sum(p[1] for p in parts)
        for e in ranks:
            e.apply_reduced(total.clone(), touched.clone())
        torch.cuda.synchronize()
        oracle.ebc_backward_update(B, lookup_table, ev, comb, keys, br, row_start, dense,
                                   np.stack([g.reshape(-1) for g in grads]),
                                   optimizer={"sgd": 0, "adagrad": 1, "ftrl": 2}[opt_name], lr=0.1,
                                   scaler=2.0, epsilon=1e-6, accum=accum, num_gpus=world,
                                   batch_major=True, ftrl=ftrl, ftrl_z=ftrl_z)
        assert all(torch.equal(ranks[0].table, e.table) for e in ranks), "replicas diverged"
        assert_close(ranks[0].table.cpu().numpy(), dense, 1e-5, 1e-6, f"dp tables it{it}")


@pytest.mark.parametrize("dtype", ["float16", "bfloat16"])
def test_data_parallel_average_with_16_bit_vectors_rounds_once(dtype):
    """replicated tables, Average lookups, 16-bit output: the fp32 sum is divided by the bucket's key
    count and rounded ONCE (multi_to_one_*_kernel through DPForward...MultiToOneDesc,
    model_forward.cu:29-66, generic_lookup.cuh:336-348), and the divided gradient stays fp32 on its
    way into the local reduce (AverageCombiner, data_parallel_embedding.cpp:226-243) -- i.e. the
    16-bit collection is the fp32 collection with its output rounded, and both take the same
    step from the same 16-bit gradient"""
    import torch
    import hugectr_amd as ha
    from hugectr_amd import _lib
    from hugectr_amd.embedding_collection import DataParallelCollection
    rng = np.random.default_rng(77)
    B, ev = 24, 16
    vocabs = [40, 9]
    lookup_table = [0, 1, 0]
    combiners = ["mean", "sum", "mean"]
    tcfg = [ha.EmbeddingTableConfig(f"t{i}", v, ev) for i, v in enumerate(vocabs)]
    cfg = ha.EmbeddingCollectionConfig()
    for l, t in enumerate(lookup_table):
        cfg.embedding_lookup(tcfg[t], f"in{l}", f"out{l}", combiners[l])
    tdt = getattr(torch, dtype)
    kw = dict(lr=0.1, optimizer=_lib.OPT_SGD, scaler=1.0, max_hotness=5, rank=0, world=1, seed=9)
    wide = DataParallelCollection(cfg, B, out_dtype=torch.float32, **kw)
    narrow = DataParallelCollection(cfg, B, out_dtype=tdt, **kw)
    assert torch.equal(wide.table, narrow.table)
    for it in range(2):
        keys, br = _make_inputs(rng, B, vocabs, lookup_table, 5)
        gk, gbr = torch.from_numpy(keys).cuda(), torch.from_numpy(br).cuda()
        a, b = wide.forward(gk, gbr), narrow.forward(gk, gbr)
        assert b.dtype == tdt and torch.equal(a.to(tdt), b), f"forward it{it}"
        g = torch.randn(b.shape, device="cuda").to(tdt)
        for e, gg in ((wide, g.float()), (narrow, g)):
            d, t = e.backward_local(gg)
            e.apply_reduced(d, t)
        assert torch.equal(wide.table, narrow.table), f"tables it{it}"


def test_static_table_ilookup_returns_vector_addresses():
    """ILookup::lookup(keys, num_keys, num_keys_per_table_offset, num_table_offset, table_id_list,
    float** embedding_vec) on a static table shard (ragged_static_embedding.cu:33-51): tables of
    different vector sizes in one flat array; every position's pointer must be the address of its
    vector, pooling through the pointers must equal pooling through row indices bit for bit, and
    positions that name a foreign table / an index outside the shard are flagged and get NULL."""
    import torch
    from hugectr_amd import _lib
    from hugectr_amd._lib import check, lib, ptr, stream_ptr
    from hugectr_amd.embedding_collection import static_lookup
    rng = np.random.default_rng(8)
    local_ids = [1, 4, 6]                  # this shard holds tables 1, 4 and 6 of the group
    rows, evs = [50, 7, 300], [8, 16, 4]
    idx_start = np.concatenate([[100], 100 + np.cumsum(rows)]).astype(np.int64)  # index numbering
    ev_off = np.concatenate([[0], np.cumsum(np.array(rows) * np.array(evs))[:-1]]).astype(np.int64)
    table = torch.randn(int(sum(r * e for r, e in zip(rows, evs))), device="cuda")
    # positions: table 4 x 20, table 1 x 33, table 6 x 41 (the call's own table order)
    order, counts = [4, 1, 6], [20, 33, 41]
    idx = np.concatenate([rng.integers(idx_start[local_ids.index(t)],
                                       idx_start[local_ids.index(t) + 1], c)
                          for t, c in zip(order, counts)]).astype(np.int64)
    offs = np.concatenate([[0], np.cumsum(counts)]).astype(np.int32)
    dev = lambda a, dt: torch.from_numpy(np.ascontiguousarray(a)).to("cuda", dt)
    args = (dev(offs, torch.int32), dev(order, torch.int32), dev(local_ids, torch.int32),
            dev(idx_start, torch.int64), table, dev(ev_off, torch.int64), dev(evs, torch.int32))
    ptrs, err = static_lookup(dev(idx, torch.int64), *args)
    assert int(err) == 0
    want, pos = [], 0
    for t, c in zip(order, counts):
        j = local_ids.index(t)
        for q in range(c):
            want.append(table.data_ptr() + 4 * (ev_off[j] + (idx[pos] - idx_start[j]) * evs[j]))
            pos += 1
    assert ptrs.cpu().tolist() == want
    # pooling through the pointers == pooling through rows (one table, ragged buckets)
    j = 2
    n, ev = 64, evs[j]
    lens = rng.integers(0, 5, 24)
    br = np.concatenate([[0], np.cumsum(lens)]).astype(np.int64)
    n = int(br[-1])
    ridx = rng.integers(0, rows[j], n).astype(np.int64)
    p2, err = static_lookup(dev(ridx + idx_start[j], torch.int64), dev([0, n], torch.int32),
                            dev([6], torch.int32), *args[2:])
    assert int(err) == 0
    out_p = torch.empty(24, ev, device="cuda")
    d_br, d_ridx = dev(br, torch.int64), dev(ridx, torch.int64)  # (kept alive across the launches)
    check(lib.hctr_forward_pool_ptrs(24, ev, 1, ptr(d_br), ptr(p2), ptr(out_p), _lib.F32,
                                     stream_ptr()))
    sub = table[ev_off[j]:ev_off[j] + rows[j] * ev].view(rows[j], ev)
    out_i = torch.empty(24, ev, device="cuda")
    check(lib.hctr_forward_pool(24, ev, 1, ptr(d_br), _lib.KEY_I64, ptr(d_ridx), ptr(sub),
                                ptr(out_i), _lib.F32, stream_ptr()))
    assert torch.equal(out_p, out_i)
    # a table this shard does not hold (bit 0) and an index beyond the shard (bit 1)
    bad, err = static_lookup(dev([idx_start[0], idx_start[1] + 10 ** 6], torch.int64),
                             dev([0, 1, 2], torch.int32), dev([5, 4], torch.int32), *args[2:])
    assert int(err) == 3 and bad.cpu().tolist() == [0, 0]


@pytest.mark.parametrize("world", [1, 2])
def test_multi_hot_concat_combiner(oracle, world):
    """Combiner::Concat with several keys per bucket (generic_lookup.cuh one_to_one_*; CPU
    reference reference_embedding.hpp:125-139, pinned in tests/test_ref_ebc_cpu.py for sum / average):
    the batch-major output of the lookup is max_hotness vectors side by side, key r in slot r,
    missing keys zero; the backward hands slot r's gradient to key r.  Checked against a plain
    numpy statement of those two rules, next to a sum and an average lookup on shared tables, on 1
    and 2 ranks (single process), with an SGD step."""
    import torch
    import hugectr_amd as ha
    from hugectr_amd import _lib
    from hugectr_amd.embedding_collection import (EmbeddingCollection, EmbeddingCollectionConfig,
                                                  EmbeddingTableConfig)
    rng = np.random.default_rng(2 + world)
    ev, B = 8, 16
    bpg = B // world
    rows = [40, 9]
    tabs = [EmbeddingTableConfig(f"t{i}", r, ev) for i, r in enumerate(rows)]
    cfg = EmbeddingCollectionConfig()
    spec = [(0, "concat", 3), (1, "sum", 4), (0, "mean", 2), (1, "concat", 2)]
    cfg.embedding_lookup(table_config=[tabs[t] for t, _, _ in spec],
                         bottom_name=[f"d{i}" for i in range(len(spec))], top_name="emb",
                         combiner=[c for _, c, _ in spec])
    cfg.shard(shard_matrix=[["t0", "t1"]] * world, shard_strategy=[("mp", ["t0", "t1"])])
    hot = [h for _, _, h in spec]
    shards = [EmbeddingCollection.for_rank(r, world, cfg, B, lr=0.5, optimizer=_lib.OPT_SGD,
                                           batch_major=True, max_hotness=max(hot), hotness=hot)
              for r in range(world)]
    # logical tables: key k of table t lives on shard k % world at local row k // world
    full = [rng.standard_normal((r, ev)).astype(np.float32) for r in rows]
    for r, e in enumerate(shards):
        for t in e.local_tables:
            ns = len(e.owners[t])
            sid = e.owners[t].index(r)
            n = -(-rows[t] // ns)
            blk = np.zeros((n, ev), np.float32)
            own = np.arange(sid, rows[t], ns)
            blk[:own.size] = full[t][own]
            s0 = e.row_start_of_table[t]
            e.table[s0:s0 + n] = torch.from_numpy(blk).cuda()
    lens = np.concatenate([rng.integers(0, h + 1, B) for h in hot])
    br = np.concatenate([[0], np.cumsum(lens)]).astype(np.int64)
    keys = np.concatenate([rng.integers(0, rows[t], int(lens[l * B:(l + 1) * B].sum()))
                           for l, (t, _, _) in enumerate(spec)]).astype(np.int64)
    widths = [ev * (h if c == "concat" else 1) for (_, c, h) in spec]
    off = np.concatenate([[0], np.cumsum(widths)])
    want = np.zeros((B, off[-1]), np.float32)
    for l, (t, c, h) in enumerate(spec):
        for b in range(B):
            ks = keys[br[l * B + b]:br[l * B + b + 1]]
            if c == "concat":
                for r, k in enumerate(ks):
                    want[b, off[l] + r * ev:off[l] + (r + 1) * ev] = full[t][k]
            elif len(ks):
                v = full[t][ks].sum(0, dtype=np.float32)
                want[b, off[l]:off[l] + ev] = v / np.float32(len(ks)) if c == "mean" else v
    gk, gbr = torch.from_numpy(keys).cuda(), torch.from_numpy(br).cuda()
    sends = [e.route_and_pool(gk, gbr) for e in shards]
    outs = []
    for dst, e in enumerate(shards):  # emulate the all-to-all: dst gets its sample slice
        parts = []
        for src, es in enumerate(shards):
            blk = sends[src].view(world, es.n_local, bpg, ev)[dst]
            parts.append(blk.reshape(-1, ev))
        outs.append(e.network_forward(torch.cat(parts)))
    got = torch.cat([o.reshape(bpg, -1) for o in outs]).float().cpu().numpy()
    np.testing.assert_allclose(got, want, rtol=1e-6, atol=1e-6)
    assert shards[0].L == sum(h if c == "concat" else 1 for _, c, h in spec)
    # backward + SGD: slot r's gradient reaches key r
    g = rng.standard_normal((B, off[-1])).astype(np.float32)
    ref = [f.copy() for f in full]
    acc = [np.zeros_like(f) for f in full]
    for l, (t, c, h) in enumerate(spec):
        for b in range(B):
            ks = keys[br[l * B + b]:br[l * B + b + 1]]
            for r, k in enumerate(ks):
                if c == "concat":
                    acc[t][k] += g[b, off[l] + r * ev:off[l] + (r + 1) * ev]
                else:
                    acc[t][k] += g[b, off[l]:off[l] + ev] / (len(ks) if c == "mean" else 1)
    for t in range(len(rows)):
        ref[t] -= 0.5 * acc[t]
    gt = torch.from_numpy(g).cuda()
    bsend = [e.network_backward(gt[d * bpg:(d + 1) * bpg].contiguous().view(bpg, e.L, ev))
             for d, e in enumerate(shards)]
    for own, e in enumerate(shards):  # the mirror all-to-all
        parts = []
        for d in range(world):
            base = sum(shards[d].n_local_of[:own]) * bpg
            parts.append(bsend[d][base:base + e.n_local * bpg])
        e.apply_gradients(torch.cat(parts))
    torch.cuda.synchronize()
    for r, e in enumerate(shards):
        for t in e.local_tables:
            ns = len(e.owners[t])
            own = np.arange(e.owners[t].index(r), rows[t], ns)
            s0 = e.row_start_of_table[t]
            np.testing.assert_allclose(e.table[s0:s0 + own.size].cpu().numpy(), ref[t][own],
                                       rtol=1e-5, atol=1e-5)


@pytest.mark.parametrize("batch_major", [False, True])
@pytest.mark.parametrize("max_hot", [1, 5])
@pytest.mark.parametrize("combiners", [["sum"] * 5, ["sum", "mean", "mean", "sum", "mean"]])
@pytest.mark.parametrize("opt_name,dtype", [("sgd", "float32"), ("adagrad", "bfloat16"),
                                            ("ftrl", "float16")])
def test_one_gpu_direct_path_equals_staged(monkeypatch, batch_major, max_hot, opt_name, dtype,
                                           combiners):
    """One GPU: pooling straight into the output (transposed store for batch-major) and
    the update reading the output's gradient in place must reproduce, bit for bit, the staged
    route -> pool -> network_forward / network_backward -> update path the reference runs
    (R/HugeCTR/embedding/model_parallel_embedding.cpp forward_per_gpu / backward_per_gpu) --
    Average lookups included: their receiver-side division (network_forward.cu:272-292)
    is applied in place to the pooled sums, its mirror to the gradient."""
    import torch
    import hugectr_amd as ha
    from hugectr_amd import _lib
    rng = np.random.default_rng(11 + max_hot)
    B, ev = 96, 32
    vocabs = [70, 9, 400]
    lookup_table = [0, 1, 2, 2, 0]
    tcfg = [ha.EmbeddingTableConfig(f"t{i}", v, ev) for i, v in enumerate(vocabs)]
    cfg = ha.EmbeddingCollectionConfig()
    for l, t in enumerate(lookup_table):
        cfg.embedding_lookup(tcfg[t], f"in{l}", f"out{l}", combiners[l])
    opt = {"sgd": _lib.OPT_SGD, "adagrad": _lib.OPT_ADAGRAD, "ftrl": _lib.OPT_FTRL}[opt_name]
    kw = dict(lr=0.1, optimizer=opt, scaler=4.0, epsilon=1e-6, batch_major=batch_major,
              max_hotness=max_hot, ftrl=(0.02, 0.05, 0.3), out_dtype=getattr(torch, dtype), seed=3)
    monkeypatch.setenv("HCTR_EBC_DIRECT", "0")
    staged = ha.EmbeddingCollection.for_rank(0, 1, cfg, B, **kw)
    monkeypatch.setenv("HCTR_EBC_DIRECT", "1")
    direct = ha.EmbeddingCollection.for_rank(0, 1, cfg, B, **kw)
    assert direct._direct and not staged._direct
    assert torch.equal(staged.table, direct.table)
    for step in range(3):
        if max_hot == 1:
            keys = np.concatenate([rng.integers(0, vocabs[t], size=B) for t in lookup_table]).astype(np.int64)
            br = np.arange(len(lookup_table) * B + 1, dtype=np.int64)
        else:
            keys, br = _make_inputs(rng, B, vocabs, lookup_table, max_hot)
        kt, brt = torch.from_numpy(keys).cuda(), torch.from_numpy(br).cuda()
        a, b = staged.forward(kt, brt), direct.forward(kt, brt)
        assert a.shape == b.shape and torch.equal(a, b), step
        g = torch.randn(a.shape, device="cuda").to(a.dtype)
        staged.backward_and_update(g)
        direct.backward_and_update(g)
        assert torch.equal(staged.table, direct.table), step
        if staged.accum is not None:
            assert torch.equal(staged.accum, direct.accum), step
    assert direct._direct_avg == ("mean" in combiners)


def test_forward_pool_mapped_rejects_bad_shapes():
    import torch
    from hugectr_amd import _lib
    ro = torch.arange(7, dtype=torch.int64, device="cuda")
    vi = torch.zeros(6, dtype=torch.int64, device="cuda")
    tab = torch.zeros((4, 8), device="cuda")
    out = torch.zeros((6, 8), device="cuda")
    rc = _lib.lib.hctr_forward_pool_mapped(6, 8, 0, _lib.ptr(ro), _lib.KEY_I64, _lib.ptr(vi), _lib.ptr(tab),
                                           _lib.ptr(out), _lib.F32, 0, 4, 2, None, _lib.stream_ptr())
    assert rc != 0  # samples * lookups != buckets
    u = __import__("ctypes").c_void_p()
    _lib.check(_lib.lib.hctr_updater_create(6, 4, 8, __import__("ctypes").byref(u)))
    _lib.check(_lib.lib.hctr_updater_set_grad_map(u, 3, 2))
    g = torch.zeros((6, 8), device="cuda")
    rc = _lib.lib.hctr_updater_update(u, 6, 6, _lib.ptr(ro), _lib.ptr(vi), _lib.ptr(g), _lib.F32,
                                      _lib.OPT_SGD, _lib.UPDATE_LOCAL, 0.1, 0.9, 0.999, 1e-7, 0.0, 1.0, 1,
                                      _lib.ptr(tab), None, None, _lib.stream_ptr())
    assert rc == 0  # hctr_updater_update is sum-only: the map is accepted
    _lib.check(_lib.lib.hctr_updater_set_grad_map(u, 4, 2))
    rc = _lib.lib.hctr_updater_update(u, 6, 6, _lib.ptr(ro), _lib.ptr(vi), _lib.ptr(g), _lib.F32,
                                      _lib.OPT_SGD, _lib.UPDATE_LOCAL, 0.1, 0.9, 0.999, 1e-7, 0.0, 1.0, 1,
                                      _lib.ptr(tab), None, None, _lib.stream_ptr())
    assert rc != 0  # 4 * 2 != 6 buckets
    _lib.lib.hctr_updater_destroy(u)


@pytest.mark.parametrize("seed", range(6))
def test_network_buffer_layout_is_the_reference_network_indices(seed):
    """Where NetworkForward finds the partial vectors of a lookup: the reference's own host code
    (NetworkIndices::init, R/HugeCTR/embedding/operators/network_forward.cu:23-62, compiled from the
    checkout into oracle/_ref/libref_network_indices.so) run on the local lookup lists of a random
    sharding, next to the block table this package builds (d_src_blocks: per lookup, per shard, the
    block of the received buffer = the source rank's first block + the lookup's index among that
    rank's local lookups).  Same sources for every lookup; their order inside a lookup is the
    owners' order here and whatever std::sort leaves in the reference (ties are not ordered there)."""
    import ctypes
    import hugectr_amd as ha
    path = os.path.join(os.path.dirname(__file__), "..", "oracle", "_ref", "libref_network_indices.so")
    if not os.path.exists(path):
        pytest.skip("oracle/_ref not built (needs the reference checkout)")
    R = ctypes.CDLL(path)
    P = ctypes.c_void_p
    R.refnet_indices.argtypes = [ctypes.c_int, P, P, P, P, P, P]
    rng = np.random.default_rng(seed)
    world = int(rng.choice([2, 3, 4, 8]))
    T = int(rng.integers(1, 7))
    L = int(rng.integers(T, T + 4))
    lookup_table = list(range(T))
    for _ in range(L - T):
        t = int(rng.integers(0, T))
        lookup_table.insert(int(rng.integers(lookup_table.index(t) + 1, len(lookup_table) + 1)), t)
    tcfg = [ha.EmbeddingTableConfig(f"t{i}", 50 + i, 8) for i in range(T)]
    cfg = ha.EmbeddingCollectionConfig()
    for l in range(L):
        cfg.embedding_lookup(tcfg[lookup_table[l]], f"in{l}", f"out{l}", "sum")
    sm = [[0] * T for _ in range(world)]
    for t in range(T):  # every table on a random non-empty set of ranks
        owners = np.flatnonzero(rng.random(world) < 0.4)
        if owners.size == 0:
            owners = np.array([int(rng.integers(0, world))])
        for g in owners:
            sm[int(g)][t] = 1
    cfg.shard(sm)
    e = ha.EmbeddingCollection.for_rank(0, world, cfg, world * 2, lr=0.1, max_hotness=2)
    # the ranks' local lookups in the order the package numbers them (= ascending lookup id)
    local = [[l for l in range(L) if r in e.owners[e.lookup_table[l]]] for r in range(world)]
    assert [len(x) for x in local] == list(e.n_local_of)
    offs = np.concatenate([[0], np.cumsum([len(x) for x in local])]).astype(np.int32)
    flat = np.array([l for x in local for l in x] + [0], dtype=np.int32)
    total = int(offs[-1])
    ids, gpus = np.full(total, -1, np.int32), np.full(total, -1, np.int32)
    noff, dst = np.full(L + 1, -1, np.int32), np.full(L, -1, np.int32)
    p = lambda a: a.ctypes.data_as(P)  # noqa: E731
    n_dst = R.refnet_indices(world, p(offs), p(flat), p(ids), p(gpus), p(noff), p(dst))
    assert n_dst == L and list(dst[:L]) == list(range(L))  # every lookup has at least one owner
    src = e.d_src_blocks.cpu().numpy().reshape(L, e.max_shards)
    base = np.concatenate([[0], np.cumsum(e.n_local_of)])
    for l in range(L):
        ref = sorted(int(base[gpus[i]] + ids[i]) for i in range(noff[l], noff[l + 1]))
        mine = sorted(int(b) for b in src[l] if b >= 0)
        assert mine == ref, (l, mine, ref)
        # ... and in this package the shards are added in ascending rank order
        assert [int(b) for b in src[l] if b >= 0] == sorted(mine)
