"""GPU parity: embedding_collection path (key routing -> keys_to_indices -> pooled lookup ->
network forward / backward -> static-table optimizer) vs the CPU restatement of the reference's
EmbeddingReferenceCPU (R/test/utest/embedding_collection/reference_embedding.hpp:32-237)."""
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


def test_keys_to_indices_bit_exact(oracle):
    import torch
    from hugectr_amd import _lib
    rng = np.random.default_rng(0)
    keys = rng.integers(0, 10**9, size=10000).astype(np.int64)
    out = torch.empty(keys.size, dtype=torch.int64, device="cuda")
    for start, ns in ((0, 1), (12345, 8), (7, 3)):
        _lib.check(_lib.lib.hctr_ebc_keys_to_indices(_lib.ptr(torch.from_numpy(keys).cuda()), _lib.KEY_I64,
                                                     keys.size, start, ns, _lib.ptr(out), _lib.stream_ptr()))
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
