"""TEST INFRASTRUCTURE ONLY: randomized differential test of embedding_collection (static tables) --
the product's Python + the kernels' source under the host interpreter (tests/emu, fakecuda) against
the oracle's EBC reference, over random table / lookup / sharding layouts: 1, 2 or 4 ranks driven
from one process (the all-to-alls replayed with tensor copies, as tests/test_ebc_gpu.py does),
table-wise / row-wise / mixed sharding, tables shared by several lookups, sum / mean, feature- and
batch-major outputs, SGD / AdaGrad / Ftrl, ragged hotness with empty buckets.

    python tests/emu/fuzz_ebc.py --seed 0 --cases 100"""
import argparse
import os
import sys
import traceback

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
for p in (ROOT, os.path.join(ROOT, "tests"), HERE):
    if p not in sys.path:
        sys.path.insert(0, p)
os.environ["HCTR_EMU"] = "1"

import fakecuda  # noqa: E402

fakecuda.install(os.environ.get("HCTR_EMU_VARIANT"))

import torch  # noqa: E402

import hugectr_amd as ha  # noqa: E402
from hugectr_amd import _lib  # noqa: E402
from oracle import pyoracle as oracle  # noqa: E402
from util import assert_close  # noqa: E402


def make_inputs(rng, B, vocabs, lookup_table, max_hot, empty):
    L = len(lookup_table)
    lens = rng.integers(0 if max_hot > 1 else 1, max_hot + 1, size=L * B).astype(np.int64)
    lens[rng.random(L * B) < empty] = 0
    br = np.zeros(L * B + 1, np.int64)
    np.cumsum(lens, out=br[1:])
    keys = np.concatenate([rng.integers(0, vocabs[lookup_table[l]],
                                        size=int(lens[l * B:(l + 1) * B].sum()))
                           for l in range(L)] + [np.zeros(0, np.int64)]).astype(np.int64)
    return keys, br


def one_case(seed):
    rng = np.random.default_rng(seed)
    world = int(rng.choice([1, 1, 2, 4]))
    B = world * int(rng.choice([1, 2, 5, 8, 16]))
    ev = int(rng.choice([4, 8, 16, 32, 64, 128]))
    T = int(rng.integers(1, 6))
    vocabs = [int(rng.choice([1, 2, 7, 50, 300, 2000])) for _ in range(T)]
    L = int(rng.integers(T, T + 3))
    # (tables are numbered in the order of their first lookup, as the configuration numbers them)
    lookup_table = list(range(T))
    for _ in range(L - T):
        t = int(rng.integers(0, T))
        lookup_table.insert(int(rng.integers(lookup_table.index(t) + 1, len(lookup_table) + 1)), t)
    combiners = [str(rng.choice(["sum", "mean"])) for _ in range(L)]
    batch_major = bool(rng.integers(0, 2))
    opt_name = str(rng.choice(["sgd", "adagrad", "ftrl"]))
    max_hot = int(rng.choice([1, 2, 4, 9]))
    empty = float(rng.choice([0.0, 0.15, 0.7]))
    shard = str(rng.choice(["table", "row", "mixed"]))
    desc = dict(seed=seed, world=world, B=B, ev=ev, vocabs=vocabs, lookup_table=lookup_table,
                combiners=combiners, batch_major=batch_major, opt=opt_name, max_hot=max_hot,
                empty=empty, shard=shard)
    tcfg = [ha.EmbeddingTableConfig(f"t{i}", v, ev) for i, v in enumerate(vocabs)]
    cfg = ha.EmbeddingCollectionConfig()
    for l in range(L):
        cfg.embedding_lookup(tcfg[lookup_table[l]], f"in{l}", f"out{l}", combiners[l])
    if shard == "table":
        sm = [[1 if t % world == g else 0 for t in range(T)] for g in range(world)]
    elif shard == "row":
        sm = [[1] * T for _ in range(world)]
    else:  # every table on a random non-empty set of ranks
        sm = [[0] * T for _ in range(world)]
        for t in range(T):
            owners = np.flatnonzero(rng.random(world) < 0.5)
            if owners.size == 0:
                owners = np.array([int(rng.integers(0, world))])
            for g in owners:
                sm[int(g)][t] = 1
    cfg.shard(sm)
    opt = {"sgd": _lib.OPT_SGD, "adagrad": _lib.OPT_ADAGRAD, "ftrl": _lib.OPT_FTRL}[opt_name]
    ftrl = (0.02, 0.05, 0.3)
    ranks = [ha.EmbeddingCollection.for_rank(r, world, cfg, B, lr=0.1, optimizer=opt, scaler=2.0,
                                             epsilon=1e-6, batch_major=batch_major,
                                             max_hotness=max_hot, ftrl=ftrl)
             for r in range(world)]
    row_start = np.concatenate([[0], np.cumsum(vocabs)[:-1]]).astype(np.int64)
    dense = np.zeros((sum(vocabs), ev), np.float32)

    def shards():
        for t in range(T):
            owners = ranks[0].owners[t]
            for sid, g in enumerate(owners):
                e = ranks[g]
                s0 = e.row_start_of_table[t]
                ks = np.arange(sid, vocabs[t], len(owners))
                yield t, sid, e, s0, ks
    for t, sid, e, s0, ks in shards():
        dense[row_start[t] + ks] = e.table[s0:s0 + ks.size].cpu().numpy()
    accum, ftrl_z = np.zeros_like(dense), np.zeros_like(dense)
    bpg = B // world
    comb = [0 if c == "sum" else 1 for c in combiners]
    for it in range(2):
        keys, br = make_inputs(rng, B, vocabs, lookup_table, max_hot, empty)
        gk, gbr = torch.from_numpy(keys).cuda(), torch.from_numpy(br).cuda()
        if world == 1:
            outs = [ranks[0].forward_global(gk, gbr)]
        else:
            sends = [e.route_and_pool(gk, gbr) for e in ranks]
            outs = []
            for d, e in enumerate(ranks):
                blocks = []
                for s, es in enumerate(ranks):
                    if es.n_local:
                        blocks.append(sends[s].view(world, es.n_local, bpg, ev)[d].reshape(-1, ev))
                recv = torch.cat(blocks) if blocks else torch.empty((0, ev), device="cuda")
                outs.append(e.network_forward(recv.contiguous()))
        want = oracle.ebc_forward(B, lookup_table, ev, comb, keys, br, row_start, dense,
                                  num_gpus=world, batch_major=batch_major)
        for d in range(world):
            assert_close(outs[d].cpu().numpy().reshape(-1), want[d], 1e-5, 2e-5,
                         f"{desc} fwd rank{d} it{it}")
        grads = [rng.standard_normal(tuple(outs[d].shape)).astype(np.float32) for d in range(world)]
        if world == 1:
            ranks[0].backward_and_update(torch.from_numpy(grads[0]).cuda())
        else:
            bsends = [ranks[d].network_backward(torch.from_numpy(grads[d]).cuda())
                      for d in range(world)]
            for s, es in enumerate(ranks):
                if es.n_local == 0:
                    continue
                base = sum(ranks[0].n_local_of[:s])
                tops = [bsends[d].view(-1, bpg, ev)[base:base + es.n_local] for d in range(world)]
                es.apply_gradients(torch.stack(tops).contiguous())
        oracle.ebc_backward_update(B, lookup_table, ev, comb, keys, br, row_start, dense,
                                   np.stack([g.reshape(-1) for g in grads]),
                                   optimizer={"sgd": 0, "adagrad": 1, "ftrl": 2}[opt_name], lr=0.1,
                                   scaler=2.0, epsilon=1e-6, accum=accum, num_gpus=world,
                                   batch_major=batch_major, ftrl=ftrl, ftrl_z=ftrl_z)
        for t, sid, e, s0, ks in shards():
            assert_close(e.table[s0:s0 + ks.size].cpu().numpy(), dense[row_start[t] + ks], 1e-4,
                         1e-5, f"{desc} table {t} shard {sid} it{it}")
    return desc


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--seed", type=int, default=0)
    ap.add_argument("--cases", type=int, default=50)
    a = ap.parse_args()
    bad = 0
    for i in range(a.cases):
        seed = a.seed * 1_000_003 + i
        try:
            one_case(seed)
        except Exception as e:  # noqa: BLE001
            bad += 1
            print(f"FAIL seed {seed}: {type(e).__name__} {str(e)[:500]}", flush=True)
            if os.environ.get("FUZZ_TRACE"):
                traceback.print_exc()
    print(f"{a.cases - bad} / {a.cases} cases agree with the oracle", flush=True)
    sys.exit(1 if bad else 0)


if __name__ == "__main__":
    main()
