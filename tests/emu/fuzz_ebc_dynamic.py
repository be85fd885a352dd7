"""TEST INFRASTRUCTURE ONLY: randomized differential test of embedding_collection on DYNAMIC tables
with the flat row store (hctr_det_row_store: the static tables' gather and, for SGD, their sort +
segmented reduce on the table-wide row numbers) against the pointer-per-key / unique-key flow
(HCTR_DYNAMIC_FLAT=0, the one the oracle tests pin) -- the product's Python and the kernels' source
under the host interpreter.  Random tables / lookups sharing tables, sum and mean combiners, ragged
and empty buckets, both output layouts and dtypes, the one-GPU direct and the staged path, tiny
initial capacities (the classes grow and the store moves while training), SGD and optimizers with a
state table (which keep the unique-key flow behind the flat gather).  Pooled vectors and exported
tables must agree BIT FOR BIT.

    python tests/emu/fuzz_ebc_dynamic.py --seed 0 --cases 40"""
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


def one_case(seed):
    rng = np.random.default_rng(seed)
    B = int(rng.choice([1, 3, 8, 32, 70]))
    ev = int(rng.choice([4, 8, 16, 32, 128]))
    T = int(rng.integers(1, 5))
    vocabs = [int(rng.choice([2, 9, 300, 5000, 10 ** 7])) for _ in range(T)]
    L = int(rng.integers(T, T + 3))
    lookup_table = list(range(T))
    for _ in range(L - T):
        t = int(rng.integers(0, T))
        lookup_table.insert(int(rng.integers(lookup_table.index(t) + 1, len(lookup_table) + 1)), t)
    combiners = [str(rng.choice(["sum", "sum", "mean"])) for _ in range(L)]
    batch_major = bool(rng.integers(0, 2))
    dtype = [torch.float32, torch.float16, torch.bfloat16][int(rng.integers(0, 3))]
    opt_name = str(rng.choice(["sgd", "adagrad", "adam", "momentum"]))
    max_hot = int(rng.choice([1, 2, 5, 9]))
    empty = float(rng.choice([0.0, 0.2, 0.7]))
    direct = str(rng.choice(["0", "1"]))
    cap0 = int(rng.choice([4, 16, 64]))
    desc = dict(seed=seed, B=B, ev=ev, vocabs=vocabs, lookup_table=lookup_table, combiners=combiners,
                batch_major=batch_major, dtype=str(dtype), opt=opt_name, max_hot=max_hot, empty=empty,
                direct=direct, cap0=cap0)
    tcfg = [ha.EmbeddingTableConfig(f"t{i}", -1, ev) for i in range(T)]
    cfg = ha.EmbeddingCollectionConfig()
    for l in range(L):
        cfg.embedding_lookup(tcfg[lookup_table[l]], f"in{l}", f"out{l}", combiners[l])
    opt = {"sgd": _lib.OPT_SGD, "adagrad": _lib.OPT_ADAGRAD, "adam": _lib.OPT_ADAM,
           "momentum": _lib.OPT_MOMENTUM_SGD}[opt_name]
    kw = dict(lr=0.05, optimizer=opt, scaler=float(rng.choice([1.0, 8.0])), epsilon=1e-6,
              batch_major=batch_major, max_hotness=max_hot, out_dtype=dtype, seed=seed % 97,
              storage="dynamic", initializer="", init_capacity=cap0)
    os.environ["HCTR_EBC_DIRECT"] = direct
    os.environ["HCTR_DYNAMIC_FLAT"] = "0"
    ptrs = ha.EmbeddingCollection.for_rank(0, 1, cfg, B, **kw)
    os.environ["HCTR_DYNAMIC_FLAT"] = "1"
    flat = ha.EmbeddingCollection.for_rank(0, 1, cfg, B, **kw)
    assert flat._dyn_flat and not ptrs._dyn_flat, desc
    for step in range(int(rng.integers(2, 5))):
        lens = rng.integers(0, max_hot + 1, size=L * B).astype(np.int64)
        lens[rng.random(L * B) < empty] = 0
        br = np.zeros(L * B + 1, np.int64)
        np.cumsum(lens, out=br[1:])
        keys = np.concatenate([rng.integers(0, vocabs[lookup_table[l]],
                                            size=int(lens[l * B:(l + 1) * B].sum()))
                               for l in range(L)] + [np.zeros(0, np.int64)]).astype(np.int64)
        kt, brt = torch.from_numpy(keys).cuda(), torch.from_numpy(br).cuda()
        a, b = ptrs.forward(kt, brt), flat.forward(kt, brt)
        assert a.shape == b.shape and torch.equal(a, b), (desc, step, "forward")
        g = torch.from_numpy(rng.standard_normal(tuple(a.shape)).astype(np.float32)).cuda().to(a.dtype)
        ptrs.backward_and_update(g)
        flat.backward_and_update(g)
    assert ptrs.det.size() == flat.det.size(), desc
    for c in range(len(flat.det.dims)):
        (ka, va), (kb, vb) = ptrs.det.export(c), flat.det.export(c)
        oa, ob = torch.argsort(ka), torch.argsort(kb)
        assert torch.equal(ka[oa], kb[ob]) and torch.equal(va[oa], vb[ob]), (desc, "class", c)
    return desc


if __name__ == "__main__":
    ap = argparse.ArgumentParser()
    ap.add_argument("--seed", type=int, default=0)
    ap.add_argument("--cases", type=int, default=40)
    a = ap.parse_args()
    ok = 0
    for s in range(a.seed, a.seed + a.cases):
        try:
            one_case(s)
            ok += 1
        except Exception:
            print("seed", s)
            traceback.print_exc()
    print(f"{ok} / {a.cases} cases agree")
    sys.exit(0 if ok == a.cases else 1)
