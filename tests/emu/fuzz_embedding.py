"""TEST INFRASTRUCTURE ONLY: randomized differential test of the legacy embedding path -- the
kernels' source under the host interpreter (tests/emu) against the oracle, over shapes no
parametrized test enumerates (odd vector sizes, one sample, one slot, vocabularies of a single
key, batches that alternate one-hot / ragged / empty, every optimizer, 16-bit outputs).

    python tests/emu/fuzz_embedding.py --seed 0 --cases 200 [--variant tools/wip/<name>]

Prints one line per failing case (its seed reproduces it) and a summary; exit code 1 on failure."""
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

import emu  # noqa: E402
from oracle import pyoracle as orc  # noqa: E402
from util import assert_close, make_csr  # noqa: E402

OPTS = [
    ("sgd", dict(optimizer=6, atomic_update=0)),
    ("adam_local", dict(optimizer=1, update_type=0)),
    ("adam_global", dict(optimizer=1, update_type=1)),
    ("adam_lazy", dict(optimizer=1, update_type=2)),
    ("adagrad", dict(optimizer=3)),
    ("momentum_local", dict(optimizer=5, update_type=0, momentum_factor=0.9)),
    ("momentum_global", dict(optimizer=5, update_type=1, momentum_factor=0.9)),
    ("nesterov_local", dict(optimizer=4, update_type=0, momentum_factor=0.9)),
    ("nesterov_global", dict(optimizer=4, update_type=1, momentum_factor=0.9)),
]
ORC_OPT = {1: orc.OPT_ADAM, 3: orc.OPT_ADAGRAD, 5: orc.OPT_MOMENTUM, 4: orc.OPT_NESTEROV,
           6: orc.OPT_SGD}


def one_case(lib, _lib, seed):
    rng = np.random.default_rng(seed)
    key_bytes = int(rng.choice([8, 4]))
    D = int(rng.choice([1, 2, 4, 6, 8, 11, 16, 32, 64, 128, 256]))
    combiner = int(rng.integers(0, 2))
    B = int(rng.choice([1, 2, 3, 7, 31, 64, 65, 150]))
    S = int(rng.choice([1, 2, 3, 5, 8, 13, 26]))
    hot = int(rng.choice([1, 2, 3, 8, 40]))
    vps = int(rng.choice([1, 2, 5, 40, 500]))
    dt = str(rng.choice(["f32", "f32", "f16", "bf16"]))
    name, kw = OPTS[int(rng.integers(0, len(OPTS)))]
    scaler = float(rng.choice([1.0, 4.0, 1024.0]))
    steps = int(rng.integers(1, 4))
    # the shard of one rank of N (every rank sees the full-batch CSR and filters its share:
    # localized = the slots s with s % N == rank, distributed = the keys k with k % N == rank)
    world = int(rng.choice([1, 1, 2, 3, 4, 8]))
    rank = int(rng.integers(0, world))
    localized = bool(rng.integers(0, 2)) or world == 1
    if world > 1:
        B = B * world
        if not localized:
            combiner = 0  # (mean on a distributed embedding divides after the reduce-scatter:
                          #  tests/test_embedding_gpu.py::test_distributed_mean_divides_...)
    desc = dict(seed=seed, key_bytes=key_bytes, D=D, combiner=combiner, B=B, S=S, hot=hot, vps=vps,
                dt=dt, opt=name, scaler=scaler, steps=steps, world=world, rank=rank,
                localized=localized)
    # the hot-row path of one-hot batches (rows below HCTR_HOT_ROWS leave the sort; read when the
    # handle is created): forced on for a share of the cases, with few enough hot rows that a
    # batch has both kinds
    hot_min = str(rng.choice(["", "", "1"]))
    hot_rows = str(rng.choice(["8192", "2", "16", "300"]))
    for k, v in (("HCTR_HOT_MIN", hot_min), ("HCTR_HOT_ROWS", hot_rows)):
        if v and hot_min:
            os.environ[k] = v
        else:
            os.environ.pop(k, None)
    desc.update(hot_min=hot_min, hot_rows=hot_rows if hot_min else "")
    V = S * vps + int(rng.integers(0, 20))
    kd = np.int64 if key_bytes == 8 else np.uint32
    opt = dict(lr=0.05, scaler=scaler, beta1=0.9, beta2=0.999, epsilon=1e-7, **kw)
    emb = emu.Embedding(lib, _lib.EMB_LOCALIZED if localized else _lib.EMB_DISTRIBUTED, B, V, D,
                        S * hot, S, combiner, opt, key_dtype=kd,
                        out_dtype={"f32": 0, "f16": 1, "bf16": 2}[dt], rank=rank, world=world)
    s_r = orc.slots_on_gpu(S, rank, world) if localized else S

    def shard(ro, keys):
        if world == 1:
            return ro, keys
        f = orc.localized_filter if localized else orc.distributed_filter
        return f(ro, keys, B, S, rank, world)
    table = emb.table().copy()
    ns = {1: 2, 3: 1, 5: 1, 4: 1, 6: 0}[kw["optimizer"]]
    s0 = np.zeros_like(table) if ns >= 1 else None
    s1 = np.zeros_like(table) if ns >= 2 else None
    pt = np.ones(table.shape, dtype=np.uint64) if name == "adam_lazy" else None
    ht = orc.HashTable(V, key_bytes)
    half_state = dt == "f16" and ns > 0  # (fp16 embeddings keep fp16-valued state: own GPU test)
    for it in range(steps):
        kind = int(rng.integers(0, 4))  # one-hot / ragged / ragged with many empties / all empty
        ro, keys = make_csr(rng, B, S, hot, vps, empty_frac=[0.0, 0.2, 0.9, 1.0][kind],
                            one_hot=(kind == 0))
        rok, kk = ro.astype(kd), keys.astype(kd)
        out = emb.forward(True, rok, kk)
        ro, keys = shard(ro, keys)
        vi = ht.get_insert(keys)
        assert (emb.value_index(keys.size) == vi).all(), (desc, it, "rows")
        if dt == "f32":
            want = orc.forward(ro, vi, table, D, combiner)
            got = out.reshape(-1, D)
            assert (got.view(np.uint32) == want.view(np.uint32)).all(), (desc, it, "forward")
        else:
            want = orc.forward_mixed(ro, vi, table, D, combiner, dt)
            if dt == "f16":
                got = out.reshape(-1, D).astype(np.float32)
            else:
                got = (out.reshape(-1, D).astype(np.uint32) << 16).view(np.float32)
            assert (got.view(np.uint32) == want.view(np.uint32)).all(), (desc, it, "forward16")
        g = (rng.standard_normal((B * s_r, D)) * 2).astype(np.float32)
        if dt == "f32":
            gg, wg = g, orc.backward(ro, g, D, combiner)
        else:
            wg = orc.backward_mixed(ro, g, D, combiner, dt)
            g16 = orc.round_to(g, dt)
            gg = g16.astype(np.float16) if dt == "f16" else (g16.view(np.uint32) >> 16).astype(np.uint16)
        emb.backward(gg.reshape(B, s_r, D))
        emb.update_params()
        if half_state:  # (state rounding has its own GPU test: rows / forward stay checked)
            table[...] = emb.table()
            continue
        o = orc.OptParamsC()
        o.optimizer, o.update_type, o.lr = ORC_OPT[kw["optimizer"]], kw.get("update_type", 0), 0.05
        o.beta1, o.beta2, o.epsilon = 0.9, 0.999, 1e-7
        o.momentum_factor, o.scaler, o.times = kw.get("momentum_factor", 0.0), scaler, it + 1
        orc.update_params(ro, vi, wg, o, table, s0, s1, pt)
        # (1e-3 = the north star's bound; the absolute part scales with the array: an element whose
        #  gradient sum cancels carries the rounding of terms far larger than itself)
        def atol(a):
            return 1e-5 + 1e-4 * float(np.abs(a).max())
        # (tables: an Adam step is lr * m / (sqrt(v) + eps) whatever the gradient's size)
        assert_close(emb.table(), table, 1e-3, max(atol(table), 1e-3 * 0.05), f"{desc} table it{it}")
        if s0 is not None:
            assert_close(emb.opt_state(0), s0, 1e-3, atol(s0), f"{desc} state0 it{it}")
        if s1 is not None:
            assert_close(emb.opt_state(1), s1, 1e-3, atol(s1), f"{desc} state1 it{it}")
        # the kernels and the oracle go on from the SAME numbers (no drift across steps)
        table[...] = emb.table()
        if s0 is not None:
            s0[...] = emb.opt_state(0)
        if s1 is not None:
            s1[...] = emb.opt_state(1)
    if half_state:
        return desc
    # evaluation batch: unseen keys read as zeros and still count in the mean
    ro, keys = make_csr(rng, B, S, hot, vps * 2 + 1, one_hot=bool(rng.integers(0, 2)))
    out = emb.forward(False, ro.astype(kd), keys.astype(kd))
    ro, keys = shard(ro, keys)
    vi = ht.get_mark(keys)
    if dt == "f32":
        want = orc.forward(ro, vi, table, D, combiner)
        assert (out.reshape(-1, D).view(np.uint32) == want.view(np.uint32)).all(), (desc, "eval")
    return desc


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--seed", type=int, default=0)
    ap.add_argument("--cases", type=int, default=100)
    ap.add_argument("--variant", default=None)
    a = ap.parse_args()
    lib = emu.load(a.variant, os.path.basename(os.path.normpath(a.variant))) if a.variant else emu.load()
    _lib = emu.bind(lib)
    bad = 0
    for i in range(a.cases):
        seed = a.seed * 1_000_003 + i
        try:
            one_case(lib, _lib, seed)
        except Exception as e:  # noqa: BLE001
            bad += 1
            msg = str(e).replace("\n", " ")[:400]
            print(f"FAIL seed {seed}: {type(e).__name__} {msg}", flush=True)
            if os.environ.get("FUZZ_TRACE"):
                traceback.print_exc()
    print(f"{a.cases - bad} / {a.cases} cases agree with the oracle", flush=True)
    sys.exit(1 if bad else 0)


if __name__ == "__main__":
    main()
