"""TEST INFRASTRUCTURE ONLY: randomized differential test of the HIP kernels' source (tests/emu build
of hugectr_amd/csrc, through hctr_emb_*) against the REFERENCE'S DEVICE CODE executed by the same
interpreter (oracle/_ref/libref_gpu_kernels.so + libref_hashtable.so: its GPU hash table, filter
kernels, forward kernels, backward kernels and EmbeddingOptimizer::update) -- shapes, key widths,
rank shards, optimizers and step counts that tests/test_ref_gpu_kernels_cpu.py does not enumerate.

    python tests/emu/fuzz_ref_kernels.py --seed 0 --cases 100

Per case: one rank of a random world runs several training steps on both sides; compared are the
row of every key (hash table), the pooled vectors (bit for bit), the tables and the optimizer
state after every update.  Prints one line per failing case (its seed reproduces it)."""
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
import test_ref_gpu_kernels_cpu as T  # noqa: E402
import test_ref_hashtable_cpu as H  # noqa: E402
from util import assert_close, make_csr  # noqa: E402

OPTS = [
    ("sgd", dict(optimizer=6, atomic_update=0), dict(optimizer=6), 0),
    ("adam_local", dict(optimizer=1, update_type=0), dict(optimizer=1, update_type=0), 2),
    ("adam_global", dict(optimizer=1, update_type=1), dict(optimizer=1, update_type=1), 2),
    ("adagrad", dict(optimizer=3), dict(optimizer=3), 1),
    ("momentum_local", dict(optimizer=5, update_type=0, momentum_factor=0.7),
     dict(optimizer=5, update_type=0, mu=0.7), 1),
    ("momentum_global", dict(optimizer=5, update_type=1, momentum_factor=0.7),
     dict(optimizer=5, update_type=1, mu=0.7), 1),
    ("nesterov_local", dict(optimizer=4, update_type=0, momentum_factor=0.6),
     dict(optimizer=4, update_type=0, mu=0.6), 1),
    ("nesterov_global", dict(optimizer=4, update_type=1, momentum_factor=0.6),
     dict(optimizer=4, update_type=1, mu=0.6), 1),
]


def one_case(lib, ref, seed):
    from hugectr_amd import _lib
    rng = np.random.default_rng(seed)
    world = int(rng.choice([1, 1, 2, 3, 4]))
    rank = int(rng.integers(0, world))
    dist = bool(rng.integers(0, 2)) and world > 1
    B = world * int(rng.choice([1, 2, 5, 9]))
    S = int(rng.choice([1, 2, 5, 9, 26]))
    hot = int(rng.choice([1, 1, 2, 6]))
    D = int(rng.choice([1, 3, 4, 11, 16, 32, 64, 128]))
    vps = int(rng.choice([1, 7, 60]))
    combiner = int(rng.integers(0, 2))
    fp16 = int(rng.integers(0, 2))
    kb = int(rng.choice([8, 4]))
    name, hip_kw, ref_kw, ns = OPTS[int(rng.integers(0, len(OPTS)))]
    V = S * vps + 4
    kdt = np.int64 if kb == 8 else np.uint32
    what = (f"world {world} rank {rank} {'dist' if dist else 'loc'} B{B} S{S} hot{hot} D{D} vps{vps} "
            f"comb{combiner} fp16={fp16} kb{kb} {name}")
    opt = dict(lr=0.05, scaler=4.0, beta1=0.9, beta2=0.999, epsilon=1e-7, **hip_kw)
    emb = emu.Embedding(lib, _lib.EMB_DISTRIBUTED if dist else _lib.EMB_LOCALIZED, B, V, D,
                        max(S * hot, 1), S, combiner, opt, key_dtype=kdt, out_dtype=1 if fp16 else 0,
                        rank=rank, world=world)
    spr = S if dist else emb.slots_on_rank
    t_ref = emb.table().copy()
    sdt = np.float16 if fp16 else np.float32
    r0 = np.zeros((V, D), sdt) if ns >= 1 else None
    r1 = np.zeros((V, D), sdt) if ns >= 2 else None
    ht = H.RefTable(V, kb)
    steps = 0  # updates made so far (adam.times)
    try:
        for it in range(int(rng.integers(1, 4))):
            ro, keys = make_csr(rng, B, S, hot, vps, one_hot=bool(rng.integers(0, 2)))
            out = emb.forward(True, ro.astype(kdt), keys.astype(kdt))
            fro, fkeys = T._ref_filter(ref, kb, 1 if dist else 0, B, S, rank, world, ro, keys)
            if spr == 0:
                continue
            if fkeys.size == 0:
                # (no key for this rank: the reference's update() reads hash_value_flag_sumed[-1]
                #  then -- undefined --, so neither side is stepped on such a batch)
                assert not out.astype(np.float32).any(), what
                continue
            vi = emb.value_index(fkeys.size).copy()
            assert np.array_equal(vi, ht.get_insert(fkeys)), f"rows it{it}: {what}"
            # distributed + mean on N > 1 GPUs: partial SUMS leave the rank (reduce-scatter, then
            # forward_scale), and backward divides by the bucket's key count over ALL GPUs, i.e.
            # by the full-batch row offsets (distributed_slot_sparse_embedding_hash.hpp:162-221)
            part = dist and combiner == 1 and world > 1
            # (pooled from the HIP side's own table: the two tables agree to rounding after an
            #  update, bit-equal pooling needs bit-equal rows)
            want = ref.forward(kb, fp16, 0 if part else combiner, B, spr, D, fro, vi,
                               emb.table().copy())
            bits = np.uint16 if fp16 else np.uint32
            assert np.array_equal(out.reshape(-1, D).view(bits), want.view(bits)), \
                f"forward it{it}: {what}"
            top = rng.standard_normal((B, spr, D)).astype(sdt)
            emb.backward(top)
            emb.update_params()
            wg = ref.backward(kb, fp16, combiner, B, spr, D, ro if part else fro,
                              top.reshape(B * spr, D))
            steps += 1
            ref.update(kb, fp16, dict(ref_kw, lr=0.05, scaler=4.0, times=steps), B, spr, D, V, fro,
                       vi, wg, t_ref, r0, r1, None)
            rt, at = (4e-3, 2e-5) if (fp16 and ns) else (2e-5, 2e-6)
            assert_close(emb.table(), t_ref, rt, at, f"table it{it}: {what}")
            if ns >= 1:
                assert_close(emb.opt_state(0), r0.astype(np.float32), rt, at, f"state0 it{it}: {what}")
            if ns >= 2:
                assert_close(emb.opt_state(1), r1.astype(np.float32), rt, max(at, 1e-7),
                             f"state1 it{it}: {what}")
    finally:
        ht.close()
    return what


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--seed", type=int, default=0)
    ap.add_argument("--cases", type=int, default=100)
    a = ap.parse_args()
    lib = emu.load_under_test()
    ref = T.RefGpu()
    bad = 0
    for i in range(a.cases):
        seed = a.seed * 100003 + i
        try:
            one_case(lib, ref, seed)
        except Exception as e:  # noqa: BLE001
            bad += 1
            print(f"FAIL seed {seed}: {e!r}"[:600])
            if os.environ.get("FUZZ_TRACE"):
                traceback.print_exc()
    print(f"{a.cases - bad} of {a.cases} cases agree")
    sys.exit(1 if bad else 0)


if __name__ == "__main__":
    main()
