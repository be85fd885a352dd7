"""Stage-by-stage timing of the sparse-embedding hot path at the DLRM Criteo-1TB shape.
Usage: python tools/microbench_embedding.py [--batch 65536] [--scale 1.0] [--alpha 1.1]"""
import argparse
import json
import os
import sys
import time

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import hugectr_amd as ha  # noqa: E402
from hugectr_amd import _lib  # noqa: E402

# R/test/embedding_collection_test/dgx_a100_one_hot.py:24-51 (Criteo-1TB slot sizes)
CRITEO_1TB = [39884406, 39043, 17289, 7420, 20263, 3, 7120, 1543, 63, 38532951, 2953546, 403346,
              10, 2208, 11938, 155, 4, 976, 14, 39979771, 25641295, 39664984, 585935, 12972, 108,
              36]


def powerlaw(rng, n, vocab, alpha):
    if alpha <= 0:
        return rng.integers(0, vocab, size=n).astype(np.int64)
    u = rng.random(n, dtype=np.float32).astype(np.float64)
    a = 1.0 - alpha
    y = ((float(vocab) ** a - 1.0) * u + 1.0) ** (1.0 / a)
    return np.clip(np.round(y) - 1, 0, vocab - 1).astype(np.int64)


def make_batch(rng, B, sizes, alpha):
    S = len(sizes)
    offs = np.concatenate([[0], np.cumsum(sizes)[:-1]]).astype(np.int64)
    keys = np.empty((B, S), dtype=np.int64)
    for s, v in enumerate(sizes):
        keys[:, s] = powerlaw(rng, B, v, alpha) + offs[s]
    return keys.reshape(-1)


def timeit(fn, iters=20, warm=3):
    for _ in range(warm):
        fn()
    torch.cuda.synchronize()
    ev = [torch.cuda.Event(enable_timing=True) for _ in range(iters + 1)]
    ev[0].record()
    for i in range(iters):
        fn()
        ev[i + 1].record()
    torch.cuda.synchronize()
    ts = sorted(ev[i].elapsed_time(ev[i + 1]) for i in range(iters))
    return ts[len(ts) // 2], ts[0]


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--batch", type=int, default=65536)
    ap.add_argument("--scale", type=float, default=1.0)
    ap.add_argument("--alpha", type=float, default=1.1)
    ap.add_argument("--dim", type=int, default=128)
    ap.add_argument("--nbatches", type=int, default=4)
    a = ap.parse_args()
    sizes = [max(1, int(v * a.scale)) for v in CRITEO_1TB]
    V, B, S, D = sum(sizes), a.batch, len(sizes), a.dim
    rng = np.random.default_rng(1234)
    res = {"vocab_rows": V, "batch": B, "slots": S, "D": D, "alpha": a.alpha,
           "table_GiB": V * D * 4 / 2**30}
    t0 = time.time()
    emb = ha.SparseEmbeddingHash(_lib.EMB_LOCALIZED, B, 0, V, D, S, S, 0,
                                 ha.OptParams(optimizer=_lib.OPT_SGD, lr=0.01, atomic_update=False),
                                 slot_size_array=sizes)
    emb.init_params()
    torch.cuda.synchronize()
    res["create_s"] = time.time() - t0
    ro = torch.arange(0, B * S + 1, dtype=torch.int64, device="cuda")
    batches = [torch.from_numpy(make_batch(rng, B, sizes, a.alpha)).cuda() for _ in range(a.nbatches)]
    out = torch.empty((B, S, D), dtype=torch.float32, device="cuda")
    grad = torch.randn((B, S, D), dtype=torch.float32, device="cuda")
    # cold: first sight of every batch (inserts)
    t0 = time.time()
    for kb in batches:
        emb.forward(True, ro, kb, out=out)
    torch.cuda.synchronize()
    res["cold_forward_ms_per_batch"] = (time.time() - t0) * 1e3 / len(batches)
    res["unique_rows"] = emb.get_vocabulary_size()
    i = [0]

    def fwd():
        emb.forward(True, ro, batches[i[0] % len(batches)], out=out)
        i[0] += 1

    res["forward_ms(med,min)"] = timeit(fwd)

    # isolate stages through the stateless ABI
    vi = emb.value_index(B * S)
    table = emb.table()

    def pool():
        _lib.check(_lib.lib.hctr_forward_pool(B * S, D, 0, _lib.ptr(ro), _lib.KEY_I64, _lib.ptr(vi),
                                              _lib.ptr(table), _lib.ptr(out), _lib.F32,
                                              _lib.stream_ptr()))

    med, mn = timeit(pool)
    nnz = B * S
    alg = nnz * 8 + nnz * 8 + nnz * D * 4 + B * S * D * 4
    res["pool_ms(med,min)"] = (med, mn)
    res["pool_alg_GBps"] = alg / (med * 1e-3) / 1e9
    res["pool_alg_bytes"] = alg

    def copy():
        out.copy_(grad)

    med, mn = timeit(copy)
    res["d2d_copy_GBps(read+write)"] = 2 * out.numel() * 4 / (med * 1e-3) / 1e9

    def step():
        emb.forward(True, ro, batches[i[0] % len(batches)], out=out)
        i[0] += 1
        emb.backward(grad)
        emb.update_params()

    res["fwd+update_sorted_ms(med,min)"] = timeit(step, iters=10)
    emb2 = None
    del emb
    torch.cuda.empty_cache()
    emb = ha.SparseEmbeddingHash(_lib.EMB_LOCALIZED, B, 0, V, D, S, S, 0,
                                 ha.OptParams(optimizer=_lib.OPT_SGD, lr=0.01, atomic_update=True),
                                 slot_size_array=sizes)
    emb.init_params()
    for kb in batches:
        emb.forward(True, ro, kb, out=out)
    res["fwd+update_atomic_ms(med,min)"] = timeit(step, iters=10)
    print(json.dumps(res))


if __name__ == "__main__":
    main()
