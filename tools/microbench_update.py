"""Sparse-update-only loop at the DLRM Criteo-1TB shape (run under rocprofv3 --kernel-trace --stats).
Usage: python tools/microbench_update.py [--alpha 1.1] [--iters 20] [--dtype bf16|f32]"""
import argparse
import os
import sys

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import hugectr_amd as ha  # noqa: E402
from hugectr_amd import _lib  # noqa: E402
from microbench_embedding import CRITEO_1TB, make_batch  # noqa: E402


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--batch", type=int, default=65536)
    ap.add_argument("--scale", type=float, default=1.0)
    ap.add_argument("--alpha", type=float, default=1.1)
    ap.add_argument("--dim", type=int, default=128)
    ap.add_argument("--iters", type=int, default=20)
    ap.add_argument("--dtype", default="bf16")
    a = ap.parse_args()
    sizes = [max(1, int(v * a.scale)) for v in CRITEO_1TB]
    V, B, S, D = sum(sizes), a.batch, len(sizes), a.dim
    dt = torch.bfloat16 if a.dtype == "bf16" else torch.float32
    rng = np.random.default_rng(1234)
    emb = ha.SparseEmbeddingHash(_lib.EMB_LOCALIZED, B, 0, V, D, S, S, 0,
                                 ha.OptParams(optimizer=_lib.OPT_SGD, lr=0.01, atomic_update=False),
                                 slot_size_array=sizes, out_dtype=dt)
    emb.init_params()
    ro = torch.arange(0, B * S + 1, dtype=torch.int64, device="cuda")
    batches = [torch.from_numpy(make_batch(rng, B, sizes, a.alpha)).cuda() for _ in range(4)]
    out = torch.empty((B, S, D), dtype=dt, device="cuda")
    grad = torch.randn((B, S, D), dtype=torch.float32, device="cuda").to(dt)
    for kb in batches:
        emb.forward(True, ro, kb, out=out)
    emb.profiling(True)
    for i in range(a.iters):
        emb.forward(True, ro, batches[i % 4], out=out)
        emb.backward(grad)
        emb.update_params()
    torch.cuda.synchronize()
    print({k: (v[0] / max(v[1], 1)) * 1e3 for k, v in emb.profile().items()})


if __name__ == "__main__":
    main()
