"""embedding_collection forward / backward+update at the DLRM Criteo-1TB shape, 1 GPU
(argv[1] = static | dynamic; dynamic tables start empty and grow, the timed steps run on warm tables)."""
import os
import sys

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from hugectr_amd import _lib  # noqa: E402
from hugectr_amd.embedding_collection import (EmbeddingCollection, EmbeddingCollectionConfig,  # noqa: E402
                                              EmbeddingTableConfig)
from microbench_embedding import CRITEO_1TB, powerlaw  # noqa: E402


def timed(fn, it=10):
    for _ in range(2):
        fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(it):
        fn()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / it * 1e3


def main():
    B, D = 65536, 128
    cfg = EmbeddingCollectionConfig()
    tabs = [EmbeddingTableConfig(f"t{i}", v, D) for i, v in enumerate(CRITEO_1TB)]
    cfg.embedding_lookup(tabs, [f"b{i}" for i in range(26)], [f"e{i}" for i in range(26)],
                         ["sum"] * 26)
    storage = sys.argv[1] if len(sys.argv) > 1 else "static"
    bm = len(sys.argv) > 2 and sys.argv[2] == "bm"  # batch-major output
    opt = {"sgd": _lib.OPT_SGD, "adagrad": _lib.OPT_ADAGRAD, "ftrl": _lib.OPT_FTRL}[
        sys.argv[3] if len(sys.argv) > 3 else "sgd"]
    ebc = EmbeddingCollection.for_rank(0, 1, cfg, B, lr=0.01, optimizer=opt,
                                       out_dtype=torch.bfloat16, max_hotness=1, storage=storage,
                                       init_capacity=1 << 21, batch_major=bm)
    rng = np.random.default_rng(0)
    keys = np.concatenate([powerlaw(rng, B, v, 1.1) for v in CRITEO_1TB]).astype(np.int64)  # feature-major
    kt = torch.from_numpy(keys).cuda()
    br = torch.arange(0, 26 * B + 1, dtype=torch.int64, device="cuda")
    out = ebc.forward(kt, br)
    g = torch.randn(out.shape, device="cuda").to(out.dtype)
    print({"storage": storage, "batch_major": bm, "direct": ebc._direct, "optimizer": opt, "forward_us": round(timed(lambda: ebc.forward(kt, br)), 1),
           "backward+update_us": round(timed(lambda: ebc.backward_and_update(g)), 1),
           "out_shape": list(out.shape)})


if __name__ == "__main__":
    main()
