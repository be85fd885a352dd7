"""Two-pass vs fused SGD update (HCTR_SGD_FUSED=0|1): prints a digest of the table after a few
updates on power-law one-hot + multi-hot keys; run once per setting and compare the lines."""
import hashlib
import os
import sys

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from hugectr_amd import _lib  # noqa: E402
from hugectr_amd.embedding_collection import (EmbeddingCollection, EmbeddingCollectionConfig,  # noqa: E402
                                              EmbeddingTableConfig)


def main():
    B, D = 8192, 128
    sizes = [200000, 50, 3000, 7, 90000]
    hot = [3, 1, 20, 2, 1]
    cfg = EmbeddingCollectionConfig()
    tabs = [EmbeddingTableConfig(f"t{i}", v, D) for i, v in enumerate(sizes)]
    cfg.embedding_lookup(tabs, [f"b{i}" for i in range(5)], "e", ["sum"] * 5)
    for dt in (torch.float16, torch.float32):
        ebc = EmbeddingCollection(cfg, B, lr=0.05, optimizer=_lib.OPT_SGD, scaler=1024.0, out_dtype=dt,
                                  batch_major=True, max_hotness=max(hot), hotness=hot, seed=5)
        rng = np.random.default_rng(1)
        for step in range(4):
            ks = [np.minimum(rng.zipf(1.2, B * h) - 1, v - 1) for v, h in zip(sizes, hot)]
            br = np.concatenate([[0], np.cumsum(np.repeat(hot, B))]).astype(np.int64)
            out = ebc.forward(torch.from_numpy(np.concatenate(ks).astype(np.int64)).cuda(),
                              torch.from_numpy(br).cuda())
            g = torch.from_numpy(rng.standard_normal(out.shape).astype(np.float32)).cuda().to(dt)
            ebc.backward_and_update(g)
        torch.cuda.synchronize()
        print(str(dt), hashlib.sha256(ebc.table.cpu().numpy().tobytes()).hexdigest())


if __name__ == "__main__":
    main()
