"""wall time of the first training steps of a dynamic-table collection (the steps in which the
classes grow: rows AND optimizer state are re-laid out), MLPerf DCNv2 tables and hotness:
   HCTR_DYNAMIC_FLAT=0|1 python tools/dyn_growth.py adam|adagrad|sgd [log2 initial capacity]"""
import json
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch  # noqa: E402

import bench  # noqa: E402
from hugectr_amd import _lib  # noqa: E402
from hugectr_amd.embedding_collection import (EmbeddingCollection, EmbeddingCollectionConfig,  # noqa: E402
                                              EmbeddingTableConfig)

opt = sys.argv[1] if len(sys.argv) > 1 else "adam"
cap = 1 << (int(sys.argv[2]) if len(sys.argv) > 2 else 20)
B, D = 65536, 128
dev = torch.device("cuda", 0)
torch.cuda.set_device(0)
sizes, hot = bench.MLPERF_TABLES, bench.MLPERF_HOTNESS
cfg = EmbeddingCollectionConfig()
tabs = [EmbeddingTableConfig(f"t{i}", v, D) for i, v in enumerate(sizes)]
cfg.embedding_lookup(tabs, [f"b{i}" for i in range(26)], "sparse_embedding", ["sum"] * 26)
code = {"sgd": _lib.OPT_SGD, "adagrad": _lib.OPT_ADAGRAD, "adam": _lib.OPT_ADAM}[opt]
ebc = EmbeddingCollection(cfg, B, lr=0.01, optimizer=code, scaler=1024.0, out_dtype=torch.float16,
                          batch_major=True, max_hotness=max(hot), hotness=hot, storage="dynamic",
                          init_capacity=cap)
g = torch.Generator(device=dev)
g.manual_seed(99)
steps = []
grad = None
for it in range(8):
    ks = []
    for v, h in zip(sizes, hot):
        u = torch.rand(B * h, device=dev, generator=g, dtype=torch.float32).double()
        a = 1.0 - 1.1
        y = ((float(v) ** a - 1.0) * u + 1.0) ** (1.0 / a)
        ks.append((torch.round(y) - 1).clamp_(0, v - 1).to(torch.int64))
    br = torch.zeros(26 * B + 1, dtype=torch.int64, device=dev)
    torch.cumsum(torch.tensor(hot, device=dev).repeat_interleave(B), 0, out=br[1:])
    keys = torch.cat(ks)
    torch.cuda.synchronize()
    caps0 = sum(ebc.det.capacity_per_class())
    t0 = time.perf_counter()
    out = ebc.forward(keys, br)
    torch.cuda.synchronize()
    t1 = time.perf_counter()
    if grad is None:
        grad = (torch.randn(out.shape, device=dev) * 1e-3).to(out.dtype)
    ebc.backward_and_update(grad)
    torch.cuda.synchronize()
    steps.append({"ms": round((time.perf_counter() - t0) * 1e3, 2),
                  "forward_ms": round((t1 - t0) * 1e3, 2), "rows_before": caps0,
                  "rows_after": sum(ebc.det.capacity_per_class()),
                  "repairs": ebc.det.repair_count()})
print(json.dumps({"optimizer": opt, "flat": os.environ.get("HCTR_DYNAMIC_FLAT", "1"),
                  "initial_capacity_per_class": cap, "steps": steps}))
