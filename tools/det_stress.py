"""create / grow / destroy dynamic tables in a loop next to allocator and GEMM activity: a stress of
hctr_det_destroy (one segmentation fault inside it in sixty suite runs, round 6).
   HCTR_LIB_VARIANT=dbg python tools/det_stress.py 400"""
import gc
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from hugectr_amd.dynamic_table import DynamicEmbeddingTable  # noqa: E402

n_iter = int(sys.argv[1]) if len(sys.argv) > 1 else 200
g = torch.Generator(device="cuda").manual_seed(1)
side = torch.cuda.Stream()
keep = []
for it in range(n_iter):
    dims = [[16], [8, 8], [128], [4, 4, 4, 4]][it % 4]
    t = DynamicEmbeddingTable(dims, initializer="", initial_capacity=[8, 64, 1024][it % 3], seed=it)
    ncls = len(dims)
    for step in range(3 + it % 3):
        n = 2000 * (step + 1)
        keys = torch.randint(0, 1 << 40, (n * ncls,), device="cuda", dtype=torch.int64, generator=g)
        off = [i * n for i in range(ncls + 1)]
        v = t.lookup(keys, list(range(ncls)), off)
        if len(set(dims)) == 1:
            t.lookup_rows(keys, list(range(ncls)), off)
            if it % 2:
                t.state_store(1 + it % 2)
        t.scatter_add(keys, torch.ones_like(v), list(range(ncls)), off)
        if step == 1:
            t.remove(keys[: n // 2], [0], [0, n // 2])
        with torch.cuda.stream(side):  # allocator + library activity beside the table
            a = torch.randn(512, 512, device="cuda", dtype=torch.float16)
            b = a @ a
            keep.append(b[:1].clone())
    if it % 5 == 0:
        keep.clear()
        torch.cuda.empty_cache()
    # some tables die here, some whenever the collector gets to them
    if it % 3 == 0:
        del t
    if it % 7 == 0:
        gc.collect()
    if it % 50 == 0:
        print("iteration", it, flush=True)
torch.cuda.synchronize()
print("done", n_iter, flush=True)
