"""time of the (row, bucket) sort alone at the bench's size (1.7 M pairs, 29 key bits, power-law rows)"""
import sys, json
sys.path.insert(0, ".")
import numpy as np, torch
from hugectr_amd._lib import check, lib, ptr, stream_ptr
n = 1_703_936
rng = np.random.default_rng(0)
res = {}
for name, keys in (("power_law_rows", np.minimum(rng.zipf(1.1, n) - 1, (1 << 28) - 1).astype(np.uint32)),
                   ("uniform_rows", rng.integers(0, 187_767_399, n).astype(np.uint32))):
    k = torch.from_numpy(keys.view(np.int32)).cuda(); v = torch.arange(n, dtype=torch.int32, device="cuda")
    ko, vo = torch.empty_like(k), torch.empty_like(v)
    tb = lib.hctr_radix_sort_temp_bytes(n); tmp = torch.empty(tb, dtype=torch.uint8, device="cuda")
    for end_bit in (29, 20):
        for _ in range(5):
            check(lib.hctr_radix_sort_pairs_u32(ptr(tmp), tb, ptr(k), ptr(ko), ptr(v), ptr(vo), n, end_bit, stream_ptr()))
        a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        a.record()
        for _ in range(50):
            check(lib.hctr_radix_sort_pairs_u32(ptr(tmp), tb, ptr(k), ptr(ko), ptr(v), ptr(vo), n, end_bit, stream_ptr()))
        b.record(); torch.cuda.synchronize()
        res[f"{name}_end_bit_{end_bit}_us"] = a.elapsed_time(b) / 50 * 1e3
print(json.dumps(res))
