"""time of the (row, bucket) sort alone: 1.7 M pairs (the bench's batch) and 0.5 M (its cold pairs),
key widths on both sides of the 10- / 11-bit digit choice"""
import sys, json
sys.path.insert(0, ".")
import numpy as np, torch
from hugectr_amd._lib import check, lib, ptr, stream_ptr
rng = np.random.default_rng(0)
res = {}
for n in (1_703_936, 500_000):
    for name, keys in (("power_law_rows", np.minimum(rng.zipf(1.1, n) - 1, (1 << 20) - 1).astype(np.uint32)),
                       ("uniform_rows", rng.integers(0, 1 << 20, n).astype(np.uint32))):
        k = torch.from_numpy(keys.view(np.int32)).cuda(); v = torch.arange(n, dtype=torch.int32, device="cuda")
        ko, vo = torch.empty_like(k), torch.empty_like(v)
        tb = lib.hctr_radix_sort_temp_bytes(n); tmp = torch.empty(tb, dtype=torch.uint8, device="cuda")
        for end_bit in (20, 22, 23, 29):
            for _ in range(5):
                check(lib.hctr_radix_sort_pairs_u32(ptr(tmp), tb, ptr(k), ptr(ko), ptr(v), ptr(vo), n, end_bit, stream_ptr()))
            a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            a.record()
            for _ in range(50):
                check(lib.hctr_radix_sort_pairs_u32(ptr(tmp), tb, ptr(k), ptr(ko), ptr(v), ptr(vo), n, end_bit, stream_ptr()))
            b.record(); torch.cuda.synchronize()
            res[f"n{n}_{name}_end_bit_{end_bit}_us"] = round(a.elapsed_time(b) / 50 * 1e3, 1)
print(json.dumps(res))
