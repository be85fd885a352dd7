"""The (row, bucket) sort of the sparse update (csrc/radix_sort.hip; replaces
cub::DeviceRadixSort::SortPairs, R/HugeCTR/src/optimizers/sparse_optimizer.cu:657-676): bit-exact
against numpy's stable argsort -- stability is what fixes the summation order of a row's gradients
(SURVEY q5)."""
import numpy as np
import pytest

pytestmark = pytest.mark.gpu


def _sort(keys, vals, end_bit):
    import torch
    from hugectr_amd._lib import check, lib, ptr, stream_ptr
    n = keys.size
    k = torch.from_numpy(keys.view(np.int32)).cuda()
    v = torch.from_numpy(vals.view(np.int32)).cuda()
    ko, vo = torch.empty_like(k), torch.empty_like(v)
    tb = lib.hctr_radix_sort_temp_bytes(n)
    tmp = torch.empty(tb, dtype=torch.uint8, device="cuda")
    check(lib.hctr_radix_sort_pairs_u32(ptr(tmp), tb, ptr(k), ptr(ko), ptr(v), ptr(vo), n, end_bit,
                                        stream_ptr()))
    torch.cuda.synchronize()
    assert (k.cpu().numpy().view(np.uint32) == keys).all(), "input keys were modified"
    return ko.cpu().numpy().view(np.uint32), vo.cpu().numpy().view(np.uint32)


@pytest.mark.parametrize("n,end_bit,dist", [
    (0, 10, "u"), (1, 5, "u"), (63, 10, "u"), (64, 11, "u"), (4095, 20, "u"), (4096, 20, "u"),
    (4097, 21, "u"), (100_003, 29, "u"), (100_003, 32, "u"), (50_000, 9, "u"), (50_000, 30, "few"),
    (1_703_936, 29, "pow"), (1_703_936, 29, "pad"), (300_000, 19, "pow"),
    # 21-22 and 31-32 bits: two / three passes of 11-bit digits (ballot matching)
    (1_703_936, 22, "pow"), (1_703_936, 22, "pad"), (1_703_936, 21, "u"), (200_000, 31, "u"),
    (70_000, 22, "few")])
def test_radix_sort_is_the_stable_sort(n, end_bit, dist):
    rng = np.random.default_rng(n + end_bit)
    hi = (1 << end_bit) - 1
    if dist == "u":
        keys = rng.integers(0, hi + 1, n, dtype=np.uint64).astype(np.uint32)
    elif dist == "few":  # a handful of distinct keys: long equal runs
        keys = rng.choice(np.array([0, 1, 7, hi, hi - 1, 12345], np.uint32), n)
    else:  # power-law rows as the update sees them (hot rows repeat thousands of times)
        keys = np.minimum(rng.zipf(1.1, n) - 1, hi).astype(np.uint32)
        if dist == "pad":  # padding keys of expand_pairs sort behind every live row
            keys[n - 5000:] = 0xFFFFFFFF
    vals = np.arange(n, dtype=np.uint32)
    ko, vo = _sort(keys, vals, end_bit)
    order = np.argsort(keys, kind="stable")
    assert (ko == keys[order]).all()
    assert (vo == vals[order]).all(), "not stable"
