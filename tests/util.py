"""Shared synthetic-input builders for the parity tests (seeded, numpy)."""
import numpy as np


def make_csr(rng, batch, slot_num, max_hot, vocab_per_slot, empty_frac=0.2, one_hot=False,
             key_offset=True):
    """Full-batch CSR as the reference's reader hands it to every GPU: row_offset[batch*slot+1],
    keys with cumulative slot offsets added (R/HugeCTR/src/pybind/add_input.cpp:315-317)."""
    buckets = batch * slot_num
    if one_hot:
        lens = np.ones(buckets, dtype=np.int64)
    else:
        lens = rng.integers(0, max_hot + 1, size=buckets).astype(np.int64)
        lens[rng.random(buckets) < empty_frac] = 0
    ro = np.zeros(buckets + 1, dtype=np.int64)
    np.cumsum(lens, out=ro[1:])
    nnz = int(ro[-1])
    slot_of_bucket = np.tile(np.arange(slot_num), batch)
    slot_of_key = np.repeat(slot_of_bucket, lens)
    keys = rng.integers(0, vocab_per_slot, size=nnz).astype(np.int64)
    if key_offset:
        keys = keys + slot_of_key * vocab_per_slot
    return ro, keys


def assert_close(a, b, rtol, atol, what=""):
    a = np.asarray(a, dtype=np.float64)
    b = np.asarray(b, dtype=np.float64)
    err = np.abs(a - b)
    tol = atol + rtol * np.abs(b)
    bad = err > tol
    assert not bad.any(), (f"{what}: {bad.sum()} / {bad.size} mismatches, max abs err "
                           f"{err.max():.3e}, max rel {np.max(err / (np.abs(b) + 1e-30)):.3e}")
