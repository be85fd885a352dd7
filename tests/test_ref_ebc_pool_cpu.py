"""The HIP embedding_collection forward arithmetic against the REFERENCE'S DEVICE CODE behind every
EBC operator: R/HugeCTR/embedding/operators/generic_lookup.cuh whole (multi_to_one_* kernels,
descriptors, copy_multi_to_one's choice of kernel by vector size), cut out of the checkout and
executed by the host interpreter of tests/emu (oracle/_ref/libref_ebc_pool.so).

Composed the way the reference composes them (model_parallel_embedding.cpp forward_per_gpu):
  ModelForward     partial[shard][bucket] = round_wire( sum of the bucket's rows held by the shard )
  NetworkForward   out[bucket] = round_out( (sum over shards of partial) / keys of the bucket )   Average
                                 round_out(  sum over shards of partial )                         Sum
for fp32 and for binary16 wire / output vectors -- the 16-bit rounding points and the DIVISION by the
count (network_forward.cu:272-292; a product with the reciprocal differs in the last bit) have no
CPU counterpart in the reference's tests.  hctr_forward_pool + hctr_ebc_network_forward and the
one-GPU in-place form hctr_ebc_scale_average (HIP source under the interpreter) must give the same
BITS."""
import ctypes
import os
import sys

import numpy as np
import pytest

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "emu"))
import emu  # noqa: E402

LIB = os.path.join(os.path.dirname(__file__), "..", "oracle", "_ref", "libref_ebc_pool.so")
pytestmark = pytest.mark.skipif(not (os.path.exists(LIB) and emu.available()),
                                reason="oracle/_ref not built (needs the reference checkout)")


def _p(a):
    return a.ctypes.data_as(ctypes.c_void_p)


def _ref_multi_to_one(L, src_rows, offsets, factor, ev, src_half, dst_half):
    """src_rows: list of 1-D arrays (one vector each)"""
    n = len(offsets) - 1
    ptrs = (ctypes.c_void_p * max(len(src_rows), 1))(*[r.ctypes.data for r in src_rows])
    dst = np.full((n, ev), np.nan, np.float16 if dst_half else np.float32)
    off = np.ascontiguousarray(offsets, np.int32)
    fac = np.ascontiguousarray(factor, np.int32)
    L.refebc_multi_to_one(src_half, dst_half, n, _p(off), _p(fac), ev, ptrs, _p(dst), ev)
    assert np.isfinite(dst.astype(np.float32)).all()
    return dst


@pytest.mark.parametrize("half", [0, 1])
@pytest.mark.parametrize("ev,shards,hot", [(128, 2, 5), (16, 3, 4), (200, 2, 3), (300, 1, 6),
                                           (6, 4, 7), (64, 1, 9)])
def test_ebc_forward_arithmetic_equals_the_reference_kernels(ev, shards, hot, half):
    from hugectr_amd import _lib
    L = ctypes.CDLL(LIB)
    L.refebc_multi_to_one.argtypes = [ctypes.c_int] * 3 + [ctypes.c_void_p] * 2 + [ctypes.c_int] + \
        [ctypes.c_void_p] * 2 + [ctypes.c_int]
    lib = emu.load_under_test()
    rng = np.random.default_rng(ev * 100 + shards * 10 + hot + half)
    NL, bpg, rows = 3, 5, 40                         # lookups, samples, table rows
    combiner = np.array([1, 0, 1], np.int32)         # Average, Sum, Average
    dt = np.float16 if half else np.float32
    code = _lib.F16 if half else _lib.F32
    table = rng.standard_normal((rows, ev)).astype(np.float32)
    # buckets (l, b) with 0..hot keys; a key's shard = key % shards (row-sharded table)
    lens = rng.integers(0, hot + 1, size=NL * bpg)
    lens[rng.random(NL * bpg) < 0.15] = 0
    keys = [rng.integers(0, rows, size=n) for n in lens]
    counts = lens.astype(np.int64)
    # ---- stage 1 on both sides: per shard, the bucket's rows of that shard, summed, wire-rounded
    hip_blocks, ref_blocks = [], []
    for s in range(shards):
        mine = [k[k % shards == s] for k in keys]
        ro = np.concatenate([[0], np.cumsum([m.size for m in mine])]).astype(np.int64)
        idx = (np.concatenate(mine) if ro[-1] else np.zeros(0)).astype(np.uint64)
        out = np.full((NL * bpg, ev), np.nan, dt)
        emu.check(lib, lib.hctr_forward_pool(NL * bpg, ev, 0, _p(ro), _lib.KEY_I64, _p(idx),
                                             _p(table), _p(out), code, None))
        hip_blocks.append(out)
        src_rows = [table[int(i)] for i in idx]
        ref = _ref_multi_to_one(L, src_rows, ro, np.ones(NL * bpg), ev, 0, half)
        assert np.array_equal(out.view(np.uint16 if half else np.uint32),
                              ref.view(np.uint16 if half else np.uint32)), ("model forward", s)
        ref_blocks.append(ref)
    # ---- stage 2: blocks [(lookup, shard)][b][ev] -> out [lookup][b][ev]
    recv = np.ascontiguousarray(
        np.stack([hip_blocks[s].reshape(NL, bpg, ev)[l] for l in range(NL) for s in range(shards)]))
    src_blocks = np.arange(NL * shards, dtype=np.int32).reshape(NL, shards)
    hip_out = np.full((NL, bpg, ev), np.nan, dt)
    emu.check(lib, lib.hctr_ebc_network_forward(bpg, NL, ev, shards, _p(src_blocks), _p(combiner),
                                                _p(counts), 0, _p(recv), _p(hip_out), code, None))
    src_rows, off, fac = [], [0], []
    for l in range(NL):
        for b in range(bpg):
            for s in range(shards):
                src_rows.append(np.ascontiguousarray(ref_blocks[s].reshape(NL, bpg, ev)[l, b]))
            off.append(len(src_rows))
            fac.append(int(counts[l * bpg + b]) if combiner[l] == 1 else 1)
    ref_out = _ref_multi_to_one(L, src_rows, off, fac, ev, half, half).reshape(NL, bpg, ev)
    bits = np.uint16 if half else np.uint32
    assert np.array_equal(hip_out.view(bits), ref_out.view(bits)), "network forward"
    # ---- one GPU: the pooled sums (one shard = everything) scaled in place
    if shards == 1:
        data = hip_blocks[0].reshape(NL, bpg, ev).copy()
        emu.check(lib, lib.hctr_ebc_scale_average(bpg, NL, ev, _p(combiner), _p(counts), 0, _p(data),
                                                  code, 1, None))
        assert np.array_equal(data.view(bits), ref_out.view(bits)), "in-place average"


@pytest.mark.parametrize("half", [0, 1])
@pytest.mark.parametrize("ev,shards", [(128, 2), (16, 3), (200, 1), (300, 2), (6, 4)])
def test_ebc_backward_arithmetic_equals_the_reference_kernels(ev, shards, half):
    """NetworkBackward (copy_one_to_multi, network_backward.cu:56-100): the gradient of an output
    vector, divided by the bucket's key count for Average lookups and rounded to the wire type,
    goes to every shard of the lookup: hctr_ebc_network_backward and the in-place
    hctr_ebc_scale_average(forward = 0)"""
    from hugectr_amd import _lib
    L = ctypes.CDLL(LIB)
    L.refebc_one_to_multi.argtypes = [ctypes.c_int] * 2 + [ctypes.c_void_p] * 2 + [ctypes.c_int] + \
        [ctypes.c_void_p] * 2 + [ctypes.c_int]
    lib = emu.load_under_test()
    rng = np.random.default_rng(ev + shards + half)
    NL, bpg = 3, 6
    combiner = np.array([1, 0, 1], np.int32)
    counts = rng.integers(0, 9, size=NL * bpg).astype(np.int64)
    dt, code = (np.float16, _lib.F16) if half else (np.float32, _lib.F32)
    bits = np.uint16 if half else np.uint32
    grad = rng.standard_normal((NL, bpg, ev)).astype(dt)
    src_blocks = np.arange(NL * shards, dtype=np.int32).reshape(NL, shards)
    send = np.full((NL * shards, bpg, ev), np.nan, dt)
    emu.check(lib, lib.hctr_ebc_network_backward(bpg, NL, ev, shards, _p(src_blocks), _p(combiner),
                                                 _p(counts), 0, _p(grad), _p(send), code, None))
    want = np.full_like(send, np.nan)
    ptrs, off, fac = [], [0], []
    for l in range(NL):
        for b in range(bpg):
            for s in range(shards):
                ptrs.append(want[l * shards + s, b].ctypes.data)
            off.append(len(ptrs))
            fac.append(int(counts[l * bpg + b]) if combiner[l] == 1 else 1)
    arr = (ctypes.c_void_p * len(ptrs))(*ptrs)
    o, f = np.ascontiguousarray(off, np.int32), np.ascontiguousarray(fac, np.int32)
    L.refebc_one_to_multi(half, NL * bpg, _p(o), _p(f), ev, _p(grad), arr, ev)
    assert np.isfinite(want.astype(np.float32)).all()
    assert np.array_equal(send.view(bits), want.view(bits)), "network backward"
    if shards == 1:
        data = grad.copy()
        emu.check(lib, lib.hctr_ebc_scale_average(bpg, NL, ev, _p(combiner), _p(counts), 0, _p(data),
                                                  code, 0, None))
        assert np.array_equal(data.view(bits), want.reshape(NL, bpg, ev).view(bits)), "in place"
