"""The oracle AND the HIP kernels' own source against the REFERENCE'S DEVICE CODE of the legacy
sparse embedding -- not its CPU test oracle (tests/test_ref_embedding_cpu.py does that) but the
CUDA kernels and launch wrappers themselves:

    forward_sum / forward_mean (+ the paired-half align2 forms)
                                     R/HugeCTR/src/embeddings/forward_per_gpu_functor.cu:22-243
    do_forward_scale                 R/HugeCTR/src/embeddings/forward_scale_functor.cu:23-101
    backward_sum / backward_mean     R/HugeCTR/src/embeddings/backward_functor.cu:23-158
    EmbeddingOptimizer::update       R/HugeCTR/src/optimizers/sparse_optimizer.cu:170-864 (expansion,
                                     sort by row, run counting, every optimizer kernel)

oracle/Makefile `ref` cuts these blocks out of the checkout, compiles them as plain C++ and the host
interpreter of tests/emu runs them (CUDA threads = fibers) -> oracle/_ref/libref_gpu_kernels.so.
The HIP side is hugectr_amd/csrc compiled for the same interpreter (tests/emu/emu.py), i.e. the
product's kernel source, through the C ABI.  Forward / backward are compared bit for bit (fp32 and
fp16, u32 and i64 keys, even and odd vector sizes: the align2 rounding rules); tables and optimizer
state after several updates within 1e-6 relative (per-row sums are ordered the same way, the
HIP update adds tile partials where the reference adds one gradient after the other)."""
import ctypes
import os
import sys

import numpy as np
import pytest

from util import assert_close, make_csr

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "emu"))
import emu  # noqa: E402

LIB = os.path.join(os.path.dirname(__file__), "..", "oracle", "_ref", "libref_gpu_kernels.so")
pytestmark = pytest.mark.skipif(not os.path.exists(LIB),
                                reason="oracle/_ref not built (needs the reference checkout)")

INVALID = np.uint64(0xFFFFFFFFFFFFFFFF)


def _p(a):
    return a.ctypes.data_as(ctypes.c_void_p) if a is not None else None


class RefGpu:
    def __init__(self):
        L = self.L = ctypes.CDLL(LIB)
        P, Z, I, F = ctypes.c_void_p, ctypes.c_size_t, ctypes.c_int, ctypes.c_float
        L.refgpu_forward.argtypes = [I, I, I, Z, Z, Z, P, P, P, P]
        L.refgpu_forward_scale.argtypes = [I, I, Z, Z, Z, P, P]
        L.refgpu_backward.argtypes = [I, I, I, Z, Z, Z, P, P, P]
        L.refgpu_update.argtypes = [I, I, I, I, I, F, F, F, F, F, F, ctypes.c_ulonglong, Z, Z, Z, Z,
                                    Z, P, P, P, P, P, P, P]
        L.refgpu_update.restype = I

    @staticmethod
    def _ro(ro, kb):
        return np.ascontiguousarray(ro, np.int64 if kb == 8 else np.uint32)

    def forward(self, kb, fp16, combiner, B, S, D, ro, vi, table):
        out = np.full((B * S, D), np.nan, np.float16 if fp16 else np.float32)
        r, v = self._ro(ro, kb), np.ascontiguousarray(vi, np.uint64)
        self.L.refgpu_forward(kb, fp16, combiner, B, S, D, _p(r), _p(v), _p(table), _p(out))
        return out

    def forward_scale(self, kb, fp16, B, S, D, ro, feature):
        f = np.ascontiguousarray(feature, np.float16 if fp16 else np.float32).copy()
        r = self._ro(ro, kb)
        self.L.refgpu_forward_scale(kb, fp16, B, S, D, _p(r), _p(f))
        return f

    def backward(self, kb, fp16, combiner, B, S, D, ro, top):
        t = np.ascontiguousarray(top, np.float16 if fp16 else np.float32)
        w = np.full_like(t, np.nan)
        r = self._ro(ro, kb)
        self.L.refgpu_backward(kb, fp16, combiner, B, S, D, _p(r), _p(t), _p(w))
        return w

    def update(self, kb, fp16, opt, B, S, D, V, ro, vi, wgrad, table, s0, s1, pt):
        r, v = self._ro(ro, kb), np.ascontiguousarray(vi, np.uint64).copy()
        w = np.ascontiguousarray(wgrad, np.float16 if fp16 else np.float32)
        rc = self.L.refgpu_update(kb, fp16, opt["optimizer"], opt.get("update_type", 0),
                                  opt.get("atomic", 0), opt["lr"], opt["scaler"], 0.9, 0.999, 1e-7,
                                  opt.get("mu", 0.0), opt["times"], B, S, D, V, int(ro[-1]), _p(r),
                                  _p(v), _p(w), _p(table), _p(s0), _p(s1), _p(pt))
        assert rc == 0


@pytest.fixture(scope="module")
def ref():
    return RefGpu()


def _batch(rng, B, S, hot, V, one_hot=False, invalid=0.0):
    ro, _ = make_csr(rng, B, S, hot, 10, one_hot=one_hot)
    nnz = int(ro[-1])
    vi = rng.integers(0, V, size=nnz).astype(np.uint64)
    if invalid > 0:
        vi[rng.random(nnz) < invalid] = INVALID   # a key the full table had no row for
    return ro, vi


SHAPES = [(8, 5, 4, 16), (6, 3, 7, 11), (5, 4, 3, 128), (4, 26, 1, 64)]  # B, S, max hotness, D


@pytest.mark.parametrize("kb", [8, 4])
@pytest.mark.parametrize("combiner", [0, 1])
@pytest.mark.parametrize("B,S,hot,D", SHAPES)
def test_oracle_forward_equals_the_reference_kernels(oracle, ref, B, S, hot, D, combiner, kb):
    rng = np.random.default_rng(B * 100 + S * 10 + D + combiner)
    V = 300
    table = rng.standard_normal((V, D)).astype(np.float32)
    ro, vi = _batch(rng, B, S, hot, V, invalid=0.05)
    got = ref.forward(kb, 0, combiner, B, S, D, ro, vi, table)
    want = oracle.forward(ro, vi, table, D, combiner)
    assert np.array_equal(got.view(np.uint32), want.view(np.uint32)), "fp32"
    # fp16 output: convert(sum) for sum; mean = hmul2(half(sum), half(1/n)) for even D (align2),
    # convert(sum * (1/n)) for odd D
    got16 = ref.forward(kb, 1, combiner, B, S, D, ro, vi, table)
    want16 = oracle.forward_mixed(ro, vi, table, D, combiner, "f16").astype(np.float16)
    assert np.array_equal(got16.view(np.uint16), want16.view(np.uint16)), "fp16"


@pytest.mark.parametrize("kb", [8, 4])
@pytest.mark.parametrize("combiner", [0, 1])
@pytest.mark.parametrize("B,S,hot,D", SHAPES)
def test_oracle_backward_equals_the_reference_kernels(oracle, ref, B, S, hot, D, combiner, kb):
    rng = np.random.default_rng(B * 100 + S * 10 + D + combiner + 7)
    ro, _ = _batch(rng, B, S, hot, 50)
    top = rng.standard_normal((B * S, D)).astype(np.float32)
    got = ref.backward(kb, 0, combiner, B, S, D, ro, top)
    want = oracle.backward(ro, top, D, combiner)
    assert np.array_equal(got.view(np.uint32), want.view(np.uint32)), "fp32"
    top16 = top.astype(np.float16)
    got16 = ref.backward(kb, 1, combiner, B, S, D, ro, top16)
    want16 = oracle.backward_mixed(ro, top16.astype(np.float32), D, combiner, "f16") \
        .astype(np.float16)
    assert np.array_equal(got16.view(np.uint16), want16.view(np.uint16)), "fp16"


# the reference's Optimizer_t / Update_t values (common.hpp:82-94) -> the oracle's codes
OPTS = [
    ("sgd", dict(optimizer=6), 0),
    ("sgd_atomic", dict(optimizer=6, atomic=1), 0),
    ("adam_local", dict(optimizer=1, update_type=0), 2),
    ("adam_global", dict(optimizer=1, update_type=1), 2),
    ("adam_lazy", dict(optimizer=1, update_type=2), 2),
    ("adagrad", dict(optimizer=3), 1),
    ("momentum_local", dict(optimizer=5, update_type=0, mu=0.9), 1),
    ("momentum_global", dict(optimizer=5, update_type=1, mu=0.9), 1),
    ("nesterov_local", dict(optimizer=4, update_type=0, mu=0.8), 1),
    ("nesterov_global", dict(optimizer=4, update_type=1, mu=0.8), 1),
]


def _oracle_opt(oracle, kw, times, state_half=0):
    m = {1: oracle.OPT_ADAM, 3: oracle.OPT_ADAGRAD, 5: oracle.OPT_MOMENTUM, 4: oracle.OPT_NESTEROV,
         6: oracle.OPT_SGD}
    o = oracle.OptParamsC()
    o.optimizer, o.update_type, o.lr = m[kw["optimizer"]], kw.get("update_type", 0), 0.05
    o.beta1, o.beta2, o.epsilon = 0.9, 0.999, 1e-7
    o.momentum_factor, o.scaler, o.times = kw.get("mu", 0.0), 8.0, times
    o.state_half = state_half
    return o


@pytest.mark.parametrize("name,kw,ns", OPTS, ids=[o[0] for o in OPTS])
@pytest.mark.parametrize("kb,D,combiner", [(8, 16, 0), (4, 11, 1)])
def test_oracle_update_equals_the_reference_optimizer(oracle, ref, name, kw, ns, kb, D, combiner):
    """EmbeddingOptimizer::update itself (host code and kernels) on the oracle's inputs, three
    steps: same rows, same state"""
    rng = np.random.default_rng(len(name) * 31 + D)
    B, S, hot, V = 24, 4, 5, 70
    t_ref = rng.standard_normal((V, D)).astype(np.float32)
    t_orc = t_ref.copy()
    mk = lambda: np.zeros((V, D), np.float32)  # noqa: E731
    r0, r1 = (mk() if ns >= 1 else None), (mk() if ns >= 2 else None)
    o0, o1 = (mk() if ns >= 1 else None), (mk() if ns >= 2 else None)
    lazy = name == "adam_lazy"
    rpt = np.ones((V, D), np.uint64) if lazy else None
    opt_ = np.ones((V, D), np.uint64) if lazy else None
    for it in range(3):
        ro, vi = _batch(rng, B, S, hot, V)
        top = rng.standard_normal((B * S, D)).astype(np.float32)
        wg = ref.backward(kb, 0, combiner, B, S, D, ro, top)
        ref.update(kb, 0, dict(kw, lr=0.05, scaler=8.0, times=it + 1), B, S, D, V, ro, vi, wg,
                   t_ref, r0, r1, rpt)
        oracle.update_params(ro, vi, oracle.backward(ro, top, D, combiner),
                             _oracle_opt(oracle, kw, it + 1), t_orc, o0, o1, opt_)
        # (atomic SGD adds the gradients of a row in whatever order the blocks run)
        tol = (2e-6, 2e-7) if kw.get("atomic") else (1e-6, 1e-7)
        assert_close(t_orc, t_ref, *tol, f"{name} table it{it}")
        if ns >= 1:
            assert_close(o0, r0, *tol, f"{name} state0 it{it}")
        if ns >= 2:
            assert_close(o1, r1, *tol, f"{name} state1 it{it}")


@pytest.fixture(scope="module")
def elib():
    if not emu.available():
        pytest.skip("no host clang++ / make")
    return emu.load_under_test()


HIP_OPTS = [
    ("sgd", dict(optimizer=6, atomic_update=0), dict(optimizer=6), 0),
    ("adam_local", dict(optimizer=1, update_type=0), dict(optimizer=1, update_type=0), 2),
    ("adam_global", dict(optimizer=1, update_type=1), dict(optimizer=1, update_type=1), 2),
    ("adagrad", dict(optimizer=3), dict(optimizer=3), 1),
    ("momentum_global", dict(optimizer=5, update_type=1, momentum_factor=0.9),
     dict(optimizer=5, update_type=1, mu=0.9), 1),
    ("nesterov_local", dict(optimizer=4, update_type=0, momentum_factor=0.9),
     dict(optimizer=4, update_type=0, mu=0.9), 1),
]


@pytest.mark.parametrize("name,hip_kw,ref_kw,ns", HIP_OPTS, ids=[o[0] for o in HIP_OPTS])
@pytest.mark.parametrize("D,combiner,fp16", [(16, 1, 0), (128, 0, 0), (16, 1, 1), (64, 0, 1)])
def test_hip_kernel_source_equals_the_reference_device_code(ref, elib, name, hip_kw, ref_kw, ns,
                                                            D, combiner, fp16):
    """the product's kernels (their source under the interpreter, through hctr_emb_*) and the
    reference's kernels on the same batches: the rows the HIP index stage hands out are given to
    the reference's forward / update; pooled vectors bit for bit, tables after every update"""
    from hugectr_amd import _lib
    rng = np.random.default_rng(11 + D + fp16)
    B, S, hot, vps = 32, 6, 4, 30
    V = S * vps + 8
    opt = dict(lr=0.05, scaler=8.0, beta1=0.9, beta2=0.999, epsilon=1e-7, **hip_kw)
    emb = emu.Embedding(elib, _lib.EMB_LOCALIZED, B, V, D, S * hot, S, combiner, opt,
                        out_dtype=1 if fp16 else 0)
    t_ref = emb.table().copy()
    sdt = np.float16 if fp16 else np.float32
    r0 = np.zeros((V, D), sdt) if ns >= 1 else None
    r1 = np.zeros((V, D), sdt) if ns >= 2 else None
    for it in range(3):
        ro, keys = make_csr(rng, B, S, hot, vps, one_hot=(combiner == 0 and it == 1))
        out = emb.forward(True, ro, keys)
        vi = emb.value_index(keys.size).copy()
        # (pooled from the HIP side's own table: after an update the two tables agree to rounding,
        #  bit-equal pooling needs bit-equal rows)
        want = ref.forward(8, fp16, combiner, B, S, D, ro, vi, emb.table().copy())
        bits = np.uint16 if fp16 else np.uint32
        assert np.array_equal(out.reshape(-1, D).view(bits), want.view(bits)), f"forward it{it}"
        top = rng.standard_normal((B, S, D)).astype(sdt)
        emb.backward(top)
        emb.update_params()
        wg = ref.backward(8, fp16, combiner, B, S, D, ro, top.reshape(B * S, D))
        ref.update(8, fp16, dict(ref_kw, lr=0.05, scaler=8.0, times=it + 1), B, S, D, V, ro, vi,
                   wg, t_ref, r0, r1, None)
        # fp16 state: the reference rounds m / v to binary16 when it stores them, and so does the
        # HIP path (values kept in fp32 words): a sum that lands next to a rounding boundary may
        # round the other way, one binary16 ulp of the state
        rt, at = (2e-3, 1e-5) if (fp16 and ns) else (1e-5, 1e-6)
        assert_close(emb.table(), t_ref, rt, at, f"{name} table it{it}")
        if ns >= 1:
            assert_close(emb.opt_state(0), r0.astype(np.float32), rt, at, f"{name} state0 it{it}")
        if ns >= 2:
            assert_close(emb.opt_state(1), r1.astype(np.float32), rt, max(at, 1e-7),
                         f"{name} state1 it{it}")


@pytest.mark.parametrize("kb", [8, 4])
@pytest.mark.parametrize("B,S,hot,D", SHAPES)
def test_forward_scale_of_the_reference_is_the_oracle_mean_of_a_sum(oracle, ref, B, S, hot, D, kb):
    """distributed embedding + mean on N > 1 GPUs: pooled SUMS, reduce-scatter, then this division
    (forward_scale_functor.cu) -- on one GPU that must be the mean forward (fp32: sum * (1/n))"""
    rng = np.random.default_rng(B + S + D)
    V = 200
    table = rng.standard_normal((V, D)).astype(np.float32)
    ro, vi = _batch(rng, B, S, hot, V)
    sums = ref.forward(kb, 0, 0, B, S, D, ro, vi, table)
    got = ref.forward_scale(kb, 0, B, S, D, ro, sums)
    want = oracle.forward(ro, vi, table, D, 1)
    assert np.array_equal(got.view(np.uint32), want.view(np.uint32))
    sums16 = ref.forward(kb, 1, 0, B, S, D, ro, vi, table)
    got16 = ref.forward_scale(kb, 1, B, S, D, ro, sums16)
    want16 = oracle.forward_mixed(ro, vi, table, D, 1, "f16").astype(np.float16)
    if D % 2 == 0:  # align2: hmul2(half(sum), half(1/n)) on both paths
        assert np.array_equal(got16.view(np.uint16), want16.view(np.uint16))
    else:  # odd D: forward_mean rounds sum * (1/n) once, forward_scale rounds the sum first
        assert_close(got16.astype(np.float32), want16.astype(np.float32), 2e-3, 1e-4, "fp16 odd D")


@pytest.mark.parametrize("fp16", [0, 1])
@pytest.mark.parametrize("bpg,S,D,gpus", [(5, 26, 16, 8), (4, 7, 11, 3), (3, 5, 128, 1), (6, 3, 8, 4),
                                          (2, 26, 4, 2)])
def test_reorder_maps_equal_the_reference_reorder_kernels(oracle, ref, elib, bpg, S, D, gpus, fp16):
    """the layout change around the localized embedding's all-to-all: the reference's
    forward_reorder / backward_reorder kernels (slot s lives on GPU s % N as its (s / N)-th slot;
    slot counts that do not divide by the GPU count, more GPUs than slots) against the oracle's
    maps, the HIP kernels' source, and the index form the Model uses (parallel.reorder_row_map)"""
    import torch
    from hugectr_amd import _lib
    from hugectr_amd.parallel import reorder_row_map
    L = ref.L
    L.refgpu_reorder.argtypes = [ctypes.c_int, ctypes.c_int] + [ctypes.c_size_t] * 4 + \
        [ctypes.c_void_p] * 2
    rng = np.random.default_rng(bpg * S + D + gpus)
    dt = np.float16 if fp16 else np.float32
    n = bpg * S * D
    x = rng.standard_normal(n).astype(dt)
    for backward in (0, 1):
        want = np.full(n, np.nan, dt)
        L.refgpu_reorder(fp16, backward, bpg, S, D, gpus, _p(x), _p(want))
        assert not np.isnan(want.astype(np.float32)).any()  # a bijection: every element written
        fo = oracle.backward_reorder if backward else oracle.forward_reorder
        got = np.asarray(fo(x.astype(np.float32), bpg, S, D, gpus)).reshape(-1).astype(dt)
        assert np.array_equal(got, want), ("oracle", backward)
        hip = np.full(n, np.nan, dt)
        fn = elib.hctr_backward_reorder if backward else elib.hctr_forward_reorder
        emu.check(elib, fn(bpg, S, D, gpus, _p(x), _p(hip), _lib.F16 if fp16 else _lib.F32, None))
        assert np.array_equal(hip, want), ("hip", backward)
    # index form: row of (local sample b, slot s) in the receive buffer
    rows = reorder_row_map(bpg, S, gpus).numpy().reshape(-1)
    fwd = np.full(n, np.nan, dt)
    L.refgpu_reorder(fp16, 0, bpg, S, D, gpus, _p(x), _p(fwd))
    assert np.array_equal(x.reshape(-1, D)[rows].reshape(-1), fwd)
    bwd = np.full(n, np.nan, dt)
    L.refgpu_reorder(fp16, 1, bpg, S, D, gpus, _p(x), _p(bwd))
    scat = np.empty((bpg * S, D), dt)
    scat[rows] = x.reshape(-1, D)
    assert np.array_equal(scat.reshape(-1), bwd)
    assert torch.equal(torch.sort(torch.from_numpy(rows.astype(np.int64)))[0],
                       torch.arange(bpg * S))


def _ref_filter(ref, kb, distributed, B, S, gid, gnum, ro, keys):
    L = ref.L
    L.refgpu_filter_keys.restype = ctypes.c_size_t
    L.refgpu_filter_keys.argtypes = [ctypes.c_int, ctypes.c_int] + [ctypes.c_size_t] * 4 + \
        [ctypes.c_void_p, ctypes.c_void_p, ctypes.c_size_t, ctypes.c_void_p, ctypes.c_void_p]
    kdt = np.int64 if kb == 8 else np.uint32
    r, k = np.ascontiguousarray(ro, kdt), np.ascontiguousarray(keys, kdt)
    spg = S if distributed else S // gnum + (1 if gid < S % gnum else 0)
    ro_out = np.full(B * spg + 1, 77, kdt)
    k_out = np.zeros(max(k.size, 1), kdt)
    n = L.refgpu_filter_keys(kb, distributed, B, S, gid, gnum, _p(r), _p(k), k.size, _p(ro_out),
                             _p(k_out))
    return ro_out.astype(np.int64), k_out[:n].astype(np.int64)


@pytest.mark.parametrize("kb", [8, 4])
@pytest.mark.parametrize("B,S,hot,gnum", [(6, 26, 3, 8), (5, 7, 4, 3), (4, 3, 5, 4), (7, 5, 2, 1)])
def test_oracle_key_filters_equal_the_reference_kernels(oracle, ref, B, S, hot, gnum, kb):
    """filter_keys_per_gpu of both legacy embeddings: the reference's kernels between the library
    calls its host code makes (localized: slots with slot % N == rank, more GPUs than slots
    included; distributed: keys with key % N == rank)"""
    rng = np.random.default_rng(B * S + gnum)
    ro, keys = make_csr(rng, B, S, hot, 50)
    for gid in range(gnum):
        want_ro, want_k = oracle.localized_filter(ro, keys, B, S, gid, gnum)
        got_ro, got_k = _ref_filter(ref, kb, 0, B, S, gid, gnum, ro, keys)
        assert np.array_equal(got_ro, want_ro) and np.array_equal(got_k, want_k), ("localized", gid)
        want_ro, want_k = oracle.distributed_filter(ro, keys, B, S, gid, gnum)
        got_ro, got_k = _ref_filter(ref, kb, 1, B, S, gid, gnum, ro, keys)
        assert np.array_equal(got_ro, want_ro) and np.array_equal(got_k, want_k), ("distributed", gid)


@pytest.mark.parametrize("etype,world", [("localized", 4), ("localized", 3), ("distributed", 4)])
def test_hip_rank_shard_equals_the_reference_kernel_chain(ref, elib, etype, world):
    """one rank of N through hctr_emb_forward (HIP source: filter + index + pool) against the
    reference's device code chained the same way: its filter kernels, its forward kernel on the
    rows the HIP index stage handed out, its store_slot_id kernel against the slot ids the HIP
    embedding dumps"""
    from hugectr_amd import _lib
    L = ref.L
    L.refgpu_store_slot_id.argtypes = [ctypes.c_int, ctypes.c_size_t] + [ctypes.c_int] * 4 + \
        [ctypes.c_void_p] * 3
    rng = np.random.default_rng(world + len(etype))
    B, S, hot, vps, D = 4 * world, 7, 3, 40, 8
    V = S * vps + 8
    dist = etype == "distributed"
    for rank in range(world):
        emb = emu.Embedding(elib, _lib.EMB_DISTRIBUTED if dist else _lib.EMB_LOCALIZED, B, V, D,
                            S * hot, S, 0, dict(lr=0.1, scaler=1.0, optimizer=6), rank=rank,
                            world=world)
        table = emb.table().copy()
        ro, keys = make_csr(rng, B, S, hot, vps)
        out = emb.forward(True, ro, keys)
        fro, fkeys = _ref_filter(ref, 8, 1 if dist else 0, B, S, rank, world, ro, keys)
        spr = S if dist else emb.slots_on_rank
        assert out.shape == (B, spr, D)
        if fkeys.size == 0:
            assert not out.any()
            continue
        vi = emb.value_index(fkeys.size).copy()
        want = ref.forward(8, 0, 0, B, spr, D, fro, vi, table)
        assert np.array_equal(out.reshape(-1, D).view(np.uint32), want.view(np.uint32)), rank
        if not dist:  # slot ids (localized only: the dump carries them)
            slot_ref = np.full(V, 999, np.uint64)
            fr = np.ascontiguousarray(fro, np.int64)
            L.refgpu_store_slot_id(8, B, S, spr, world, rank, _p(fr), _p(vi), _p(slot_ref))
            n = int(vi.max()) + 1
            dk = np.zeros(V, np.int64)
            ds = np.zeros(V, np.uint64)
            dv = np.zeros((V, D), np.float32)
            cnt = ctypes.c_size_t()
            emu.check(elib, elib.hctr_emb_dump(emb.h, _p(dk), _p(ds), _p(dv), ctypes.byref(cnt),
                                               None))
            assert cnt.value == n
            row_of = dict(zip(fkeys.tolist(), vi.tolist()))
            for k, s in zip(dk[:n].tolist(), ds[:n].tolist()):
                assert slot_ref[row_of[k]] == s, (rank, k)
