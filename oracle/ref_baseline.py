"""TEST / MEASUREMENT INFRASTRUCTURE ONLY -- the reference's own CPU path as a timed baseline.

Drives oracle/_ref/libref_embedding.so = the reference's `SparseEmbeddingHashCpu`
(R/test/utest/embedding/sparse_embedding_hash_cpu.hpp:52-1015, compiled from the reference
checkout by `make -C oracle ref`; every function body in that library is the reference's) the way
the reference's tests drive it (localized_slot_sparse_embedding_hash_test.cu:181-519): a Norm
dataset file + a sparse model directory on disk; per iteration `forward()` = read_a_batch (:343-377,
the reference's own DataReader-side parse of the Norm records) + hash lookup + pooling, `backward()`,
`update_params()` (:920-1015).  Single-threaded, as the reference code is.

Only bench.py's `cpu_baseline` leg and tests/ import this module; nothing in hugectr_amd/ does.

Caveat that decides the iteration count: `cpu_csr_sort` (:541-561) is an odd-even transposition
sort, O(nnz^2) compare-swaps per batch -- 7e8 at B = 1024 x 26 keys -- so one C1 iteration takes on
the order of a second whatever the optimizer.
"""
import ctypes
import os
import shutil
import tempfile
import time

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
LIB = os.path.join(HERE, "_ref", "libref_embedding.so")

REF_OPT = {"adam": 1, "adagrad": 3, "nesterov": 4, "momentum": 5, "sgd": 6}  # common.hpp:82-92
# R/README.md:72-74 -- the DCN quick-start's slot_size_array (BASELINE configs[0])
C1_SLOTS = [39884, 39043, 17289, 7420, 20263, 3, 7120, 1543, 39884, 39043, 17289, 7420, 20263, 3,
            7120, 1543, 63, 63, 39884, 39043, 17289, 7420, 20263, 3, 7120, 1543]


def available() -> bool:
    return os.path.exists(LIB)


def _lib():
    L = ctypes.CDLL(LIB)
    L.ref_emb_create.restype = ctypes.c_void_p
    L.ref_emb_create.argtypes = ([ctypes.c_int] * 9 + [ctypes.c_longlong] + [ctypes.c_int] * 3 +
                                 [ctypes.c_float] * 6 + [ctypes.c_char_p] * 2)
    L.ref_emb_step.argtypes = [ctypes.c_void_p, ctypes.c_int, ctypes.c_int, ctypes.c_void_p,
                               ctypes.c_void_p]
    L.ref_emb_destroy.argtypes = [ctypes.c_void_p, ctypes.c_int]
    L.ref_emb_stage_seconds.argtypes = [ctypes.c_void_p, ctypes.c_int, ctypes.c_void_p]
    return L


def powerlaw(rng, n, vocab, alpha):
    """IntPowerLawDataSimulator (R/HugeCTR/include/data_generator.hpp:108-129), vectorised"""
    if alpha <= 0:
        return rng.integers(0, vocab, size=n).astype(np.int64)
    u = rng.random(n, dtype=np.float32).astype(np.float64)
    a = 1.0 - alpha
    y = ((float(vocab) ** a - 1.0) * u + 1.0) ** (1.0 / a)
    return np.clip(np.round(y) - 1, 0, vocab - 1).astype(np.int64)


def write_norm_one_hot(path, label, dense, keys):
    """Norm dataset file, no CheckSum, one key per slot (DataSetHeader 8 x i64, then per sample
    label f32[L], dense f32[Dn], per slot {nnz i32, key i64}; R/HugeCTR/include/common.hpp:184-191
    and sparse_embedding_hash_cpu.hpp:343-377) -- vectorised form of hugectr_amd.data.write_norm,
    which the CPU tests pin against the reference reader."""
    n, S = keys.shape
    L, Dn = label.shape[1], dense.shape[1]
    rec = np.dtype([("label", "<f4", (L,)), ("dense", "<f4", (Dn,)),
                    ("slots", [("nnz", "<i4"), ("key", "<i8")], (S,))])
    assert rec.itemsize == 4 * (L + Dn) + 12 * S  # packed
    a = np.zeros(n, dtype=rec)
    a["label"], a["dense"] = label, dense
    a["slots"]["nnz"] = 1
    a["slots"]["key"] = keys
    with open(path, "wb") as f:
        f.write(np.array([0, n, L, Dn, S, 0, 0, 0], dtype="<i8").tobytes())
        a.tofile(f)


def time_reference_cpu(slot_sizes, batch, dim, optimizer, update_type, alpha, warmup, iters,
                       budget_s, seed=4321, lr=0.001):
    """-> dict(samples_per_s, iters, seconds, s_per_iter, ...) for the reference CPU embedding path
    (reader + hash + forward + backward + update) on one-hot power-law keys over `slot_sizes`.
    Stops early once `budget_s` of timed work is spent (at least 2 timed iterations)."""
    S = len(slot_sizes)
    V = int(sum(slot_sizes))
    offs = np.concatenate([[0], np.cumsum(slot_sizes)[:-1]]).astype(np.int64)
    rng = np.random.default_rng(seed)
    n = batch * (warmup + iters)
    keys = np.stack([powerlaw(rng, n, v, alpha) + o for v, o in zip(slot_sizes, offs)], axis=1)
    d = tempfile.mkdtemp(prefix="hctr_refcpu_")
    try:
        write_norm_one_hot(os.path.join(d, "data.bin"), rng.random((n, 1), dtype=np.float32),
                           rng.random((n, 13), dtype=np.float32), keys)
        with open(os.path.join(d, "list.txt"), "w") as f:
            f.write(f"1\n{d}/data.bin\n")
        os.makedirs(os.path.join(d, "model"))
        np.arange(V, dtype="<i8").tofile(os.path.join(d, "model", "key"))
        ((rng.random((V, dim), dtype=np.float32) - 0.5) * 0.1).astype("<f4").tofile(
            os.path.join(d, "model", "emb_vector"))
        L = _lib()
        h = L.ref_emb_create(0, batch, S, V, dim, S, 1, 13, 0, n, 0, REF_OPT[optimizer],
                             update_type, lr, 1.0, 0.9, 0.999, 1e-7, 0.0,
                             os.path.join(d, "list.txt").encode(),
                             os.path.join(d, "model").encode())
        if not h:
            raise RuntimeError("ref_emb_create failed")
        try:
            done_w = 0
            t_w = time.perf_counter()
            for _ in range(warmup):
                if L.ref_emb_step(h, 0, 1, None, None) != 0:
                    raise RuntimeError("ref_emb_step failed")
                done_w += 1
                if time.perf_counter() - t_w > budget_s / 4:  # the warm-up shares the bound
                    break
            st0 = (ctypes.c_double * 3)()
            L.ref_emb_stage_seconds(h, 0, st0)
            t0 = time.perf_counter()
            done = 0
            while done < iters:
                if L.ref_emb_step(h, 0, 1, None, None) != 0:
                    raise RuntimeError("ref_emb_step failed")
                done += 1
                if done >= 2 and time.perf_counter() - t0 > budget_s:
                    break
            el = time.perf_counter() - t0
            st1 = (ctypes.c_double * 3)()
            L.ref_emb_stage_seconds(h, 0, st1)
            stages = {k: (st1[i] - st0[i]) / done * 1e3 for i, k in enumerate(
                ("forward_read_hash_pool", "backward", "update_sort_optimizer"))}
        finally:
            L.ref_emb_destroy(h, 0)
    finally:
        shutil.rmtree(d, ignore_errors=True)
    return {"samples_per_s": batch * done / el, "iters": done, "warmup_iters": done_w,
            "seconds": el, "s_per_iter": el / done, "batch": batch, "rows": V, "dim": dim,
            "nnz_per_batch": batch * S, "stage_ms_per_iter": stages}
