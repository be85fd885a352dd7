"""numpy/ctypes front-end of the CPU oracle (oracle/hctr_oracle.c).

TEST INFRASTRUCTURE ONLY: imported by tests/, __graft_entry__.smoke() and bench.py's cpu_baseline
leg -- never by the product package hugectr_amd/.
"""
import ctypes
import os
import subprocess
from ctypes import POINTER, Structure, c_float, c_int, c_int64, c_uint32, c_uint64, c_void_p

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
_SO = os.path.join(_HERE, "libhctr_oracle.so")

INVALID = np.uint64(0xFFFFFFFFFFFFFFFF)
OPT_ADAM, OPT_ADAGRAD, OPT_MOMENTUM, OPT_NESTEROV, OPT_SGD = range(5)
UPDATE_LOCAL, UPDATE_GLOBAL, UPDATE_LAZY = range(3)


def build(force=False):
    if force or not os.path.exists(_SO) or os.path.getmtime(_SO) < os.path.getmtime(
            os.path.join(_HERE, "hctr_oracle.c")):
        subprocess.check_call(["make", "-C", _HERE, "-s"])
    return _SO


class OptParamsC(Structure):
    _fields_ = [("optimizer", c_int), ("update_type", c_int), ("lr", c_float), ("beta1", c_float),
                ("beta2", c_float), ("epsilon", c_float), ("momentum_factor", c_float),
                ("scaler", c_float), ("times", c_uint64), ("state_half", c_int)]


def _p(a):
    return a.ctypes.data_as(c_void_p) if a is not None else None


_lib = None


def lib():
    global _lib
    if _lib is None:
        build()
        L = ctypes.CDLL(_SO)
        L.hco_murmur3_32.restype = c_uint32
        L.hco_murmur3_32.argtypes = [c_void_p, c_int, c_uint32]
        L.hco_hash_key.restype = c_uint32
        L.hco_hash_key.argtypes = [c_int64, c_int]
        L.hco_ht_create.restype = c_void_p
        L.hco_ht_create.argtypes = [c_uint64, c_int]
        L.hco_ht_destroy.argtypes = [c_void_p]
        L.hco_ht_clear.argtypes = [c_void_p]
        for f in ("hco_ht_table_size", "hco_ht_size", "hco_ht_value_head"):
            getattr(L, f).restype = c_uint64
            getattr(L, f).argtypes = [c_void_p]
        L.hco_ht_set_value_head.argtypes = [c_void_p, c_uint64]
        L.hco_ht_insert.restype = c_int
        L.hco_ht_insert.argtypes = [c_void_p, c_void_p, c_void_p, c_uint64]
        L.hco_ht_get_insert.restype = c_int
        L.hco_ht_get_insert.argtypes = [c_void_p, c_void_p, c_void_p, c_uint64]
        L.hco_ht_get_mark.argtypes = [c_void_p, c_void_p, c_void_p, c_uint64]
        L.hco_ht_dump.restype = c_uint64
        L.hco_ht_dump.argtypes = [c_void_p, c_void_p, c_void_p]
        L.hco_localized_filter.restype = c_uint64
        L.hco_localized_filter.argtypes = [c_void_p, c_void_p, c_int64, c_int64, c_int64, c_int64,
                                           c_void_p, c_void_p]
        L.hco_distributed_filter.restype = c_uint64
        L.hco_distributed_filter.argtypes = L.hco_localized_filter.argtypes
        L.hco_slots_on_gpu.restype = c_int64
        L.hco_slots_on_gpu.argtypes = [c_int64, c_int64, c_int64]
        L.hco_forward.argtypes = [c_int64, c_int64, c_int, c_void_p, c_void_p, c_void_p, c_void_p,
                                  c_int]
        L.hco_backward.argtypes = [c_int64, c_int64, c_int, c_void_p, c_void_p, c_void_p]
        L.hco_forward_reorder.argtypes = [c_int64, c_int64, c_int64, c_int64, c_void_p, c_void_p]
        L.hco_backward_reorder.argtypes = L.hco_forward_reorder.argtypes
        L.hco_update_params.restype = c_int64
        L.hco_update_params.argtypes = [c_int64, c_int64, c_int64, c_void_p, c_void_p, c_void_p,
                                        POINTER(OptParamsC), c_void_p, c_void_p, c_void_p,
                                        c_void_p, c_int, c_int]
        L.hco_interaction_fwd.argtypes = [c_int64, c_int64, c_int64, c_void_p, c_void_p, c_void_p]
        L.hco_interaction_bwd.argtypes = [c_int64, c_int64, c_int64, c_void_p, c_void_p, c_void_p,
                                          c_void_p, c_void_p]
        L.hco_cross_v1_fwd.argtypes = [c_int64, c_int64, c_int, c_void_p, c_void_p, c_void_p,
                                       c_void_p, c_void_p]
        L.hco_cross_v1_bwd.argtypes = [c_int64, c_int64, c_int] + [c_void_p] * 8
        L.hco_cross_v2_fwd.argtypes = [c_int64, c_int64, c_int64, c_int] + [c_void_p] * 7
        L.hco_cross_v2_bwd.argtypes = [c_int64, c_int64, c_int64, c_int] + [c_void_p] * 11
        L.hco_ebc_forward.argtypes = [c_int64, c_int64] + [c_void_p] * 8 + [c_int64, c_int,
                                                                          c_void_p]
        L.hco_ebc_backward_update.argtypes = [c_int64, c_int64, c_void_p, c_int64, c_void_p,
                                              c_void_p, c_void_p, c_void_p, c_int64, c_int64,
                                              c_int, c_void_p, c_int, c_float, c_float, c_float,
                                              c_void_p, c_void_p, c_float, c_float, c_float,
                                              c_void_p]
        L.hco_keys_to_indices.argtypes = [c_int64, c_void_p, c_int64, c_int64, c_void_p]
        L.hco_powerlaw_keys.argtypes = [c_uint32, c_int64, c_int64, c_float, c_void_p]
        _lib = L
    return _lib


# ---- thin numpy API ---------------------------------------------------------------------------
def murmur3_32(data: bytes, seed: int = 0) -> int:
    buf = ctypes.create_string_buffer(data, len(data))
    return int(lib().hco_murmur3_32(ctypes.cast(buf, c_void_p), len(data), seed))


def hash_keys(keys, key_bytes):
    keys = np.asarray(keys, dtype=np.int64)
    return np.array([lib().hco_hash_key(int(k), key_bytes) for k in keys], dtype=np.uint32)


class HashTable:
    def __init__(self, capacity, key_bytes=8):
        self.h = lib().hco_ht_create(capacity, key_bytes)

    def __del__(self):
        if getattr(self, "h", None):
            lib().hco_ht_destroy(self.h)
            self.h = None

    def table_size(self):
        return int(lib().hco_ht_table_size(self.h))

    def size(self):
        return int(lib().hco_ht_size(self.h))

    def value_head(self):
        return int(lib().hco_ht_value_head(self.h))

    def set_value_head(self, v):
        lib().hco_ht_set_value_head(self.h, v)

    def insert(self, keys, vals):
        keys = np.ascontiguousarray(keys, dtype=np.int64)
        vals = np.ascontiguousarray(vals, dtype=np.uint64)
        return lib().hco_ht_insert(self.h, _p(keys), _p(vals), keys.size)

    def get_insert(self, keys):
        keys = np.ascontiguousarray(keys, dtype=np.int64)
        out = np.empty(keys.size, dtype=np.uint64)
        rc = lib().hco_ht_get_insert(self.h, _p(keys), _p(out), keys.size)
        assert rc == 0, "oracle hash table full"
        return out

    def get_mark(self, keys):
        keys = np.ascontiguousarray(keys, dtype=np.int64)
        out = np.empty(keys.size, dtype=np.uint64)
        lib().hco_ht_get_mark(self.h, _p(keys), _p(out), keys.size)
        return out

    def dump(self):
        n = self.table_size()
        k = np.empty(n, dtype=np.int64)
        v = np.empty(n, dtype=np.uint64)
        c = int(lib().hco_ht_dump(self.h, _p(k), _p(v)))
        return k[:c], v[:c]


def slots_on_gpu(slot_num, gid, gnum):
    return int(lib().hco_slots_on_gpu(slot_num, gid, gnum))


def localized_filter(row_offset, keys, batch, slot_num, gid, gnum):
    ro = np.ascontiguousarray(row_offset, dtype=np.int64)
    k = np.ascontiguousarray(keys, dtype=np.int64)
    spg = slots_on_gpu(slot_num, gid, gnum)
    oro = np.zeros(batch * spg + 1, dtype=np.int64)
    ok = np.empty(max(k.size, 1), dtype=np.int64)
    n = int(lib().hco_localized_filter(_p(ro), _p(k), batch, slot_num, gid, gnum, _p(oro), _p(ok)))
    return oro, ok[:n]


def distributed_filter(row_offset, keys, batch, slot_num, gid, gnum):
    ro = np.ascontiguousarray(row_offset, dtype=np.int64)
    k = np.ascontiguousarray(keys, dtype=np.int64)
    oro = np.zeros(batch * slot_num + 1, dtype=np.int64)
    ok = np.empty(max(k.size, 1), dtype=np.int64)
    n = int(lib().hco_distributed_filter(_p(ro), _p(k), batch, slot_num, gid, gnum, _p(oro),
                                         _p(ok)))
    return oro, ok[:n]


def forward(row_offset, value_index, table, D, combiner, threads=1):
    ro = np.ascontiguousarray(row_offset, dtype=np.int64)
    vi = np.ascontiguousarray(value_index, dtype=np.uint64)
    t = np.ascontiguousarray(table, dtype=np.float32)
    buckets = ro.size - 1
    out = np.empty((buckets, D), dtype=np.float32)
    lib().hco_forward(buckets, D, combiner, _p(ro), _p(vi), _p(t), _p(out), threads)
    return out


def backward(row_offset, top_grad, D, combiner):
    ro = np.ascontiguousarray(row_offset, dtype=np.int64)
    g = np.ascontiguousarray(top_grad, dtype=np.float32)
    w = np.empty_like(g)
    lib().hco_backward(ro.size - 1, D, combiner, _p(ro), _p(g), _p(w))
    return w


def forward_reorder(x, bpg, slot_num, D, gnum):
    x = np.ascontiguousarray(x, dtype=np.float32)
    out = np.empty(bpg * slot_num * D, dtype=np.float32)
    lib().hco_forward_reorder(bpg, slot_num, D, gnum, _p(x), _p(out))
    return out.reshape(bpg, slot_num, D)


def backward_reorder(x, bpg, slot_num, D, gnum):
    x = np.ascontiguousarray(x, dtype=np.float32)
    out = np.empty(bpg * slot_num * D, dtype=np.float32)
    lib().hco_backward_reorder(bpg, slot_num, D, gnum, _p(x), _p(out))
    return out


def update_params(row_offset, value_index, wgrad, opt: OptParamsC, table, state0=None, state1=None,
                  prev_time=None, fast_sort=True, threads=1):
    """in-place on table/state arrays (must be C-contiguous float32 / uint64)."""
    ro = np.ascontiguousarray(row_offset, dtype=np.int64)
    vi = np.ascontiguousarray(value_index, dtype=np.uint64)
    g = np.ascontiguousarray(wgrad, dtype=np.float32)
    vocab, D = table.shape
    assert table.flags.c_contiguous and table.dtype == np.float32
    return int(lib().hco_update_params(ro.size - 1, D, vocab, _p(ro), _p(vi), _p(g),
                                       ctypes.byref(opt), _p(table), _p(state0), _p(state1),
                                       _p(prev_time), 1 if fast_sort else 0, threads))


def interaction_fwd(mlp, emb):
    mlp = np.ascontiguousarray(mlp, dtype=np.float32)
    emb = np.ascontiguousarray(emb, dtype=np.float32)
    B, W = mlp.shape
    n_emb = emb.shape[1]
    n_ins = n_emb + 1
    out = np.empty((B, W + n_ins * (n_ins - 1) // 2 + 1), dtype=np.float32)
    lib().hco_interaction_fwd(B, n_emb, W, _p(mlp), _p(emb), _p(out))
    return out


def interaction_bwd(mlp, emb, top_grad):
    mlp = np.ascontiguousarray(mlp, dtype=np.float32)
    emb = np.ascontiguousarray(emb, dtype=np.float32)
    g = np.ascontiguousarray(top_grad, dtype=np.float32)
    B, W = mlp.shape
    mg = np.empty_like(mlp)
    eg = np.empty_like(emb)
    lib().hco_interaction_bwd(B, emb.shape[1], W, _p(mlp), _p(emb), _p(g), _p(mg), _p(eg))
    return mg, eg


def cross_v1_fwd(x0, kernels, biases):
    x0 = np.ascontiguousarray(x0, dtype=np.float32)
    k = np.ascontiguousarray(kernels, dtype=np.float32)
    b = np.ascontiguousarray(biases, dtype=np.float32)
    B, w = x0.shape
    L = k.shape[0]
    outputs = np.empty((L, B, w), dtype=np.float32)
    hiddens = np.empty((L, B), dtype=np.float32)
    lib().hco_cross_v1_fwd(B, w, L, _p(x0), _p(k), _p(b), _p(outputs), _p(hiddens))
    return outputs, hiddens


def cross_v1_bwd(x0, kernels, outputs, hiddens, out_grad):
    x0 = np.ascontiguousarray(x0, dtype=np.float32)
    k = np.ascontiguousarray(kernels, dtype=np.float32)
    g = np.ascontiguousarray(out_grad, dtype=np.float32)
    B, w = x0.shape
    L = k.shape[0]
    ig = np.empty_like(x0)
    kg = np.empty_like(k)
    bg = np.empty_like(k)
    lib().hco_cross_v1_bwd(B, w, L, _p(x0), _p(k), _p(outputs), _p(hiddens), _p(g), _p(ig), _p(kg),
                           _p(bg))
    return ig, kg, bg


def cross_v2_fwd(x0, U, V, biases):
    x0 = np.ascontiguousarray(x0, dtype=np.float32)
    U = np.ascontiguousarray(U, dtype=np.float32)
    V = np.ascontiguousarray(V, dtype=np.float32)
    b = np.ascontiguousarray(biases, dtype=np.float32)
    B, w = x0.shape
    L, _, p = U.shape
    outputs = np.empty((L, B, w), dtype=np.float32)
    hiddens = np.empty((L, B, w), dtype=np.float32)
    XUs = np.empty((L, B, p), dtype=np.float32)
    lib().hco_cross_v2_fwd(B, w, p, L, _p(x0), _p(U), _p(V), _p(b), _p(outputs), _p(hiddens),
                           _p(XUs))
    return outputs, hiddens, XUs


def cross_v2_bwd(x0, U, V, outputs, hiddens, XUs, out_grad):
    x0 = np.ascontiguousarray(x0, dtype=np.float32)
    U = np.ascontiguousarray(U, dtype=np.float32)
    V = np.ascontiguousarray(V, dtype=np.float32)
    g = np.ascontiguousarray(out_grad, dtype=np.float32)
    B, w = x0.shape
    L, _, p = U.shape
    ig = np.empty_like(x0)
    dU = np.empty_like(U)
    dV = np.empty_like(V)
    db = np.empty((L, w), dtype=np.float32)
    lib().hco_cross_v2_bwd(B, w, p, L, _p(x0), _p(U), _p(V), _p(outputs), _p(hiddens), _p(XUs),
                           _p(g), _p(ig), _p(dU), _p(dV), _p(db))
    return ig, dU, dV, db


def powerlaw_keys(seed, n, vocab, alpha):
    out = np.empty(n, dtype=np.int64)
    lib().hco_powerlaw_keys(seed, n, vocab, alpha, _p(out))
    return out


def ebc_forward(batch, table_ids, ev, combiners, keys, bucket_range, table_row_start, tables,
                num_gpus=1, batch_major=False):
    """tables: one flat [total_rows, ev] array (all tables share ev); returns [num_gpus, ...]"""
    L = len(table_ids)
    t = np.ascontiguousarray(table_ids, dtype=np.int32)
    evs = np.full(L, ev, dtype=np.int32)
    c = np.ascontiguousarray(combiners, dtype=np.int32)
    k = np.ascontiguousarray(keys, dtype=np.int64)
    br = np.ascontiguousarray(bucket_range, dtype=np.int64)
    rs = np.ascontiguousarray(table_row_start, dtype=np.int64)
    tes = rs * ev  # element offset of each table in the flat array
    tab = np.ascontiguousarray(tables, dtype=np.float32)
    out = np.zeros((num_gpus, L * ev * (batch // num_gpus)), dtype=np.float32)
    lib().hco_ebc_forward(batch, L, _p(t), _p(evs), _p(c), _p(k), _p(br), _p(rs), _p(tes), _p(tab),
                          num_gpus, 1 if batch_major else 0, _p(out))
    return out


def ebc_backward_update(batch, table_ids, ev, combiners, keys, bucket_range, table_row_start,
                        tables, top_grad, optimizer=0, lr=0.1, scaler=1.0, epsilon=1e-7, accum=None,
                        num_gpus=1, batch_major=False, ftrl=(0.0, 0.0, 0.0), ftrl_z=None):
    """in place on `tables` (flat [total_rows, ev] float32, C-contiguous) and `accum`."""
    t = np.ascontiguousarray(table_ids, dtype=np.int32)
    c = np.ascontiguousarray(combiners, dtype=np.int32)
    k = np.ascontiguousarray(keys, dtype=np.int64)
    br = np.ascontiguousarray(bucket_range, dtype=np.int64)
    rs = np.ascontiguousarray(table_row_start, dtype=np.int64)
    g = np.ascontiguousarray(top_grad, dtype=np.float32)
    assert tables.flags.c_contiguous and tables.dtype == np.float32
    lib().hco_ebc_backward_update(batch, len(table_ids), _p(t), ev, _p(c), _p(k), _p(br), _p(rs),
                                  tables.shape[0], num_gpus, 1 if batch_major else 0, _p(g),
                                  optimizer, lr, scaler, epsilon, _p(tables), _p(accum),
                                  float(ftrl[0]), float(ftrl[1]), float(ftrl[2]), _p(ftrl_z))


# ---- mixed precision: the 16-bit embedding output / gradient modes -------------------
def round_to(x, dtype: str):
    """fp32 array rounded (nearest even) to "f16" / "bf16", returned as fp32; "f32" is the identity"""
    x = np.asarray(x, dtype=np.float32)
    if dtype == "f32":
        return x.copy()
    if dtype == "f16":
        return x.astype(np.float16).astype(np.float32)
    u = x.view(np.uint32).astype(np.uint64)
    r = ((u + 0x7FFF + ((u >> 16) & 1)) >> 16) << 16  # round-to-nearest-even on the top 16 bits
    return r.astype(np.uint32).view(np.float32)


def forward_mixed(row_offset, value_index, table, D, combiner, dtype: str):
    """forward with a 16-bit output.  sum: convert(fp32 sum) (forward_sum_kernel).  mean, even D:
    forward_mean_align2_kernel (forward_per_gpu_functor.cu:136-176) = hmul2(half(sum), half(1/n));
    mean, odd D: forward_mean_kernel :98-131 = convert(sum * (1/n)) in fp32."""
    ro = np.asarray(row_offset, dtype=np.int64)
    s = forward(ro, value_index, table, D, 0)  # fp32 sums in key order
    if combiner == 0 or dtype == "f32":
        if combiner == 1:
            return forward(ro, value_index, table, D, 1)
        return round_to(s, dtype)
    n = (ro[1:] - ro[:-1]).astype(np.float32)
    sc = np.where(n > 1, np.float32(1.0) / np.maximum(n, 1), np.float32(1.0)).astype(np.float32)
    if D % 2 == 0:
        return round_to(round_to(s, dtype) * round_to(sc, dtype)[:, None], dtype)
    return round_to(s * sc[:, None], dtype)


def backward_mixed(row_offset, top_grad, D, combiner, dtype: str):
    """wgrad in the gradient's own 16-bit type: backward_mean_align2_kernel (backward_functor.cu:83-104)
    multiplies by half(1/n) in half precision for even D, backward_mean_kernel :57-80 in fp32 else."""
    ro = np.asarray(row_offset, dtype=np.int64)
    g = round_to(top_grad, dtype).reshape(-1, D)
    if combiner == 0:
        return g
    n = (ro[1:] - ro[:-1]).astype(np.float32)
    sc = np.where(n > 1, np.float32(1.0) / np.maximum(n, 1), np.float32(1.0)).astype(np.float32)
    if dtype != "f32" and D % 2 == 0:
        sc = round_to(sc, dtype)
    return round_to(g * sc[:, None], dtype)
