"""Embedding cache in HBM and the host<->HBM tiered table (BASELINE config 4): host mirror of
gpu_cache::gpu_cache (R/gpu_cache/include/nv_gpu_cache.hpp:46-124 -- Query / Replace / Update /
Dump) and of the role of gpu_cache::UvmTable (R/gpu_cache/include/uvm_table.hpp:133-174).
Thin ctypes calls into hctr_cache_* / hctr_tiered_* -- no compute here."""
from __future__ import annotations

import ctypes

import numpy as np
import torch

from . import _lib
from ._lib import check, lib, ptr, stream_ptr


class GpuCache:
    """capacity_in_set sets of 64 slots; vectors are fp32[embedding_vec_size]"""

    def __init__(self, capacity_in_set: int, embedding_vec_size: int, key_dtype=torch.int64,
                 _handle=None):
        self.capacity_in_set, self.D, self.key_dtype = capacity_in_set, embedding_vec_size, key_dtype
        self._own = _handle is None
        self._h = ctypes.c_void_p(_handle) if _handle is not None else ctypes.c_void_p()
        if self._own:
            kt = _lib.KEY_I64 if key_dtype == torch.int64 else _lib.KEY_U32
            check(lib.hctr_cache_create(capacity_in_set, embedding_vec_size, kt, ctypes.byref(self._h)))
        self._len = torch.zeros(1, dtype=torch.int64, device="cuda")

    def __del__(self):
        if getattr(self, "_own", False) and getattr(self, "_h", None):
            lib.hctr_cache_destroy(self._h)
            self._h = None

    def Query(self, keys: torch.Tensor, values: torch.Tensor = None):
        """hits are written to values[i]; returns (missing_index, missing_keys) (host sync on the
        count, for the caller's convenience -- the ABI keeps it on the device)"""
        n = keys.numel()
        mi = torch.empty(n, dtype=torch.int64, device=keys.device)
        mk = torch.empty(n, dtype=keys.dtype, device=keys.device)
        check(lib.hctr_cache_query(self._h, ptr(keys), n, ptr(values), ptr(mi), ptr(mk),
                                   ptr(self._len), stream_ptr()))
        m = int(self._len.item())
        return mi[:m], mk[:m]

    def Replace(self, keys: torch.Tensor, values: torch.Tensor):
        values = values.contiguous()  # (a named tensor: the pointer must outlive this line)
        check(lib.hctr_cache_replace(self._h, ptr(keys), keys.numel(), ptr(values), stream_ptr()))

    def Update(self, keys: torch.Tensor, values: torch.Tensor):
        values = values.contiguous()
        check(lib.hctr_cache_update(self._h, ptr(keys), keys.numel(), ptr(values), stream_ptr()))

    def Dump(self, start_set_index: int = 0, end_set_index: int = None) -> torch.Tensor:
        end = self.capacity_in_set if end_set_index is None else end_set_index
        out = torch.empty(max(end - start_set_index, 0) * 64, dtype=self.key_dtype, device="cuda")
        check(lib.hctr_cache_dump(self._h, ptr(out), ptr(self._len), start_set_index, end,
                                  stream_ptr()))
        return out[:int(self._len.item())]


class TieredTable:
    """[host_rows, vec] fp32 table in pinned host memory behind a GpuCache; key = row"""

    def __init__(self, host_rows: int, embedding_vec_size: int, cache_capacity_in_set: int):
        self.rows, self.D = host_rows, embedding_vec_size
        self._h = ctypes.c_void_p()
        check(lib.hctr_tiered_create(host_rows, embedding_vec_size, cache_capacity_in_set,
                                     ctypes.byref(self._h)))
        addr = lib.hctr_tiered_host_rows(self._h)
        buf = (ctypes.c_float * (host_rows * embedding_vec_size)).from_address(addr)
        self.host = np.ctypeslib.as_array(buf).reshape(host_rows, embedding_vec_size)
        self.cache = GpuCache(cache_capacity_in_set, embedding_vec_size,
                              _handle=lib.hctr_tiered_cache(self._h))
        self._miss = torch.zeros(1, dtype=torch.int64, device="cuda")

    def __del__(self):
        if getattr(self, "_h", None):
            self.host = None
            lib.hctr_tiered_destroy(self._h)
            self._h = None

    def lookup(self, keys: torch.Tensor) -> torch.Tensor:
        out = torch.empty((keys.numel(), self.D), dtype=torch.float32, device=keys.device)
        check(lib.hctr_tiered_lookup(self._h, ptr(keys), keys.numel(), ptr(out), ptr(self._miss),
                                     stream_ptr()))
        return out

    def last_missing(self) -> int:
        return int(self._miss.item())

    def flush(self):
        """cached rows that are newer than the host table go home (write-back cache): call before
        reading `.host`"""
        check(lib.hctr_tiered_flush(self._h, stream_ptr()))

    def scatter_add(self, unique_keys: torch.Tensor, values: torch.Tensor, alpha: float = 1.0):
        check(lib.hctr_tiered_scatter(self._h, ptr(unique_keys), unique_keys.numel(),
                                      ptr(values.contiguous()), 1, alpha, stream_ptr()))

    def scatter_update(self, unique_keys: torch.Tensor, values: torch.Tensor):
        check(lib.hctr_tiered_scatter(self._h, ptr(unique_keys), unique_keys.numel(),
                                      ptr(values.contiguous()), 0, 1.0, stream_ptr()))


class UvmTable:
    """gpu_cache::UvmTable(device_table_capacity, host_table_capacity, max_batch_size, vec_size,
    default_value) (R/gpu_cache/include/uvm_table.hpp:127-174): arbitrary keys -> fp32 vectors that
    live in host memory, the hot ones in HBM.  add(h_keys, h_vectors) / query(d_keys) -> d_vectors /
    clear() as in the reference; lookup / scatter_rows serve the training path (hugectr_amd.h)."""

    def __init__(self, device_table_capacity: int, host_table_capacity: int, max_batch_size: int,
                 vec_size: int, default_value: float = 0.0, key_dtype=torch.int64):
        self.D, self.capacity, self.max_batch = vec_size, host_table_capacity, max_batch_size
        self.key_dtype = key_dtype
        self._h = ctypes.c_void_p()
        kt = _lib.KEY_I64 if key_dtype == torch.int64 else _lib.KEY_U32
        check(lib.hctr_uvm_create(device_table_capacity, host_table_capacity, max_batch_size,
                                  vec_size, float(default_value), kt, ctypes.byref(self._h)))
        tier = lib.hctr_uvm_tier(self._h)
        addr = lib.hctr_tiered_host_rows(tier)
        buf = (ctypes.c_float * (host_table_capacity * vec_size)).from_address(addr)
        self.host = np.ctypeslib.as_array(buf).reshape(host_table_capacity, vec_size)
        self.cache = GpuCache(-(-device_table_capacity // 64), vec_size,
                              _handle=lib.hctr_tiered_cache(tier))
        self._miss = torch.zeros(1, dtype=torch.int64, device="cuda")

    def __del__(self):
        if getattr(self, "_h", None):
            self.host = None
            lib.hctr_uvm_destroy(self._h)
            self._h = None

    def add(self, h_keys: np.ndarray, h_vectors: np.ndarray):
        kd = np.int64 if self.key_dtype == torch.int64 else np.uint32
        k = np.ascontiguousarray(h_keys, dtype=kd)
        v = np.ascontiguousarray(h_vectors, dtype=np.float32).reshape(k.size, self.D)
        check(lib.hctr_uvm_add(self._h, k.ctypes.data_as(ctypes.c_void_p),
                               v.ctypes.data_as(ctypes.c_void_p), k.size))

    def query(self, d_keys: torch.Tensor) -> torch.Tensor:
        out = torch.empty((d_keys.numel(), self.D), dtype=torch.float32, device=d_keys.device)
        check(lib.hctr_uvm_query(self._h, ptr(d_keys), d_keys.numel(), ptr(out), stream_ptr()))
        return out

    def clear(self):
        check(lib.hctr_uvm_clear(self._h, stream_ptr()))

    def flush(self):
        """cached rows newer than the host store go home: call before reading `.host`"""
        check(lib.hctr_tiered_flush(lib.hctr_uvm_tier(self._h), stream_ptr()))

    def lookup(self, d_keys: torch.Tensor, want_rows: bool = True):
        """training: unseen keys take the next host row; -> (vectors, rows | None)"""
        n = d_keys.numel()
        out = torch.empty((n, self.D), dtype=torch.float32, device=d_keys.device)
        rows = torch.empty(n, dtype=torch.int64, device=d_keys.device) if want_rows else None
        check(lib.hctr_uvm_lookup(self._h, ptr(d_keys), n, ptr(out), ptr(rows), ptr(self._miss),
                                  stream_ptr()))
        return out, rows

    def scatter_rows(self, unique_rows: torch.Tensor, values: torch.Tensor, add: bool = True,
                     alpha: float = 1.0):
        values = values.contiguous()
        check(lib.hctr_uvm_scatter_rows(self._h, ptr(unique_rows), unique_rows.numel(), ptr(values),
                                        1 if add else 0, alpha, stream_ptr()))

    def check_overflow(self):
        check(lib.hctr_uvm_check_overflow(self._h, stream_ptr()))

    def size(self) -> int:
        n = ctypes.c_size_t()
        check(lib.hctr_uvm_size(self._h, stream_ptr(), ctypes.byref(n)))
        return n.value

    def last_missing(self) -> int:
        return int(self._miss.item())


class TieredEmbedding:
    """One-hot embedding over a table that does not fit HBM (BASELINE config 4): key = row of a
    flat [rows, vec] table in host memory, hot rows cached in HBM.  forward = tiered lookup;
    backward_update = per-row gradient sums (hctr_ebc_local_reduce, ascending position order) and a
    write-through SGD step on the unique rows.  `keys` of one call: int64 [n]."""

    def __init__(self, host_rows: int, embedding_vec_size: int, cache_capacity_in_set: int,
                 max_keys_per_batch: int, lr: float = 0.01):
        self.table = TieredTable(host_rows, embedding_vec_size, cache_capacity_in_set)
        self.D, self.lr, self.max_keys = embedding_vec_size, lr, max_keys_per_batch
        self._upd = ctypes.c_void_p()
        check(lib.hctr_updater_create(max_keys_per_batch, max_keys_per_batch, embedding_vec_size,
                                      ctypes.byref(self._upd)))
        dev = torch.device("cuda", torch.cuda.current_device())
        self._ro = torch.arange(max_keys_per_batch + 1, dtype=torch.int64, device=dev)
        self._urow = torch.empty(max_keys_per_batch, dtype=torch.int64, device=dev)
        self._wgrad = torch.empty((max_keys_per_batch, embedding_vec_size), dtype=torch.float32,
                                  device=dev)
        self._keys = None

    def __del__(self):
        if getattr(self, "_upd", None):
            lib.hctr_updater_destroy(self._upd)
            self._upd = None

    def forward(self, keys: torch.Tensor) -> torch.Tensor:
        assert keys.numel() <= self.max_keys and keys.dtype == torch.int64
        self._keys = keys.contiguous()
        return self.table.lookup(self._keys)

    def backward_update(self, grad: torch.Tensor):
        """grad [n, vec] of the vectors the last forward returned"""
        n = self._keys.numel()
        nu = ctypes.c_size_t()
        dt = {torch.float32: _lib.F32, torch.float16: _lib.F16, torch.bfloat16: _lib.BF16}[grad.dtype]
        grad = grad.contiguous()
        check(lib.hctr_ebc_local_reduce(self._upd, n, n, ptr(self._ro), ptr(self._keys),
                                        self.table.rows, None, ptr(grad), dt,
                                        ctypes.byref(nu), ptr(self._urow), None, ptr(self._wgrad),
                                        stream_ptr()))
        self.table.scatter_add(self._urow[:nu.value], self._wgrad[:nu.value], alpha=-self.lr)


class UvmEmbedding:
    """One-hot embedding over tables whose KEY SPACE does not fit anywhere (BASELINE configs[3]:
    4 x 10 B rows): arbitrary int64 keys -> UvmTable rows (host store, hot rows in HBM, first touch
    takes the next row).  forward = hctr_uvm_lookup; backward_update = per-row gradient sums
    (hctr_ebc_local_reduce on the rows, ascending position order) and the write-back SGD step on
    the distinct rows.  Several tables share the store through disjoint key ranges."""

    def __init__(self, device_table_capacity: int, host_table_capacity: int, embedding_vec_size: int,
                 max_keys_per_batch: int, lr: float = 0.01):
        self.table = UvmTable(device_table_capacity, host_table_capacity, max_keys_per_batch,
                              embedding_vec_size)
        self.D, self.lr, self.max_keys = embedding_vec_size, lr, max_keys_per_batch
        self._upd = ctypes.c_void_p()
        check(lib.hctr_updater_create(max_keys_per_batch, max_keys_per_batch, embedding_vec_size,
                                      ctypes.byref(self._upd)))
        dev = torch.device("cuda", torch.cuda.current_device())
        self._ro = torch.arange(max_keys_per_batch + 1, dtype=torch.int64, device=dev)
        self._urow = torch.empty(max_keys_per_batch, dtype=torch.int64, device=dev)
        self._wgrad = torch.empty((max_keys_per_batch, embedding_vec_size), dtype=torch.float32,
                                  device=dev)
        self._rows = None

    def __del__(self):
        if getattr(self, "_upd", None):
            lib.hctr_updater_destroy(self._upd)
            self._upd = None

    def forward(self, keys: torch.Tensor) -> torch.Tensor:
        assert keys.numel() <= self.max_keys and keys.dtype == torch.int64
        out, self._rows = self.table.lookup(keys.contiguous())
        return out

    def backward_update(self, grad: torch.Tensor):
        """grad [n, vec] of the vectors the last forward returned"""
        n = self._rows.numel()
        nu = ctypes.c_size_t()
        dt = {torch.float32: _lib.F32, torch.float16: _lib.F16, torch.bfloat16: _lib.BF16}[grad.dtype]
        grad = grad.contiguous()
        check(lib.hctr_ebc_local_reduce(self._upd, n, n, ptr(self._ro), ptr(self._rows),
                                        self.table.capacity, None, ptr(grad), dt,
                                        ctypes.byref(nu), ptr(self._urow), None, ptr(self._wgrad),
                                        stream_ptr()))
        self.table.scatter_rows(self._urow[:nu.value], self._wgrad[:nu.value], add=True,
                                alpha=-self.lr)
