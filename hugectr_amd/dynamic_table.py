"""Dynamic embedding table: host mirror of det::DynamicEmbeddingTable
(R/third_party/dynamic_embedding_table/dynamic_embedding_table.hpp:25-66) and of the optimizer step
of embedding::DynamicEmbeddingTable::update (R/HugeCTR/embedding_storage/dynamic_embedding.cu).
Thin ctypes calls into hctr_det_* -- no compute here."""
from __future__ import annotations

import ctypes
from typing import Optional, Sequence

import torch

from . import _lib
from ._lib import check, lib, ptr, stream_ptr


def _sz(seq):
    arr = (ctypes.c_size_t * len(seq))(*[int(x) for x in seq])
    return arr


def _num_state(optimizer: int) -> int:
    """OptParams::num_parameters_per_weight (R/HugeCTR/include/optimizer.hpp)"""
    return {_lib.OPT_SGD: 0, _lib.OPT_ADAM: 2, _lib.OPT_FTRL: 2}.get(optimizer, 1)


class DynamicEmbeddingTable:
    """num_classes maps key -> fp32[dim_c]; grows on demand.  Method names follow the reference:
    lookup / lookup_unsafe / scatter_add / scatter_update / remove / export / clear / size /
    capacity.  Keys of one call are grouped by class: `id_spaces[i]` owns
    keys[id_space_offsets[i]:id_space_offsets[i+1]] (host lists, as in the reference)."""

    def __init__(self, dimension_per_class: Sequence[int], initializer: str = "",
                 initial_capacity: int = 1048576, key_dtype=torch.int64, seed: int = 0):
        self.dims = [int(d) for d in dimension_per_class]
        self.key_dtype = key_dtype
        self._h = ctypes.c_void_p()
        kt = _lib.KEY_I64 if key_dtype == torch.int64 else _lib.KEY_U32
        check(lib.hctr_det_create(len(self.dims), _sz(self.dims), initializer.encode(),
                                  int(initial_capacity), kt, int(seed), ctypes.byref(self._h)))

    def __del__(self):
        if getattr(self, "_h", None):
            lib.hctr_det_destroy(self._h)
            self._h = None

    # -- helpers -------------------------------------------------------------------------------
    def _ranges(self, num_keys, id_spaces, id_space_offsets):
        if id_spaces is None:
            assert len(self.dims) == 1, "id_spaces is required for more than one class"
            id_spaces, id_space_offsets = [0], [0, num_keys]
        assert len(id_space_offsets) == len(id_spaces) + 1
        return _sz(id_spaces), _sz(id_space_offsets), len(id_spaces)

    def _elements(self, id_spaces, id_space_offsets):
        return sum((id_space_offsets[i + 1] - id_space_offsets[i]) * self.dims[id_spaces[i]]
                   for i in range(len(id_spaces)))

    # -- reference verbs -----------------------------------------------------------------------
    def lookup(self, keys: torch.Tensor, id_spaces=None, id_space_offsets=None) -> torch.Tensor:
        """inserts unseen keys (initializer), returns the vectors packed back to back"""
        n = keys.numel()
        sp, so, ns = self._ranges(n, id_spaces, id_space_offsets)
        out = torch.empty(self._elements(list(sp), list(so)), dtype=torch.float32,
                          device=keys.device)
        check(lib.hctr_det_lookup(self._h, ptr(keys), ptr(out), n, sp, so, ns, stream_ptr()))
        return out

    def lookup_unsafe(self, keys: torch.Tensor, id_spaces=None, id_space_offsets=None):
        """device pointers (int64 tensor) to the stored vectors"""
        n = keys.numel()
        sp, so, ns = self._ranges(n, id_spaces, id_space_offsets)
        out = torch.empty(n, dtype=torch.int64, device=keys.device)
        check(lib.hctr_det_lookup_unsafe(self._h, ptr(keys), ptr(out), n, sp, so, ns,
                                         stream_ptr()))
        return out

    def lookup_rows(self, keys: torch.Tensor, id_spaces=None, id_space_offsets=None,
                    insert: bool = True, want_ptrs: bool = True, want_rows: bool = True):
        """(row addresses int64[n] | None, row numbers int64[n] | None, class_row_base list):
        row number = class_row_base[class] + row inside the class, unique over the classes"""
        n = keys.numel()
        sp, so, ns = self._ranges(n, id_spaces, id_space_offsets)
        ptrs = torch.empty(n, dtype=torch.int64, device=keys.device) if want_ptrs else None
        rows = torch.empty(n, dtype=torch.int64, device=keys.device) if want_rows else None
        base = (ctypes.c_uint64 * (len(self.dims) + 1))()
        check(lib.hctr_det_lookup_rows(self._h, ptr(keys), n, sp, so, ns, 1 if insert else 0,
                                       ptr(ptrs), ptr(rows), base, stream_ptr()))
        return ptrs, rows, list(base)

    def row_store(self):
        """(address, rows) of the one flat fp32 table that the row numbers of lookup_rows index when
        all classes share a dimension, else (None, 0).  Fixed for the table's life (memory is mapped
        behind a class as it grows; rows never move)."""
        p, n = ctypes.c_void_p(), ctypes.c_uint64()
        check(lib.hctr_det_row_store(self._h, ctypes.byref(p), ctypes.byref(n)))
        return (p.value, int(n.value)) if p.value else (None, 0)

    def state_store(self, num_state: int):
        """(state0, state1 | None): addresses of the optimizer state arrays that share the row
        numbers of row_store() (zeros for a row that was never updated).  Fixed as well."""
        a, b = ctypes.c_void_p(), ctypes.c_void_p()
        check(lib.hctr_det_state_store(self._h, int(num_state), ctypes.byref(a), ctypes.byref(b),
                                       stream_ptr()))
        return a.value, b.value

    def scatter_add(self, keys, elements, id_spaces=None, id_space_offsets=None):
        sp, so, ns = self._ranges(keys.numel(), id_spaces, id_space_offsets)
        elements = elements.contiguous().float()
        check(lib.hctr_det_scatter_add(self._h, ptr(keys), ptr(elements), keys.numel(), sp, so, ns,
                                       stream_ptr()))

    def scatter_update(self, keys, elements, id_spaces=None, id_space_offsets=None):
        sp, so, ns = self._ranges(keys.numel(), id_spaces, id_space_offsets)
        elements = elements.contiguous().float()
        check(lib.hctr_det_scatter_update(self._h, ptr(keys), ptr(elements), keys.numel(), sp, so,
                                          ns, stream_ptr()))

    def remove(self, keys, id_spaces=None, id_space_offsets=None):
        sp, so, ns = self._ranges(keys.numel(), id_spaces, id_space_offsets)
        check(lib.hctr_det_remove(self._h, ptr(keys), keys.numel(), sp, so, ns, stream_ptr()))

    def export(self, class_index: int = 0):
        """(keys, values[n, dim]) of one class"""
        n = self.size_per_class()[class_index]
        keys = torch.empty(n, dtype=torch.int64 if self.key_dtype == torch.int64 else torch.int32,
                           device="cuda")
        vals = torch.empty((n, self.dims[class_index]), dtype=torch.float32, device="cuda")
        got = ctypes.c_size_t()
        check(lib.hctr_det_export(self._h, class_index, ptr(keys), ptr(vals), n,
                                  ctypes.byref(got), stream_ptr()))
        return keys[:got.value], vals[:got.value]

    def clear(self):
        check(lib.hctr_det_clear(self._h, stream_ptr()))

    def size_per_class(self):
        out = (ctypes.c_size_t * len(self.dims))()
        check(lib.hctr_det_size_per_class(self._h, out, stream_ptr()))
        return list(out)

    def capacity_per_class(self):
        out = (ctypes.c_size_t * len(self.dims))()
        check(lib.hctr_det_capacity_per_class(self._h, out))
        return list(out)

    def repair_count(self) -> int:
        """inserting lookups whose hash index gave up at its grid barrier and were repaired and
        issued again inside the call (hctr_det_repair_count)"""
        out = ctypes.c_uint64()
        check(lib.hctr_det_repair_count(self._h, ctypes.byref(out)))
        return int(out.value)

    def size(self) -> int:
        return sum(self.size_per_class())

    def capacity(self) -> int:
        return sum(self.capacity_per_class())


class DynamicTableOptimizer:
    """embedding::DynamicEmbeddingTable::update: fused optimizer step on the unique keys of a
    batch; owns the zero-initialised state table (dimension x num_parameters_per_weight)."""

    def __init__(self, table: DynamicEmbeddingTable, optimizer: int, lr: float, beta1=0.9,
                 beta2=0.999, epsilon=1e-7, momentum_factor=0.9, rmsprop_beta=0.9,
                 ftrl_lambda1=0.0, ftrl_lambda2=0.0, ftrl_beta=0.0, scaler=1.0,
                 initial_capacity: int = 1048576, with_states: bool = True):
        # with_states=False: the owner keeps the optimizer state elsewhere (the flat row store's
        # state arrays, DynamicEmbeddingTable.state_store) and never calls update() with an
        # optimizer that needs it -- no second table is allocated
        self.table = table
        self.p = _lib.DetOptParams(optimizer, lr, beta1, beta2, epsilon, momentum_factor,
                                   rmsprop_beta, ftrl_lambda1, ftrl_lambda2, ftrl_beta, scaler)
        ns = _num_state(optimizer)
        self.states: Optional[DynamicEmbeddingTable] = None
        if ns and with_states:
            self.states = DynamicEmbeddingTable([d * ns for d in table.dims], "zeros",
                                                initial_capacity, table.key_dtype)

    def set_learning_rate(self, lr: float):
        self.p.lr = lr

    def update(self, unique_keys: torch.Tensor, ev_start_indices: torch.Tensor,
               wgrad: torch.Tensor, id_spaces=None, id_space_offsets=None):
        """ev_start_indices: int32/uint32 offsets of each key's gradient in wgrad (fp32)"""
        n = unique_keys.numel()
        sp, so, ns = self.table._ranges(n, id_spaces, id_space_offsets)
        assert ev_start_indices.dtype == torch.int32 and wgrad.dtype == torch.float32
        check(lib.hctr_det_update(self.table._h, self.states._h if self.states else None,
                                  ctypes.byref(self.p), ptr(unique_keys), n, sp, so, ns,
                                  ptr(ev_start_indices), ptr(wgrad.contiguous()), stream_ptr()))
