"""SparseOperationKit-shaped lookup surface on PyTorch.

Mirrors `sparse_operation_kit` (R/sparse_operation_kit/sparse_operation_kit/): `init`,
`Variable` (Distributed / Localized, distributed_variable.py:26-331), `DynamicVariable`
(dynamic_variable.py:34-300), `lookup_sparse` (lookup.py:425-700), `OptimizerWrapper`
(optimizer.py:25-250), `export` / `assign` (dynamic_variable.py:465-520).  TensorFlow's
RaggedTensor / IndexedSlices have no PyTorch equivalent, so ids are `Ragged(values, row_lengths)`
(2-D sparse COO tensors are accepted too) and the sparse gradient of a variable is kept on the
variable between `backward()` and `OptimizerWrapper.step()`.

Compute is the C ABI's: hash / index (`hctr_det_lookup_index`), gather + pooling
(`hctr_forward_pool*`), per-key gradients (`hctr_expand_key_grads`), sparse update
(`hctr_updater_update` for static variables, `hctr_det_update` for dynamic ones).  Torch is used
for buffers, for the key-ownership masks of the multi-GPU route and for `torch.distributed`.

Multi-GPU (one process per GPU): the reference's schedule -- all-gather keys, every GPU pools the
rows it owns for the global batch, partial sums return to the sample's GPU (reduce-scatter), mean
is divided on the receiver.  Distributed variables own row r on GPU r % N at local row r // N
(distributed_variable.py:231-233); dynamic variables own key k on GPU k % N; a localized variable
lives on its target GPU.
"""
from __future__ import annotations

import ctypes
import os
from typing import List, Optional, Sequence, Union

import numpy as np
import torch
import torch.distributed as dist

from . import _lib
from ._lib import check, lib, ptr, stream_ptr
from .dynamic_table import DynamicEmbeddingTable, DynamicTableOptimizer, _num_state

_RANK, _WORLD = 0, 1
INVALID = -1  # 0xFFFFFFFFFFFFFFFF as int64: "row not on this GPU / unknown key"


def init(group=None):
    """sok.init(): picks rank / size up from torch.distributed (horovod in the reference)"""
    global _RANK, _WORLD
    if dist.is_available() and dist.is_initialized():
        _RANK, _WORLD = dist.get_rank(group), dist.get_world_size(group)
    else:
        _RANK, _WORLD = 0, 1


def rank() -> int:
    return _RANK


def num_gpus() -> int:
    return _WORLD


class Ragged:
    """values[nnz] (int64 keys or float weights) + row_lengths[batch] -- tf.RaggedTensor's two
    components as lookup.py:446-460 reads them"""

    def __init__(self, values: torch.Tensor, row_lengths: torch.Tensor):
        self.values = values.contiguous()
        self.row_lengths = row_lengths.to(torch.int64).contiguous()
        assert int(self.row_lengths.sum()) == self.values.numel()

    @staticmethod
    def from_sparse(sp: torch.Tensor) -> "Ragged":
        sp = sp.coalesce()
        rows = sp.indices()[0]
        lens = torch.bincount(rows, minlength=sp.shape[0])
        return Ragged(sp.values(), lens)

    @property
    def batch(self) -> int:
        return self.row_lengths.numel()


def _as_ragged(x) -> Ragged:
    if isinstance(x, Ragged):
        return x
    if isinstance(x, torch.Tensor) and x.is_sparse:
        return Ragged.from_sparse(x)
    if isinstance(x, (tuple, list)) and len(x) == 2:
        return Ragged(x[0], x[1])
    raise TypeError("sp_ids / sp_weights must be sok.Ragged, a 2-D sparse COO tensor or "
                    "(values, row_lengths)")


def _offsets(lens: torch.Tensor) -> torch.Tensor:
    ro = torch.zeros(lens.numel() + 1, dtype=torch.int64, device=lens.device)
    torch.cumsum(lens, 0, out=ro[1:])
    return ro


# ---- variables ----------------------------------------------------------------------------------
class _VariableBase:
    """what lookup_sparse needs of a variable: dimension, target_gpu, key -> local row"""
    dimension: int
    target_gpu: int  # -1: distributed over all GPUs

    _count = 0

    def __init__(self, name: Optional[str] = None):
        _VariableBase._count += 1
        self.name = name or f"sok_variable_{_VariableBase._count}"
        # autograd handle: lookups take it as an input so that backward reaches the variable
        self._token = torch.zeros(1, device="cuda", requires_grad=True)
        self._pending: List[tuple] = []  # (row_offset, rows/keys, bucket grads | None, key grads | None)

    def _owned(self, keys: torch.Tensor) -> torch.Tensor:
        if self.target_gpu >= 0:
            full = self.target_gpu == _RANK
            return torch.full_like(keys, full, dtype=torch.bool)
        if _WORLD == 1:
            return torch.ones_like(keys, dtype=torch.bool)
        return (keys % _WORLD) == _RANK


class DistributedVariable(_VariableBase):
    """rows sharded round-robin: global row r -> GPU r % N, local row r // N"""

    def __init__(self, initial_value: torch.Tensor, target_gpu: int = -1,
                 name: Optional[str] = None):
        super().__init__(name)
        v = torch.as_tensor(initial_value, dtype=torch.float32)
        assert v.dim() == 2
        self.global_shape = tuple(v.shape)
        self.dimension = v.shape[1]
        self.target_gpu = target_gpu
        if target_gpu >= 0:
            local = v if target_gpu == _RANK else v[:0]
        elif _WORLD > 1:
            local = v[_RANK::_WORLD]
        else:
            local = v
        self.weight = local.contiguous().cuda()
        self._updater = None
        self._states: List[torch.Tensor] = []

    def key_map(self, keys: torch.Tensor) -> torch.Tensor:
        if self.target_gpu >= 0 or _WORLD == 1:
            return keys
        return torch.div(keys, _WORLD, rounding_mode="floor")

    def _rows(self, keys: torch.Tensor, train: bool) -> torch.Tensor:
        return self.key_map(keys)

    def _table(self):
        return self.weight

    def numpy(self):
        return self.weight.detach().cpu().numpy()


class LocalizedVariable(DistributedVariable):
    pass


def Variable(initial_value, mode: Optional[str] = None, name: Optional[str] = None, **_kw):
    """sok.Variable: mode None / "distributed" -> DistributedVariable, "localized:<gpu>" ->
    LocalizedVariable on that GPU (distributed_variable.py:26-125)"""
    if mode is None or mode == "distributed":
        return DistributedVariable(initial_value, name=name)
    if mode.startswith("localized"):
        gpu = int(mode.split(":")[1]) if ":" in mode else 0
        return LocalizedVariable(initial_value, target_gpu=gpu, name=name)
    raise ValueError(f"unknown mode {mode!r}")


class DynamicVariable(_VariableBase):
    """key -> vector map that grows on demand; keys live on GPU key % N"""

    def __init__(self, dimension: int, initializer: Union[str, float, None] = None,
                 key_type=torch.int64, init_capacity: int = 1 << 20, mode: Optional[str] = None,
                 seed: int = 0, name: Optional[str] = None):
        super().__init__(name)
        self.dimension = int(dimension)
        self.key_type = key_type
        self.target_gpu = -1
        if mode is not None and mode.startswith("localized"):
            self.target_gpu = int(mode.split(":")[1]) if ":" in mode else 0
        self.initializer_str = "" if initializer is None else str(initializer)
        self._det = DynamicEmbeddingTable([self.dimension], self.initializer_str, init_capacity,
                                          key_type, seed=seed + 1000003 * _RANK)
        self._opt: Optional[DynamicTableOptimizer] = None
        self._updater = None  # (hctr_updater handle, capacity) of the gradient reduce

    @property
    def size(self) -> int:
        return self._det.size()

    def _rows(self, keys: torch.Tensor, train: bool) -> torch.Tensor:
        idx = torch.empty(keys.numel(), dtype=torch.int64, device=keys.device)
        check(lib.hctr_det_lookup_index(self._det._h, 0, ptr(keys), keys.numel(), 1 if train else 0,
                                        ptr(idx), stream_ptr()))
        return idx

    def _table(self) -> torch.Tensor:
        p, cap = ctypes.c_void_p(), ctypes.c_size_t()
        check(lib.hctr_det_rows(self._det._h, 0, ctypes.byref(p), ctypes.byref(cap)))
        return _view_f32(p.value, (cap.value, self.dimension))

    # dynamic_variable.py:294-340
    def sparse_read(self, indices: torch.Tensor) -> torch.Tensor:
        return self._det.lookup(indices.contiguous()).view(-1, self.dimension)

    def scatter_add(self, indices, values):
        self._det.scatter_add(indices.contiguous(), values)

    def scatter_sub(self, indices, values):
        self._det.scatter_add(indices.contiguous(), -values)

    def scatter_update(self, indices, values):
        self._det.scatter_update(indices.contiguous(), values)


def _view_f32(addr: int, shape) -> torch.Tensor:
    """torch view of library-owned device memory (no copy)"""
    n = 1
    for s in shape:
        n *= s

    class _Holder:
        pass

    h = _Holder()
    h.__cuda_array_interface__ = {"shape": (n,), "typestr": "<f4", "data": (addr, False),
                                  "version": 2}
    return torch.as_tensor(h, device="cuda").view(*shape)


def export(var: DynamicVariable):
    """(indices, values) of a DynamicVariable (dynamic_variable.py:465-492)"""
    return var._det.export(0)


def assign(var: DynamicVariable, indices: torch.Tensor, values: torch.Tensor):
    """insert-or-overwrite (dynamic_variable.py:494-520)"""
    var._det.lookup(indices.contiguous())       # inserts what is missing
    var._det.scatter_update(indices.contiguous(), values)


# ---- collectives (NCCL = RCCL on the GPU boxes; gloo stages through the host, used by tests) ------
def _backend_is_gloo() -> bool:
    return dist.get_backend() == "gloo"


def _all_gather_cat(t: torch.Tensor) -> torch.Tensor:
    """concatenate every rank's 1-D tensor (sizes may differ)"""
    n = torch.tensor([t.numel()], dtype=torch.int64, device=t.device)
    sizes = [torch.zeros_like(n) for _ in range(_WORLD)]
    if _backend_is_gloo():
        sizes = [s.cpu() for s in sizes]
        dist.all_gather(sizes, n.cpu())
    else:
        dist.all_gather(sizes, n)
    sizes = [int(s) for s in sizes]
    m = max(sizes)
    pad = torch.zeros(m, dtype=t.dtype, device=t.device)
    pad[:t.numel()] = t
    if _backend_is_gloo():
        outs = [torch.zeros(m, dtype=t.dtype) for _ in range(_WORLD)]
        dist.all_gather(outs, pad.cpu())
        outs = [o.to(t.device) for o in outs]
    else:
        outs = [torch.zeros_like(pad) for _ in range(_WORLD)]
        dist.all_gather(outs, pad)
    return torch.cat([o[:s] for o, s in zip(outs, sizes)])


def _reduce_scatter_rows(x: torch.Tensor, local_rows: int) -> torch.Tensor:
    """x [N * local_rows, D] partial sums -> my [local_rows, D] slice of the total"""
    if _backend_is_gloo():
        c = x.cpu()
        dist.all_reduce(c)
        return c[_RANK * local_rows:(_RANK + 1) * local_rows].to(x.device)
    out = torch.empty((local_rows, x.shape[1]), dtype=x.dtype, device=x.device)
    dist.reduce_scatter_tensor(out, x.contiguous())
    return out


def _all_gather_rows(x: torch.Tensor) -> torch.Tensor:
    if _backend_is_gloo():
        outs = [torch.zeros(x.shape, dtype=x.dtype) for _ in range(_WORLD)]
        dist.all_gather(outs, x.cpu())
        return torch.cat(outs).to(x.device)
    out = torch.empty((x.shape[0] * _WORLD, x.shape[1]), dtype=x.dtype, device=x.device)
    dist.all_gather_into_tensor(out, x.contiguous())
    return out


# ---- lookup -------------------------------------------------------------------------------------
def _pool(table: torch.Tensor, ro: torch.Tensor, rows: torch.Tensor, weights, combiner: int, D):
    buckets = ro.numel() - 1
    if rows.numel() == 0:  # nothing of this batch lives here: all pooled vectors are zero
        return torch.zeros((buckets, D), dtype=torch.float32, device=ro.device)
    out = torch.empty((buckets, D), dtype=torch.float32, device=ro.device)
    if weights is None:
        multi = rows.numel() > buckets + buckets // 2
        fn = lib.hctr_forward_pool_multihot if multi else lib.hctr_forward_pool
        check(fn(buckets, D, combiner, ptr(ro), _lib.KEY_I64, ptr(rows), ptr(table), ptr(out),
                 _lib.F32, stream_ptr()))
    else:
        check(lib.hctr_forward_pool_weighted(buckets, D, combiner, ptr(ro), ptr(rows), ptr(weights),
                                             ptr(table), ptr(out), stream_ptr()))
    return out


class _LookupFn(torch.autograd.Function):
    """one (variable, ids[, weights]) lookup; the token input routes backward to the variable"""

    @staticmethod
    def forward(ctx, token, var, ids: Ragged, w: Optional[Ragged], combiner: int, train: bool):
        D = var.dimension
        lens, keys = ids.row_lengths, ids.values
        weights = w.values.float().contiguous() if w is not None else None
        b_local = lens.numel()
        if _WORLD > 1:
            # all-gather keys / lengths / weights: the global batch in rank order (lookup.py:484-496)
            keys = _all_gather_cat(keys)
            lens = _all_gather_cat(lens)
            if weights is not None:
                weights = _all_gather_cat(weights)
            own = var._owned(keys)
            # this GPU pools only the rows it owns: filtered CSR over the global batch
            seg = torch.repeat_interleave(torch.arange(lens.numel(), device=lens.device), lens)
            lens_own = torch.bincount(seg[own], minlength=lens.numel())
            keys_own = keys[own].contiguous()
            w_own = weights[own].contiguous() if weights is not None else None
        else:
            lens_own, keys_own, w_own = lens, keys, weights
        ro = _offsets(lens_own)
        rows = var._rows(keys_own, train)
        table = var._table()
        # the receiver divides for mean (after all shards are added), so shards always sum
        part = _pool(table, ro, rows, w_own, combiner if _WORLD == 1 else 0, D)
        if _WORLD > 1:
            out = _reduce_scatter_rows(part, b_local)
            if combiner == 1:
                if w is not None:
                    seg_l = torch.repeat_interleave(
                        torch.arange(b_local, device=out.device), ids.row_lengths)
                    den = torch.zeros(b_local, device=out.device).index_add_(
                        0, seg_l, w.values.float())
                else:
                    den = ids.row_lengths.float()
                out = out / den.clamp_min(1e-30).unsqueeze(1) * (den > 0).unsqueeze(1)
        else:
            out = part
        ctx.var, ctx.combiner, ctx.train = var, combiner, train
        ctx.ro, ctx.rows, ctx.keys, ctx.w = ro, rows, keys_own, w_own
        ctx.local = (ids.row_lengths, w.values.float() if w is not None else None, b_local)
        return out

    @staticmethod
    def backward(ctx, g):
        var, combiner = ctx.var, ctx.combiner
        g = g.contiguous().float()
        if _WORLD > 1:
            lens_l, w_l, b_local = ctx.local
            if combiner == 1:  # mean was divided on the receiver: its gradient scales here
                if w_l is not None:
                    seg_l = torch.repeat_interleave(torch.arange(b_local, device=g.device), lens_l)
                    den = torch.zeros(b_local, device=g.device).index_add_(0, seg_l, w_l)
                else:
                    den = lens_l.float()
                g = g / den.clamp_min(1e-30).unsqueeze(1) * (den > 0).unsqueeze(1)
            g = _all_gather_rows(g)  # every owner sees the whole global batch's gradients
            comb_local = 0
        else:
            comb_local = combiner
        var._pending.append((ctx.ro, ctx.rows, ctx.keys, ctx.w, g, comb_local))
        return torch.zeros(1, device=g.device), None, None, None, None, None


def lookup_sparse(params, sp_ids, sp_weights=None, combiners=None, training: bool = True):
    """sok.lookup_sparse(params, sp_ids, sp_weights=None, combiners=None): fused lookup of several
    variables; returns one [batch, dimension] tensor per variable (a list iff sp_ids is one)."""
    is_list = isinstance(sp_ids, (list, tuple)) and not (
        len(sp_ids) == 2 and isinstance(sp_ids[0], torch.Tensor) and not sp_ids[0].is_sparse
        and sp_ids[0].dim() == 1 and isinstance(params, _VariableBase))
    params = list(params) if isinstance(params, (list, tuple)) else [params]
    sp_ids = list(sp_ids) if is_list else [sp_ids]
    if combiners is None:
        combiners = ["mean"] * len(params)  # lookup.py:620-623
    combiners = list(combiners) if isinstance(combiners, (list, tuple)) else [combiners]
    if sp_weights is None:
        sp_weights = [None] * len(params)
    elif not isinstance(sp_weights, (list, tuple)) or not is_list:
        sp_weights = [sp_weights]
    if not (len(params) == len(sp_ids) == len(combiners) == len(sp_weights)):
        raise RuntimeError("params, sp_ids, sp_weights and combiners must have the same length")
    for p in params[1:]:
        if type(p) is not type(params[0]) and not (
                isinstance(p, DistributedVariable) and isinstance(params[0], DistributedVariable)):
            raise RuntimeError("Distributed/Localized/Dynamic Variable cannot be used in the same "
                               "lookup currently")  # lookup.py:436-440
    outs = []
    for var, ids, w, c in zip(params, sp_ids, sp_weights, combiners):
        if c not in ("sum", "mean"):
            raise ValueError('combiner must be "sum" or "mean"')
        ids = _as_ragged(ids)
        w = _as_ragged(w) if w is not None else None
        if w is not None and not torch.equal(w.row_lengths, ids.row_lengths):
            raise RuntimeError("sp_id and sp_weight should be have same shape.")
        outs.append(_LookupFn.apply(var._token, var, ids, w, 1 if c == "mean" else 0, training))
    return outs if is_list else outs[0]


# ---- optimizer ----------------------------------------------------------------------------------
class OptimizerWrapper:
    """sok.OptimizerWrapper: applies the sparse gradients `lookup_sparse` left on its variables.
    optimizer: "sgd" | "adagrad" | "adam" | "momentum" | "nesterov" (static and dynamic
    variables) | "rmsprop" | "ftrl" (dynamic variables only, as in the reference's tables)."""

    _CODES = {"sgd": _lib.OPT_SGD, "adagrad": _lib.OPT_ADAGRAD, "adam": _lib.OPT_ADAM,
              "momentum": _lib.OPT_MOMENTUM_SGD, "nesterov": _lib.OPT_NESTEROV,
              "rmsprop": _lib.OPT_RMSPROP, "ftrl": _lib.OPT_FTRL}

    def __init__(self, optimizer: str = "sgd", lr: float = 0.01, beta1=0.9, beta2=0.999,
                 epsilon=1e-7, momentum=0.9, rmsprop_beta=0.9, ftrl_lambda1=0.0, ftrl_lambda2=0.0,
                 ftrl_beta=0.0, scaler: float = 1.0):
        self.name = optimizer.lower()
        self.code = self._CODES[self.name]
        self.hp = dict(lr=lr, beta1=beta1, beta2=beta2, epsilon=epsilon, momentum=momentum,
                       rmsprop_beta=rmsprop_beta, ftrl_lambda1=ftrl_lambda1,
                       ftrl_lambda2=ftrl_lambda2, ftrl_beta=ftrl_beta, scaler=scaler)
        self.times = 0

    def set_learning_rate(self, lr: float):
        self.hp["lr"] = lr

    def step(self, variables: Sequence[_VariableBase]):
        """apply_gradients over the variables touched since the last step"""
        self.times += 1
        for var in variables:
            pend, var._pending = var._pending, []
            if not pend:
                continue
            if isinstance(var, DynamicVariable):
                # one optimizer step per variable per call, however many lookups used it
                ks, gs = [], []
                for ro, rows, keys, w, g, comb in pend:
                    if keys.numel() == 0:
                        continue
                    kg = torch.empty((keys.numel(), var.dimension), dtype=torch.float32,
                                     device=g.device)
                    check(lib.hctr_expand_key_grads(ro.numel() - 1, var.dimension, comb, ptr(ro),
                                                    ptr(w), ptr(g), ptr(kg), stream_ptr()))
                    ks.append(keys)
                    gs.append(kg)
                if ks:
                    self._step_dynamic(var, torch.cat(ks), torch.cat(gs))
            else:
                for ro, rows, keys, w, g, comb in pend:
                    self._step_static(var, ro, rows, w, g, comb)

    def _ensure_dynamic(self, var: "DynamicVariable"):
        """the variable's fused optimizer (and its state table) for this wrapper's optimizer"""
        if var._opt is None or var._opt.p.optimizer != self.code:
            hp = self.hp
            var._opt = DynamicTableOptimizer(
                var._det, self.code, hp["lr"], hp["beta1"], hp["beta2"], hp["epsilon"],
                hp["momentum"], hp["rmsprop_beta"], hp["ftrl_lambda1"], hp["ftrl_lambda2"],
                hp["ftrl_beta"], hp["scaler"])

    # static variable: the path's sort + segmented reduce + optimizer on the local shard
    def _step_static(self, var: DistributedVariable, ro, rows, w, g, comb):
        if self.name in ("rmsprop", "ftrl"):
            raise RuntimeError(f"{self.name} is only available for DynamicVariable")
        D = var.dimension
        nnz = rows.numel()
        if nnz == 0:
            return
        if w is not None or comb == 1:
            kg = torch.empty((nnz, D), dtype=torch.float32, device=g.device)
            check(lib.hctr_expand_key_grads(ro.numel() - 1, D, comb, ptr(ro), ptr(w), ptr(g),
                                            ptr(kg), stream_ptr()))
            g, ro = kg, torch.arange(nnz + 1, dtype=torch.int64, device=g.device)
        rows_n = max(var.weight.shape[0], 1)
        if var._updater is None or var._updater[1] < nnz:
            if var._updater is not None:
                lib.hctr_updater_destroy(var._updater[0])
            h = ctypes.c_void_p()
            cap = max(2 * nnz, 1024)
            check(lib.hctr_updater_create(cap, rows_n, D, ctypes.byref(h)))
            var._updater = (h, cap)
        ns = _num_state(self.code)
        while len(var._states) < ns:
            var._states.append(torch.zeros_like(var.weight))
        hp = self.hp
        check(lib.hctr_updater_update(
            var._updater[0], ro.numel() - 1, nnz, ptr(ro), ptr(rows), ptr(g), _lib.F32, self.code,
            _lib.UPDATE_LOCAL, hp["lr"], hp["beta1"], hp["beta2"], hp["epsilon"], hp["momentum"],
            hp["scaler"], self.times, ptr(var.weight),
            ptr(var._states[0]) if ns >= 1 else None, ptr(var._states[1]) if ns >= 2 else None,
            stream_ptr()))

    # dynamic variable: per-key gradients -> unique keys + sums (the reference's OptimizerWrapper
    # does this with tf.unique / unsorted_segment_sum, optimizer.py:170-230) -> fused HIP step
    def _step_dynamic(self, var: DynamicVariable, keys, kg):
        D = var.dimension
        n = keys.numel()
        keys = keys.contiguous()
        # unique keys + per-key gradient sums in ascending position order (deterministic): the
        # path's local reduce, sorted by the keys' row numbers in the table
        _, rows, base = var._det.lookup_rows(keys, insert=False, want_ptrs=False)
        if var._updater is None or var._updater[1] < n:
            if var._updater is not None:
                lib.hctr_updater_destroy(var._updater[0])
            h = ctypes.c_void_p()
            cap = max(2 * n, 1024)
            check(lib.hctr_updater_create(cap, cap, D, ctypes.byref(h)))
            var._updater = (h, cap)
        ro = torch.arange(n + 1, dtype=torch.int64, device=kg.device)
        urow = torch.empty(n, dtype=torch.int64, device=kg.device)
        ukey = torch.empty(n, dtype=torch.int64, device=kg.device)
        wg = torch.empty((n, D), dtype=torch.float32, device=kg.device)
        nu = ctypes.c_size_t()
        k64 = keys if keys.dtype == torch.int64 else keys.to(torch.int64)
        kg = kg.contiguous()
        check(lib.hctr_ebc_local_reduce(var._updater[0], n, n, ptr(ro), ptr(rows), base[-1],
                                        ptr(k64), ptr(kg), _lib.F32, ctypes.byref(nu),
                                        ptr(urow), ptr(ukey), ptr(wg), stream_ptr()))
        uniq = ukey[:nu.value].to(keys.dtype)
        sums = wg[:nu.value]
        self._ensure_dynamic(var)
        var._opt.set_learning_rate(self.hp["lr"])
        ev = torch.arange(0, (uniq.numel() + 1) * D, D, dtype=torch.int32, device=kg.device)
        var._opt.update(uniq.contiguous(), ev, sums.view(-1))


# ---- dump / load (R/sparse_operation_kit/sparse_operation_kit/dump_load.py; byte layout in
#      sok_format.py): per variable `<name>-key`, `<name>-weight` and, when the optimizer holds
#      state for it, `<name>-<Optimizer>-<slot>`; plus one `meta_info`.  GPU 0 writes. -------------
from . import sok_format as _fmt  # noqa: E402

# slot names follow the TF optimizers the reference wraps (optimizer.get_slot_names())
_SLOTS = {"sgd": [], "adam": ["m", "v"], "adagrad": ["accumulator"], "momentum": ["momentum"],
          "nesterov": ["momentum"], "rmsprop": ["rms"], "ftrl": ["accumulator", "linear"]}
# get_sok_optimizer_name (dump_load.py:126-143) looks for one of these in the class name
_OPT_FILE_NAME = {"sgd": "SGD", "adam": "Adam", "adagrad": "Adagrad", "ftrl": "Ftrl",
                  "momentum": "SGD", "nesterov": "SGD", "rmsprop": "RMSprop"}


def filter_variables(variables):
    """(sok variables, the others) -- sok.filter_variables (sparse_operation_kit/__init__.py)"""
    a = [v for v in variables if isinstance(v, _VariableBase)]
    return a, [v for v in variables if not isinstance(v, _VariableBase)]


def _gather_rounds(keys: torch.Tensor, mats: List[torch.Tensor], shared: bool):
    """the reference's write order (save_table_to_filesystem_*): rows go out in rounds of at most
    64 MiB; with several GPUs a round holds that slice of every rank, rank after rank"""
    n = keys.numel()
    if not shared or _WORLD == 1:
        return keys, mats
    D = mats[0].shape[1] if mats else 1
    total = torch.tensor([n], dtype=torch.int64, device=keys.device)
    sizes = _all_gather_cat(total)
    rounds, per = _fmt.rows_per_round(int(sizes.max()), D, 4)
    ks, ms = [], [[] for _ in mats]
    for r in range(rounds):
        a, b = min(r * per, n), min((r + 1) * per, n)
        ks.append(_all_gather_cat(keys[a:b]))
        for j, m in enumerate(mats):
            ms[j].append(_all_gather_cat(m[a:b].reshape(-1)).view(-1, m.shape[1]))
    return torch.cat(ks), [torch.cat(x) for x in ms]


def _var_arrays(var, optimizer):
    """(keys int64 [n], weight [n, D], [state matrices in slot order]) of this rank"""
    D = var.dimension
    slots = _SLOTS[optimizer.name] if optimizer is not None else []
    if isinstance(var, DynamicVariable):
        k, w = var._det.export(0)
        order = torch.argsort(k)
        k, w = k[order].to(torch.int64), w[order]
        states = []
        if slots and var._opt is not None and var._opt.states is not None:
            sk, sv = var._opt.states.export(0)
            sv = sv[torch.argsort(sk)]
            if sv.shape[0] == k.numel():
                states = [sv[:, j * D:(j + 1) * D].contiguous() for j in range(len(slots))]
        return k, w, states
    n = var.weight.shape[0]
    if var.target_gpu >= 0:
        k = torch.arange(n, dtype=torch.int64, device=var.weight.device)
    else:
        k = torch.arange(n, dtype=torch.int64, device=var.weight.device) * _WORLD + _RANK
    states = [t for t in var._states[:len(slots)]] if len(var._states) >= len(slots) else []
    return k, var.weight, states


def dump(path: str, dump_vars, optimizer: Optional["OptimizerWrapper"] = None):
    """sok.dump(path, sok_vars, optimizer): the reference's directory layout and byte format"""
    if not isinstance(dump_vars, (list, tuple)):
        dump_vars = [dump_vars]
    os.makedirs(path, exist_ok=True)
    infos = []
    opt_name = _OPT_FILE_NAME[optimizer.name] if optimizer is not None else ""
    slots = _SLOTS[optimizer.name] if optimizer is not None else []
    for var in dump_vars:
        k, w, states = _var_arrays(var, optimizer)
        shared = var.target_gpu < 0
        writer = 0 if shared else var.target_gpu
        if shared:
            k, mats = _gather_rounds(k, [w] + states, True)
            w, states = mats[0], mats[1:]
        if isinstance(var, DynamicVariable) and _RANK == writer and k.numel() == 0:
            raise Exception(f"dynamic table don't have value in it , table_name: {var.name}")
        if _RANK == writer:
            fname = os.path.join(path, _fmt.file_table_name(var.name))
            kn = k.cpu().numpy().astype(np.int64 if shared or isinstance(var, DynamicVariable)
                                        else np.uint64)
            _fmt.write_array_file(fname + "-key", var.name, _fmt.FILE_KEY, "", kn)
            _fmt.write_array_file(fname + "-weight", var.name, _fmt.FILE_EMB, "",
                                  w.detach().cpu().numpy().astype(np.float32))
            for slot, st in zip(slots, states):
                _fmt.write_array_file(f"{fname}-{opt_name}-{slot}", var.name, _fmt.FILE_OPT_STATE,
                                      slot, st.detach().cpu().numpy().astype(np.float32))
        infos.append(_fmt.VarInfo(var.name, opt_name, _fmt.DTYPE_INDEX[np.dtype(np.int64)],
                                  _fmt.DTYPE_INDEX[np.dtype(np.float32)], int(k.numel()),
                                  var.dimension))
    if _RANK == 0:
        _fmt.save_meta_file(path, infos)
    if _WORLD > 1:
        dist.barrier()


def load(path: str, load_vars, optimizer: Optional["OptimizerWrapper"] = None):
    """sok.load(path, sok_vars, optimizer): every rank reads the files and keeps its own keys"""
    if not isinstance(load_vars, (list, tuple)):
        load_vars = [load_vars]
    meta = _fmt.load_meta_file(path)
    opt_name = _OPT_FILE_NAME[optimizer.name] if optimizer is not None else ""
    slots = _SLOTS[optimizer.name] if optimizer is not None else []
    for var in load_vars:
        if var.name not in meta:
            raise Exception(f"table {var.name} is not in the meta_info of {path}")
        fname = os.path.join(path, _fmt.file_table_name(var.name))
        state_paths = [f"{fname}-{opt_name}-{slot}" for slot in slots]
        state_paths = [p for p in state_paths if os.path.exists(p)] if slots else []
        ok, msg, n, ev = _fmt.check_weight_files(fname + "-key", fname + "-weight", state_paths)
        if not ok:
            raise Exception(msg)
        if ev != var.dimension:
            raise Exception(f"{var.name}: file vectors have {ev} elements, the variable {var.dimension}")
        keys = _fmt.read_array_file(fname + "-key").astype(np.int64)
        w = _fmt.read_array_file(fname + "-weight").reshape(n, ev).astype(np.float32)
        states = [_fmt.read_array_file(p).reshape(n, ev).astype(np.float32) for p in state_paths]
        if var.target_gpu >= 0:
            mine = np.full(n, _RANK == var.target_gpu)
        else:
            mine = keys % _WORLD == _RANK
        keys, w, states = keys[mine], w[mine], [x[mine] for x in states]
        dev = torch.device("cuda", torch.cuda.current_device())
        kt = torch.from_numpy(keys).to(dev)
        wt = torch.from_numpy(w).to(dev)
        if isinstance(var, DynamicVariable):
            if kt.numel():
                assign(var, kt.to(var.key_type), wt)
            if states and len(states) == len(slots) and kt.numel():
                optimizer._ensure_dynamic(var)
                st = var._opt.states
                sk = kt.to(var.key_type)
                st.lookup(sk)
                st.scatter_update(sk, torch.from_numpy(np.concatenate(states, axis=1)).to(dev))
        else:
            rows = var.key_map(kt)
            if rows.numel() and int(rows.max()) >= var.weight.shape[0]:
                raise Exception(f"{var.name}: the file holds more rows than the variable")
            var.weight[rows] = wt
            if states and len(states) == len(slots):
                while len(var._states) < len(slots):
                    var._states.append(torch.zeros_like(var.weight))
                for j, x in enumerate(states):
                    var._states[j][rows] = torch.from_numpy(x).to(dev)
    if _WORLD > 1:
        dist.barrier()
