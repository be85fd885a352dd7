"""Host-side mirror of HugeCTR's legacy sparse embeddings for ONE GPU (one process per GPU).

Mirrors `class IEmbedding` (R/HugeCTR/include/embedding.hpp:26-67) and
`SparseEmbeddingHashParams` (:69-93) for `LocalizedSlotSparseEmbeddingHash` /
`DistributedSlotSparseEmbeddingHash`; every method forwards to the HIP library through the C ABI
(`include/hugectr_amd.h`).  torch is used only for device memory, streams and (in
`hugectr_amd.parallel`) torch.distributed -- never for the embedding arithmetic.
"""
from __future__ import annotations

import ctypes
import os
from dataclasses import dataclass
from typing import Optional, Sequence

import numpy as np
import torch

from . import _lib
from ._lib import check, lib, ptr, stream_ptr

_TORCH_TO_EMB = {torch.float32: _lib.F32, torch.float16: _lib.F16, torch.bfloat16: _lib.BF16}


@dataclass
class OptParams:
    """OptParams (R/HugeCTR/include/optimizer.hpp:149-155) + the per-optimizer hyper parameters,
    defaults as in `CreateOptimizer` (R/HugeCTR/include/pybind/optimizer_wrapper.hpp:35-40)."""
    optimizer: int = _lib.OPT_SGD
    update_type: int = _lib.UPDATE_LOCAL
    lr: float = 0.001
    beta1: float = 0.9
    beta2: float = 0.999
    epsilon: float = 1e-7
    initial_accu_value: float = 0.0
    momentum_factor: float = 0.0
    atomic_update: bool = True
    scaler: float = 1.0


def key_type_of(t: torch.Tensor) -> int:
    if t.dtype == torch.int64:
        return _lib.KEY_I64
    if t.dtype in (torch.int32, torch.uint32):
        return _lib.KEY_U32
    raise TypeError(f"keys must be int64 (long long) or uint32/int32 (unsigned int), got {t.dtype}")


class SparseEmbeddingHash:
    """One rank's shard of a Localized/Distributed slot sparse embedding hash."""

    def __init__(self, embedding_type: int, train_batch_size: int, evaluate_batch_size: int,
                 max_vocabulary_size_per_gpu: int, embedding_vec_size: int, max_feature_num: int,
                 slot_num: int, combiner: int, opt: OptParams,
                 slot_size_array: Optional[Sequence[int]] = None, key_dtype=torch.int64,
                 out_dtype=torch.float32, rank: int = 0, world: int = 1, seed: int = 0,
                 device: Optional[torch.device] = None):
        if not torch.cuda.is_available():
            raise RuntimeError("hugectr_amd needs a HIP device (MI355X); there is no CPU fallback")
        self.device = device or torch.device("cuda", torch.cuda.current_device())
        self.key_dtype = key_dtype
        self.out_dtype = out_dtype
        self.opt = opt
        self.embedding_type = embedding_type
        self.embedding_vec_size = embedding_vec_size
        self.slot_num = slot_num
        self.rank, self.world = rank, world
        self.train_batch_size, self.evaluate_batch_size = train_batch_size, evaluate_batch_size
        p = _lib.EmbeddingParams()
        p.embedding_type = embedding_type
        p.key_type = _lib.KEY_I64 if key_dtype == torch.int64 else _lib.KEY_U32
        p.out_dtype = _TORCH_TO_EMB[out_dtype]
        p.train_batch_size = train_batch_size
        p.evaluate_batch_size = evaluate_batch_size
        p.max_vocabulary_size_per_gpu = max_vocabulary_size_per_gpu
        p.embedding_vec_size = embedding_vec_size
        p.max_feature_num = max_feature_num
        p.slot_num = slot_num
        p.combiner = combiner
        self._slot_sizes = None
        if slot_size_array:
            assert len(slot_size_array) == slot_num, "slot_size_array must have slot_num entries"
            self._slot_sizes = (ctypes.c_size_t * slot_num)(*[int(x) for x in slot_size_array])
            p.slot_size_array = ctypes.cast(self._slot_sizes, ctypes.POINTER(ctypes.c_size_t))
        p.optimizer, p.update_type, p.lr = opt.optimizer, opt.update_type, opt.lr
        p.beta1, p.beta2, p.epsilon = opt.beta1, opt.beta2, opt.epsilon
        p.initial_accu_value = opt.initial_accu_value
        p.momentum_factor = opt.momentum_factor
        p.atomic_update = 1 if opt.atomic_update else 0
        p.scaler = opt.scaler
        p.rank, p.world, p.seed = rank, world, seed
        self._h = ctypes.c_void_p()
        with torch.cuda.device(self.device):
            check(lib.hctr_emb_create(ctypes.byref(p), ctypes.byref(self._h)))
        self.slots_on_rank = int(lib.hctr_emb_slots_on_rank(self._h))
        self.max_vocabulary_size_per_gpu = int(lib.hctr_emb_get_max_vocabulary_size(self._h))
        self._top_grad = None  # keeps the gradient tensor alive until update_params

    def __del__(self):
        h = getattr(self, "_h", None)
        if h is not None and h.value:
            lib.hctr_emb_destroy(h)
            self._h = ctypes.c_void_p()

    # -- IEmbedding verbs -------------------------------------------------------------------------
    def init_params(self):
        check(lib.hctr_emb_init_params(self._h, stream_ptr()))

    def forward(self, is_train: bool, row_offset: torch.Tensor, keys: torch.Tensor,
                nnz: Optional[int] = None, out: Optional[torch.Tensor] = None) -> torch.Tensor:
        """row_offset [batch*slot_num+1], keys [nnz]: the full-batch CSR.  Returns the pooled
        vectors this rank owns: [batch, slots_on_rank, D] (localized) / [batch, slot_num, D]
        partial sums (distributed) -- i.e. the collective's send buffer."""
        batch = self.train_batch_size if is_train else self.evaluate_batch_size
        assert row_offset.is_cuda and row_offset.is_contiguous()
        assert row_offset.numel() == batch * self.slot_num + 1, "row_offset must have batch*slots+1"
        self._check_key_types(row_offset, keys)
        if nnz is None:
            nnz = int(keys.numel())
        if out is None:
            out = torch.empty((batch, self.slots_on_rank, self.embedding_vec_size),
                              dtype=self.out_dtype, device=self.device)
        check(lib.hctr_emb_forward(self._h, 1 if is_train else 0, ptr(row_offset), ptr(keys), nnz,
                                   ptr(out), stream_ptr()))
        return out

    def forward_scale(self, is_train: bool, out_local: torch.Tensor) -> torch.Tensor:
        """SparseEmbeddingFunctors::forward_scale: the distributed embedding with combiner mean on
        world > 1 GPUs divides AFTER the reduce-scatter, by each bucket's key count over all GPUs
        (R/HugeCTR/include/embeddings/distributed_slot_sparse_embedding_hash.hpp:181-197).
        out_local [batch/world, slot_num, D] = this rank's reduce-scatter output, scaled in place;
        a no-op in every other configuration."""
        assert out_local.is_cuda and out_local.is_contiguous() and out_local.dtype == self.out_dtype
        batch = self.train_batch_size if is_train else self.evaluate_batch_size
        assert out_local.numel() == batch // self.world * self.slot_num * self.embedding_vec_size
        check(lib.hctr_emb_forward_scale(self._h, 1 if is_train else 0, ptr(out_local),
                                         stream_ptr()))
        return out_local

    def _check_key_types(self, row_offset: torch.Tensor, keys: torch.Tensor):
        """the C ABI takes raw pointers: keys and row offsets must have the handle's key width"""
        want = 8 if self.key_dtype == torch.int64 else 4
        for name, t in (("row_offset", row_offset), ("keys", keys)):
            if t.element_size() != want or t.is_floating_point():
                raise TypeError(f"{name} is {t.dtype}, this embedding takes "
                                f"{'int64' if want == 8 else 'uint32 / int32'} "
                                "(solver.i64_input_key decides)")

    def index(self, is_train: bool, row_offset: torch.Tensor, keys: torch.Tensor,
              nnz: Optional[int] = None):
        """index stage only (filter + hash): rows in value_index(); no gather"""
        self._check_key_types(row_offset, keys)
        if nnz is None:
            nnz = int(keys.numel())
        check(lib.hctr_emb_index(self._h, 1 if is_train else 0, ptr(row_offset), ptr(keys), nnz,
                                 stream_ptr()))

    def index_ahead(self, row_offset: torch.Tensor, keys: torch.Tensor):
        """index stage of the training batch AFTER the current one, on the current torch stream
        (one GPU): the current batch -- its rows, its pending update -- stays what it is until
        index_adopt().  The stream must be ordered behind the previous index stage."""
        self._check_key_types(row_offset, keys)
        check(lib.hctr_emb_index_ahead(self._h, ptr(row_offset), ptr(keys), int(keys.numel()),
                                       stream_ptr()))

    def index_adopt(self):
        """the batch indexed ahead becomes the current training batch (no launch)"""
        check(lib.hctr_emb_index_adopt(self._h))

    def update_rows(self, rows: torch.Tensor, grads: torch.Tensor, row_offset: torch.Tensor):
        """sparse optimizer on (row, gradient) entries: entry i updates rows[i] with grads[i]
        (entries naming the same row are summed in entry order); row_offset = arange(n + 1)"""
        n = rows.numel()
        assert grads.is_contiguous() and row_offset.dtype == torch.int64
        check(lib.hctr_emb_update_rows(self._h, n, ptr(row_offset), ptr(rows), ptr(grads),
                                       _TORCH_TO_EMB[grads.dtype], stream_ptr()))

    def backward(self, top_grad: torch.Tensor):
        assert top_grad.is_cuda and top_grad.is_contiguous() and top_grad.dtype == self.out_dtype
        self._top_grad = top_grad
        check(lib.hctr_emb_backward(self._h, ptr(top_grad), stream_ptr()))

    def get_wgrad(self) -> torch.Tensor:
        assert self._top_grad is not None
        w = torch.empty_like(self._top_grad)
        check(lib.hctr_emb_get_wgrad(self._h, ptr(w), stream_ptr()))
        return w

    def update_params(self):
        if not getattr(self, "_trainable", True):  # IEmbedding::freeze (embedding.hpp:64-66)
            self._top_grad = None
            return
        check(lib.hctr_emb_update_params(self._h, stream_ptr()))
        self._top_grad = None

    def freeze(self):
        self._trainable = False

    def unfreeze(self):
        self._trainable = True

    def is_trainable(self) -> bool:
        return getattr(self, "_trainable", True)

    # -- optimizer state files (SparseEmbeddingFunctors::dump_opt_states / load_opt_states,
    #    R/HugeCTR/src/embeddings/opt_states_functor.cu:24-250): one file; for every state tensor of
    #    the optimizer in turn (Adam: m, v; AdaGrad / Momentum / Nesterov: one) the raw
    #    [max_vocabulary_size_per_gpu, D] arrays of all GPUs in rank order, in the embedding type
    #    (fp16 when the embedding runs in fp16, float otherwise)
    def _opt_state_count(self) -> int:
        k = 0
        while self.opt_state(k) is not None:
            k += 1
        return k

    def dump_opt_states(self, write_path: str):
        import torch.distributed as dist
        ns = self._opt_state_count()
        world, rank = (dist.get_world_size(), dist.get_rank()) if dist.is_initialized() else (1, 0)
        np_dt = np.float16 if self.out_dtype == torch.float16 else np.float32
        mine = self.max_vocabulary_size_per_gpu * self.embedding_vec_size * np.dtype(np_dt).itemsize
        sizes = [mine]
        if world > 1:
            sizes = [None] * world
            dist.all_gather_object(sizes, mine)
        total = sum(sizes)
        if rank == 0:
            with open(write_path, "wb") as f:
                f.truncate(total * ns)
        if world > 1:
            dist.barrier()
        if ns:
            mm = np.memmap(write_path, dtype=np.uint8, mode="r+")
            for k in range(ns):
                off = k * total + sum(sizes[:rank])
                a = self.opt_state(k).detach().cpu().numpy().astype(np_dt)
                mm[off:off + mine] = a.reshape(-1).view(np.uint8)
            mm.flush()
            del mm
        if world > 1:
            dist.barrier()

    def load_opt_states(self, read_path: str):
        import torch.distributed as dist
        ns = self._opt_state_count()
        world, rank = (dist.get_world_size(), dist.get_rank()) if dist.is_initialized() else (1, 0)
        np_dt = np.float16 if self.out_dtype == torch.float16 else np.float32
        n = self.max_vocabulary_size_per_gpu * self.embedding_vec_size
        mine = n * np.dtype(np_dt).itemsize
        sizes = [mine]
        if world > 1:
            sizes = [None] * world
            dist.all_gather_object(sizes, mine)
        total = sum(sizes)
        if os.path.getsize(read_path) != total * ns:
            raise _lib.HugeCTRAmdError(f"{read_path}: {os.path.getsize(read_path)} bytes, the "
                                       f"optimizer states of this embedding take {total * ns}")
        mm = np.memmap(read_path, dtype=np.uint8, mode="r")
        for k in range(ns):
            off = k * total + sum(sizes[:rank])
            a = np.frombuffer(mm[off:off + mine].tobytes(), dtype=np_dt).astype(np.float32)
            self.opt_state(k).copy_(torch.from_numpy(a).view(self.max_vocabulary_size_per_gpu,
                                                             self.embedding_vec_size))
        del mm

    def set_learning_rate(self, lr: float):
        check(lib.hctr_emb_set_learning_rate(self._h, lr))

    def get_vocabulary_size(self) -> int:
        n = ctypes.c_size_t()
        check(lib.hctr_emb_get_vocabulary_size(self._h, stream_ptr(), ctypes.byref(n)))
        return int(n.value)

    def get_max_vocabulary_size(self) -> int:
        return self.max_vocabulary_size_per_gpu

    def get_params_num(self) -> int:
        return self.max_vocabulary_size_per_gpu * self.embedding_vec_size

    def check_overflow(self):
        check(lib.hctr_emb_check_overflow(self._h, stream_ptr()))

    def poll_overflow(self):
        """check_overflow without the host synchronisation: raises once a completed copy of the
        error flags shows that the table ran out of rows (at most two calls late)"""
        check(lib.hctr_emb_poll_overflow(self._h, stream_ptr()))

    def reset(self):
        check(lib.hctr_emb_reset(self._h, stream_ptr()))

    def dump_parameters(self):
        """-> (keys int64 [n], slot_id int64 [n], emb_vector fp32 [n, D]) device tensors."""
        n = self.get_vocabulary_size()
        keys = torch.empty(max(n, 1), dtype=torch.int64, device=self.device)
        slot = torch.empty(max(n, 1), dtype=torch.int64, device=self.device)
        vec = torch.empty((max(n, 1), self.embedding_vec_size), dtype=torch.float32,
                          device=self.device)
        cnt = ctypes.c_size_t()
        check(lib.hctr_emb_dump(self._h, ptr(keys), ptr(slot), ptr(vec), ctypes.byref(cnt),
                                stream_ptr()))
        n = int(cnt.value)
        return keys[:n], slot[:n], vec[:n]

    def load_parameters(self, keys: torch.Tensor, slot_id: Optional[torch.Tensor],
                        vectors: torch.Tensor):
        keys = keys.to(self.device, torch.int64).contiguous()
        vectors = vectors.to(self.device, torch.float32).contiguous()
        if slot_id is not None:
            slot_id = slot_id.to(self.device, torch.int64).contiguous()
        check(lib.hctr_emb_load(self._h, ptr(keys), ptr(slot_id), ptr(vectors), keys.numel(),
                                stream_ptr()))

    # -- measurement support ----------------------------------------------------------------------
    PROFILE_STAGES = ("gather_pool", "hash_index", "sort", "segmented_update")

    def profiling(self, enable: bool):
        check(lib.hctr_emb_profiling(self._h, 1 if enable else 0))

    def profile(self) -> dict:
        """{stage: (total_ms, launches)} measured with hipEvents on the launch stream."""
        res = {}
        for i, name in enumerate(self.PROFILE_STAGES):
            ms, n = ctypes.c_double(), ctypes.c_uint64()
            check(lib.hctr_emb_profile_get(self._h, i, ctypes.byref(ms), ctypes.byref(n)))
            res[name] = (ms.value, int(n.value))
        return res

    # -- raw views (owned by the handle) ----------------------------------------------------------
    def _view(self, addr: int, shape, dtype):
        n = 1
        for s in shape:
            n *= s
        nbytes = n * torch.empty((), dtype=dtype).element_size()

        class _Holder:
            pass

        h = _Holder()
        h.__cuda_array_interface__ = {"shape": (nbytes,), "typestr": "|u1",
                                      "data": (int(addr), False), "version": 3}
        return torch.as_tensor(h, device=self.device).view(dtype).view(*shape)

    def table(self) -> torch.Tensor:
        """fp32 [max_vocab, D] view of the device table (no copy)."""
        addr = lib.hctr_emb_table_ptr(self._h)
        return self._view(addr, (self.max_vocabulary_size_per_gpu, self.embedding_vec_size),
                          torch.float32)

    def opt_state(self, k: int) -> Optional[torch.Tensor]:
        """[max_vocab, D] view of optimizer state k (no copy): fp32, or fp16 with fp16 embeddings
        (OptimizerTensor<TypeEmbeddingComp>, R/HugeCTR/include/optimizer.hpp:284-296)"""
        addr = lib.hctr_emb_opt_state_ptr(self._h, k)
        if not addr:
            return None
        return self._view(addr, (self.max_vocabulary_size_per_gpu, self.embedding_vec_size),
                          torch.float16 if self.out_dtype == torch.float16 else torch.float32)

    def value_index(self, nnz: int) -> torch.Tensor:
        addr = lib.hctr_emb_value_index_ptr(self._h)
        return self._view(addr, (nnz,), torch.int64)


def forward_reorder(recv: torch.Tensor, batch_per_gpu: int, slot_num: int, vec: int,
                    world: int, out: Optional[torch.Tensor] = None) -> torch.Tensor:
    """[gpu][b][slot_in_gpu][D] (all-to-all receive buffer) -> [b][slot][D]."""
    if out is None:
        out = torch.empty((batch_per_gpu, slot_num, vec), dtype=recv.dtype, device=recv.device)
    check(lib.hctr_forward_reorder(batch_per_gpu, slot_num, vec, world, ptr(recv), ptr(out),
                                   _TORCH_TO_EMB[recv.dtype], stream_ptr()))
    return out


def backward_reorder(grad: torch.Tensor, batch_per_gpu: int, slot_num: int, vec: int,
                     world: int) -> torch.Tensor:
    """[b][slot][D] top gradient -> [gpu][b][slot_in_gpu][D] (all-to-all send buffer)."""
    out = torch.empty(batch_per_gpu * slot_num * vec, dtype=grad.dtype, device=grad.device)
    check(lib.hctr_backward_reorder(batch_per_gpu, slot_num, vec, world, ptr(grad), ptr(out),
                                    _TORCH_TO_EMB[grad.dtype], stream_ptr()))
    return out
