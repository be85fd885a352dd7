"""Multi-GPU exchange of the localized / distributed slot embeddings: one process per GPU,
torch.distributed (backend "nccl" == RCCL over xGMI on ROCm; "gloo" in the CPU tests).

Wire layout and counts follow the reference exactly
(R/HugeCTR/src/embeddings/all2all_forward_functor.cu:157-264, all2all_backward_functor.cu,
reduce_scatter_functor.cu:22-65, all_gather_functor.cu:58):

  localized forward : rank r sends to peer j the sample slice j of its pooled vectors,
                      (B/N) * S_r * D elements, and receives (B/N) * S_j * D from j
                      -> recv buffer [peer][b_local][slot_in_peer][D] -> forward_reorder
  localized backward: the mirror all-to-all of the top gradients
  distributed forward: reduce-scatter(sum) of [B, S, D] partial sums -> [B/N, S, D]
  distributed backward: all-gather of the [B/N, S, D] top gradients -> [B, S, D]

Nothing here computes embeddings; this module only moves buffers, so it runs unchanged on CPU
tensors with gloo (tests/test_parallel_cpu.py).
"""
from __future__ import annotations

import torch
import torch.distributed as dist


def _staged(t: torch.Tensor) -> bool:
    """gloo cannot run these collectives on device tensors: stage through the host.  Only the
    2-ranks-on-one-GPU tests and `HCTR_BENCH_BACKEND=gloo` take this path; RCCL never does."""
    return t.is_cuda and dist.get_backend() == "gloo"


def all_to_all_single(out, inp, out_splits, in_splits, group=None, async_op=False):
    if _staged(inp):
        o = torch.empty(out.shape, dtype=out.dtype)
        dist.all_to_all_single(o, inp.cpu(), out_splits, in_splits, group=group)
        out.copy_(o)
        return None
    return dist.all_to_all_single(out, inp, out_splits, in_splits, group=group, async_op=async_op)


def all_reduce(t: torch.Tensor, group=None):
    if _staged(t):
        c = t.cpu()
        dist.all_reduce(c, group=group)
        t.copy_(c)
    else:
        dist.all_reduce(t, group=group)


def slots_on_rank(slot_num: int, rank: int, world: int) -> int:
    """R/HugeCTR/include/embeddings/localized_slot_sparse_embedding_hash.hpp:176-183"""
    return slot_num // world + (1 if rank < slot_num % world else 0)


def localized_split_sizes(batch: int, slot_num: int, vec: int, rank: int, world: int):
    """element counts (send_to_peer[j], recv_from_peer[j]) of the embedding-vector all-to-all."""
    bpg = batch // world
    send = [bpg * slots_on_rank(slot_num, rank, world) * vec] * world
    recv = [bpg * slots_on_rank(slot_num, j, world) * vec for j in range(world)]
    return send, recv


def reorder_row_map(batch_per_gpu: int, slot_num: int, world: int) -> torch.Tensor:
    """int32 [batch_per_gpu, slot_num]: row of (local sample b, global slot s) in the all-to-all
    receive buffer [peer g][b][slot in g][D] of the localized embedding, i.e. the index form of
    forward_reorder (R/HugeCTR/src/embeddings/forward_reorder_functor.cu:43-57: slot s lives on GPU
    s % N as its (s / N)-th slot).  A bijection onto the batch_per_gpu * slot_num rows: the same
    map is the scatter of backward_reorder."""
    s_of = [slots_on_rank(slot_num, g, world) for g in range(world)]
    base = [batch_per_gpu * sum(s_of[:g]) for g in range(world)]
    b = torch.arange(batch_per_gpu, dtype=torch.int64).view(-1, 1)
    sl = torch.arange(slot_num, dtype=torch.int64).view(1, -1)
    g, j = sl % world, sl // world
    row = torch.tensor(base, dtype=torch.int64)[g] + b * torch.tensor(s_of, dtype=torch.int64)[g] + j
    return row.to(torch.int32)


class LocalizedExchange:
    """all-to-all of pooled vectors (forward) and of top gradients (backward)."""

    def __init__(self, batch: int, slot_num: int, vec: int, group=None,
                 always_collective: bool = False):
        """always_collective: issue the collective even in a group of one (a self-send through
        the communicator; tests/test_rccl_gpu.py runs the RCCL code path on a 1-GPU box this way)"""
        self.group = group
        self.world = dist.get_world_size(group) if dist.is_initialized() else 1
        self.rank = dist.get_rank(group) if dist.is_initialized() else 0
        self.single = self.world == 1 and not (always_collective and dist.is_initialized())
        assert batch % self.world == 0, "batch must be divisible by the number of GPUs"
        self.batch, self.slot_num, self.vec = batch, slot_num, vec
        self.bpg = batch // self.world
        self.send, self.recv = localized_split_sizes(batch, slot_num, vec, self.rank, self.world)

    def forward(self, pooled: torch.Tensor) -> torch.Tensor:
        """pooled [B, S_r, D] (== [peer][B/N][S_r][D]) -> recv buffer [sum_j (B/N) S_j D]"""
        flat = pooled.reshape(-1)
        assert flat.numel() == sum(self.send)
        if self.single:
            return flat
        out = torch.empty(sum(self.recv), dtype=pooled.dtype, device=pooled.device)
        all_to_all_single(out, flat, self.recv, self.send, group=self.group)
        return out

    # -- non-blocking forms: the collective runs on the communicator's stream while the caller
    #    keeps launching compute; `work.wait()` orders the current stream after it ---------------
    def forward_async(self, pooled: torch.Tensor):
        flat = pooled.reshape(-1)
        assert flat.numel() == sum(self.send)
        if self.single:
            return flat, None
        out = torch.empty(sum(self.recv), dtype=pooled.dtype, device=pooled.device)
        work = all_to_all_single(out, flat, self.recv, self.send, group=self.group, async_op=True)
        return out, work

    def backward_async(self, grad_send: torch.Tensor, out: torch.Tensor):
        """grad_send [sum_j (B/N) S_j D] -> out (flat view of [B, S_r, D], caller-owned)"""
        flat = grad_send.reshape(-1)
        assert flat.numel() == sum(self.recv) and out.numel() == sum(self.send)
        if self.single:
            out.copy_(flat)
            return None
        return all_to_all_single(out, flat, self.send, self.recv, group=self.group, async_op=True)

    def backward(self, grad_send: torch.Tensor) -> torch.Tensor:
        """grad_send: backward_reorder output [sum_j (B/N) S_j D] -> [B, S_r, D] top gradients"""
        flat = grad_send.reshape(-1)
        assert flat.numel() == sum(self.recv)
        s_r = slots_on_rank(self.slot_num, self.rank, self.world)
        if self.single:
            return flat.view(self.batch, s_r, self.vec)
        out = torch.empty(sum(self.send), dtype=grad_send.dtype, device=grad_send.device)
        all_to_all_single(out, flat, self.send, self.recv, group=self.group)
        return out.view(self.batch, s_r, self.vec)


class DistributedExchange:
    """reduce-scatter (forward) / all-gather (backward) of the distributed-slot embedding."""

    def __init__(self, batch: int, slot_num: int, vec: int, group=None,
                 always_collective: bool = False):
        self.group = group
        self.world = dist.get_world_size(group) if dist.is_initialized() else 1
        self.single = self.world == 1 and not (always_collective and dist.is_initialized())
        self.batch, self.slot_num, self.vec = batch, slot_num, vec
        self.bpg = batch // self.world

    def forward(self, partial: torch.Tensor) -> torch.Tensor:
        if self.single:
            return partial.view(self.bpg, self.slot_num, self.vec)
        out = torch.empty((self.bpg, self.slot_num, self.vec), dtype=partial.dtype,
                          device=partial.device)
        if _staged(partial):  # (gloo reduces in fp32 on the host: the 2-ranks-on-one-GPU tests)
            o = torch.empty(out.shape, dtype=torch.float32)
            dist.reduce_scatter_tensor(o, partial.float().cpu().contiguous(), op=dist.ReduceOp.SUM,
                                       group=self.group)
            out.copy_(o)
            return out
        dist.reduce_scatter_tensor(out, partial.contiguous(), op=dist.ReduceOp.SUM, group=self.group)
        return out

    def backward(self, grad: torch.Tensor) -> torch.Tensor:
        if self.single:
            return grad.view(self.batch, self.slot_num, self.vec)
        out = torch.empty((self.batch, self.slot_num, self.vec), dtype=grad.dtype, device=grad.device)
        if _staged(grad):
            o = torch.empty(out.shape, dtype=grad.dtype)
            dist.all_gather_into_tensor(o, grad.contiguous().cpu(), group=self.group)
            out.copy_(o)
            return out
        dist.all_gather_into_tensor(out, grad.contiguous(), group=self.group)
        return out
