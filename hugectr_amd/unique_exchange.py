"""Unique-row form of the localized embedding exchange (multi-GPU, one key per bucket).

`LocalizedExchange` moves one pooled vector per (sample, slot) each way, as the reference does
(R/HugeCTR/src/embeddings/all2all_forward_functor.cu:157-264).  With one-hot buckets the pooled
vector is the table row itself, and power-law keys repeat rows heavily, so this exchange ships every
distinct row ONCE per destination GPU plus an 8-byte (index, bucket) pair per position, and returns
per-row gradient SUMS (fp32) instead of per-sample gradients: the payload that crosses xGMI drops
by the duplication factor of the batch (~7x on the Criteo-1TB shape at alpha = 1.1), and the owner's
sparse update sees one entry per distinct row and peer instead of one per sample.

    owner r:  index stage -> hctr_uniq_plan (sort by (peer, row), runs, distinct rows) ->
              counts all-gather + one host sync (the variable all-to-all needs host-side sizes) ->
              gather distinct rows -> all-to-all(meta, fixed size), all-to-all(rows, variable)
    receiver: hctr_uniq_expand -> E [B/N, S, D]   ... dense tower ...
              dE -> hctr_updater_reduce_presorted (per-row sums over the sorted list) ->
              all-to-all(sums, variable)
    owner r:  hctr_emb_update_rows on (row, sum) entries

Summation order differs from the per-sample exchange (gradients of one row are first added per
destination GPU, then across GPUs), so results agree to fp32 rounding, not bit for bit.
"""
from __future__ import annotations

import ctypes
from typing import List, Optional

import torch
import torch.distributed as dist

from . import _lib
from ._lib import check, lib, ptr, stream_ptr
from .parallel import slots_on_rank

_EMB_DT = {torch.float32: _lib.F32, torch.float16: _lib.F16, torch.bfloat16: _lib.BF16}


def _staged() -> bool:
    """gloo cannot move device tensors through all_to_all: stage through the host (tests only)"""
    return dist.get_backend() == "gloo"


def _a2a(out: torch.Tensor, inp: torch.Tensor, out_splits: List[int], in_splits: List[int], group):
    if _staged():
        o = torch.empty(out.shape, dtype=out.dtype)
        dist.all_to_all_single(o, inp.cpu(), out_splits, in_splits, group=group)
        out.copy_(o)
    else:
        dist.all_to_all_single(out, inp, out_splits, in_splits, group=group)


def _all_gather(out: torch.Tensor, inp: torch.Tensor, group):
    if _staged():
        o = torch.empty(out.shape, dtype=out.dtype)
        dist.all_gather_into_tensor(o, inp.cpu(), group=group)
        out.copy_(o)
    else:
        dist.all_gather_into_tensor(out, inp, group=group)


class UniqueExchange:
    def __init__(self, emb, batch_per_gpu: int, slot_num: int, vec: int, group=None,
                 sum_dtype=torch.float32):
        """sum_dtype: wire type of the per-row gradient sums (accumulated in fp32 either way).
        fp32 keeps the sums exact; a 16-bit type halves the return payload at the precision class
        of the 16-bit per-sample gradients the rows exchange ships (use set_sum_dtype to switch)."""
        self.emb, self.group = emb, group
        self.sum_dtype = sum_dtype
        self.world = dist.get_world_size(group)
        self.rank = dist.get_rank(group)
        self.bl, self.S, self.D = batch_per_gpu, slot_num, vec
        self.dtype = emb.out_dtype
        self.s_r = slots_on_rank(slot_num, self.rank, self.world)
        self.P = self.world * batch_per_gpu * self.s_r      # positions this rank owns
        self.ppp = batch_per_gpu * self.s_r                  # ... per destination GPU
        self.Q = batch_per_gpu * slot_num                    # positions this rank receives
        dev = emb.device
        assert (vec * torch.empty(0, dtype=self.dtype).element_size()) % 16 == 0
        self._h = ctypes.c_void_p()
        check(lib.hctr_uniq_create(max(self.P, 1), ctypes.byref(self._h)))
        self._upd = ctypes.c_void_p()
        check(lib.hctr_updater_create(self.Q, self.Q, vec, ctypes.byref(self._upd)))
        i32, i64 = torch.int32, torch.int64
        # plan outputs are double-buffered: prefetch() fills the other set for the next batch on
        # a side stream while this batch's set is still in use by expand / update_rows
        self._plans = [{"meta": torch.empty((max(self.P, 1), 2), dtype=i32, device=dev),
                        "urow": torch.empty(max(self.P, 1), dtype=i64, device=dev),
                        "peer_off": torch.zeros(self.world + 1, dtype=i64, device=dev),
                        "counts": torch.zeros((self.world, self.world), dtype=i64, device=dev),
                        "counts_host": torch.zeros((self.world, self.world), dtype=i64).pin_memory(),
                        "meta_recv": torch.empty((max(batch_per_gpu * slot_num, 1), 2), dtype=i32,
                                                 device=dev),
                        "keys": None, "event": None, "host_event": None} for _ in range(2)]
        self._cur = 0
        self.meta, self.urow, self.peer_off = (self._plans[0][k] for k in ("meta", "urow", "peer_off"))
        self._side = torch.cuda.Stream(device=dev)
        self._main_done = None  # event on the caller's stream after which the side stream may plan
        self.rows_send = torch.empty((max(self.P, 1), vec), dtype=self.dtype, device=dev)
        self.rows_recv = torch.empty((self.Q, vec), dtype=self.dtype, device=dev)
        self.sorted_rows = torch.empty(self.Q, dtype=i32, device=dev)
        self.sorted_buckets = torch.empty(self.Q, dtype=i32, device=dev)
        self.sums = torch.empty((self.Q, vec), dtype=torch.float32, device=dev)
        self.grads_back = torch.empty((max(self.P, 1), vec), dtype=torch.float32, device=dev)
        self._sums16 = None   # 16-bit staging of the sums / received sums, allocated on demand
        self._back16 = None
        self.arange = torch.arange(max(self.P, self.Q) + 1, dtype=i64, device=dev)
        s_of = [slots_on_rank(slot_num, j, self.world) for j in range(self.world)]
        self.meta_send_splits = [self.ppp * 2] * self.world
        self.meta_recv_splits = [batch_per_gpu * s * 2 for s in s_of]
        q = [0]
        for s in s_of:
            q.append(q[-1] + batch_per_gpu * s)
        self.q_off = torch.tensor(q, dtype=i64, device=dev)
        self.u_send: Optional[List[int]] = None
        self.u_recv: Optional[List[int]] = None
        self._work = None

    def __del__(self):
        if getattr(self, "_h", None):
            lib.hctr_uniq_destroy(self._h)
            lib.hctr_updater_destroy(self._upd)
            self._h = None

    # -- forward: begin() makes a plan current (prefetched or computed in line); finish() reads the
    #    counts on the host and runs the variable part of the exchange --------------------------------
    def _plan(self, slot: int, row_offset: torch.Tensor, keys: torch.Tensor):
        """index stage + plan of one batch into plan set `slot`, on the current stream"""
        emb, pl = self.emb, self._plans[slot]
        if keys.numel() != self.world * self.bl * self.S:
            raise _lib.HugeCTRAmdError(
                "the unique-row exchange needs exactly one key per (sample, slot) bucket; use the "
                "per-sample exchange (LocalizedExchange) for multi-hot input")
        emb.index(True, row_offset, keys)
        if self.P > 0:
            vi = emb.value_index(self.P)
            check(lib.hctr_uniq_plan(self._h, self.P, self.ppp, self.bl, self.s_r, self.S,
                                     self.rank, self.world, ptr(vi),
                                     emb.get_max_vocabulary_size(), ptr(pl["meta"]),
                                     ptr(pl["urow"]), ptr(pl["peer_off"]), stream_ptr()))
        # counts of distinct rows per (owner, destination): every rank needs its row and its column
        mine = (pl["peer_off"][1:] - pl["peer_off"][:-1]).contiguous()
        _all_gather(pl["counts"].view(-1), mine, self.group)
        pl["counts_host"].copy_(pl["counts"], non_blocking=True)
        he = torch.cuda.Event()
        he.record(torch.cuda.current_stream())
        pl["host_event"] = he
        # fixed-size part of the payload: (index, bucket) pairs
        _a2a(pl["meta_recv"].view(-1), pl["meta"].view(-1)[:self.P * 2], self.meta_recv_splits,
             self.meta_send_splits, self.group)
        pl["keys"] = keys

    def prefetch(self, row_offset: torch.Tensor, keys: torch.Tensor):
        """Index stage + plan of the NEXT batch -- and the exchange of its counts and (index,
        bucket) pairs, with the counts copied to pinned host memory -- on a side stream, to be
        called once this batch's forward_begin() has been issued.  The next step then needs no
        host-device synchronisation at all: the host only waits for a copy that finished long ago.  Only the key -> row map is touched (new keys are
        inserted a step early, which changes nothing they map to); table values are not read, so
        running ahead of this batch's update is exact -- the reference's inter-iteration overlap
        of the index calculation does the same."""
        nxt = 1 - self._cur
        self._side.wait_stream(torch.cuda.current_stream())
        with torch.cuda.stream(self._side):
            self._plan(nxt, row_offset, keys)
            ev = torch.cuda.Event()
            ev.record(self._side)
        self._plans[nxt]["event"] = ev

    def drain(self):
        """wait for an outstanding prefetch (call before using the embedding outside this class)"""
        self._side.synchronize()

    def forward_begin(self, row_offset: torch.Tensor, keys: torch.Tensor):
        nxt = 1 - self._cur
        pl = self._plans[nxt]
        if pl["event"] is not None and pl["keys"] is keys:
            torch.cuda.current_stream().wait_event(pl["event"])  # planned ahead by prefetch()
            self._cur = nxt
        else:
            self._side.synchronize()  # a stale prefetch must not race the inline plan
            self._plan(self._cur, row_offset, keys)
            pl = self._plans[self._cur]
        pl["event"] = None
        self._pl = pl
        self.meta, self.urow, self.peer_off = pl["meta"], pl["urow"], pl["peer_off"]
        self.meta_recv, self.counts = pl["meta_recv"], pl["counts"]

    def forward_finish(self, out: Optional[torch.Tensor] = None, indexed: bool = False):
        """indexed = False: returns E [batch_per_gpu, slot_num, D].  indexed = True: returns
        (rows [R, D], row_of int32 [batch_per_gpu, slot_num]) for `interaction_indexed` -- the
        expanded tensor is never written."""
        emb, W, D = self.emb, self.world, self.D
        # the variable all-to-all needs the counts on the host: wait for their (pinned) copy only --
        # with a prefetched plan that copy finished during the previous step
        self._pl["host_event"].synchronize()
        c = self._pl["counts_host"]
        self.u_send = [int(x) for x in c[self.rank]]
        self.u_recv = [int(x) for x in c[:, self.rank]]
        n_send, n_recv = sum(self.u_send), sum(self.u_recv)
        check(lib.hctr_uniq_gather_rows(n_send, D, ptr(self.urow), lib.hctr_emb_table_ptr(emb._h),
                                        ptr(self.rows_send), _EMB_DT[self.dtype], stream_ptr()))
        _a2a(self.rows_recv.view(-1)[:n_recv * D], self.rows_send.view(-1)[:n_send * D],
             [u * D for u in self.u_recv], [u * D for u in self.u_send], self.group)
        r_off = torch.zeros(W + 1, dtype=torch.int64, device=emb.device)
        torch.cumsum(self.counts[:, self.rank], 0, out=r_off[1:])
        row_of = None
        if indexed:
            row_of = torch.empty((self.bl, self.S), dtype=torch.int32, device=emb.device)
        elif out is None:
            out = torch.empty((self.bl, self.S, D), dtype=self.dtype, device=emb.device)
        check(lib.hctr_uniq_expand(self.Q, W, ptr(self.q_off), ptr(r_off), ptr(self.meta_recv),
                                   ptr(self.rows_recv), D, _EMB_DT[self.dtype],
                                   None if indexed else ptr(out), ptr(self.sorted_rows),
                                   ptr(self.sorted_buckets), ptr(row_of), stream_ptr()))
        if indexed:
            return self.rows_recv[:max(n_recv, 1)], row_of
        return out

    def forward(self, row_offset: torch.Tensor, keys: torch.Tensor,
                out: Optional[torch.Tensor] = None) -> torch.Tensor:
        """full-batch one-hot CSR -> E [batch_per_gpu, slot_num, D] of this rank's samples"""
        self.forward_begin(row_offset, keys)
        return self.forward_finish(out)

    # -- backward: begin() may run inside autograd (as soon as dE exists); finish() applies the
    #    sparse optimizer once the sums have arrived ------------------------------------------------
    def backward_begin(self, grad: torch.Tensor):
        D = self.D
        n_send, n_recv = sum(self.u_send), sum(self.u_recv)
        grad = grad.contiguous()
        check(lib.hctr_updater_reduce_presorted(
            self._upd, self.Q, self.Q, ptr(self.arange), ptr(self.sorted_rows),
            ptr(self.sorted_buckets), ptr(grad), _EMB_DT[grad.dtype], n_recv, ptr(self.sums),
            stream_ptr()))
        if self.sum_dtype == torch.float32:
            out, inp = self.grads_back.view(-1)[:n_send * D], self.sums.view(-1)[:n_recv * D]
        else:
            if self._sums16 is None or self._sums16.dtype != self.sum_dtype:
                self._sums16 = torch.empty_like(self.sums, dtype=self.sum_dtype)
                self._back16 = torch.empty_like(self.grads_back, dtype=self.sum_dtype)
            self._sums16[:n_recv].copy_(self.sums[:n_recv])
            out, inp = self._back16.view(-1)[:n_send * D], self._sums16.view(-1)[:n_recv * D]
        osz, isz = [u * D for u in self.u_send], [u * D for u in self.u_recv]
        if _staged():
            _a2a(out, inp, osz, isz, self.group)
            self._work = None
        else:
            self._work = dist.all_to_all_single(out, inp, osz, isz, group=self.group,
                                                async_op=True)

    def set_sum_dtype(self, dtype):
        self.sum_dtype = dtype

    def backward_finish(self):
        if self._work is not None:
            self._work.wait()
            self._work = None
        n_send = sum(self.u_send)
        back = self.grads_back if self.sum_dtype == torch.float32 else self._back16
        self.emb.update_rows(self.urow[:n_send], back[:n_send], self.arange[:n_send + 1])

    def backward_and_update(self, grad: torch.Tensor):
        """dE [batch_per_gpu, slot_num, D] -> per-row sums -> owners -> sparse optimizer"""
        self.backward_begin(grad)
        self.backward_finish()
