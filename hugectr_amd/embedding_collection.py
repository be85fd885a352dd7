"""embedding_collection (EBC) on static or dynamic tables -- the engine SparseOperationKit plugs into.

Mirrors `EmbeddingTableConfig` / `EmbeddingCollectionConfig.embedding_lookup(...).shard(...)`
(R/HugeCTR/include/pybind/embedding_collection_wrapper.hpp:28-66) and the forward / backward /
update of `embedding::EmbeddingCollection` (R/HugeCTR/include/embeddings/embedding_collection.hpp:
333-406) for `RaggedStaticEmbeddingTable` storage (direct index, SGD / AdaGrad;
R/HugeCTR/embedding_storage/ragged_static_embedding.cu).

Flow per rank (one process per GPU), following SparseOperationKit's lookup_sparse
(R/sparse_operation_kit/sparse_operation_kit/lookup.py:425-541):
  all-gather the data-parallel keys  ->  hctr_ebc_route_keys (keep my shards' keys, key -> row)
  ->  hctr_forward_pool into the all-to-all send layout  ->  all-to-all of embedding vectors
  ->  hctr_ebc_network_forward (sum row-shard partials, Average scaling)  ->  [lookup][b][ev]
Backward is the mirror; the update runs on the owner with the segmented sparse optimizer.
One GPU (world = 1, sum / concat lookups): there is no peer, the send layout already is the
feature-major output, and the two reorder passes drop out -- hctr_ebc_route_whole (one index pass)
-> hctr_forward_pool_mapped straight into the output (transposed store address for batch-major)
-> the update reads the output's gradient in place (hctr_updater_set_grad_map); `_direct` below.

storage="dynamic" (embedding::DynamicEmbeddingTable, R/HugeCTR/embedding_storage/
dynamic_embedding.cu:21-330; BASELINE config 5): the local shards are classes of one hctr_det table
that grows on demand; routing keeps the raw keys, `lookup` returns per-key row addresses
(hctr_det_lookup_rows = ILookup::lookup), pooling reads through them (hctr_forward_pool_ptrs), and
the backward builds Wgrad{unique_keys, ev_start_indices, data} (hctr_ebc_local_reduce) for the
table's fused optimizer step (hctr_det_update, all seven optimizers of optimizers.cuh).

Sharding: `shard_matrix[gpu][table]` in {0,1}; a table with one owner is table-wise sharded,
with several owners row-wise (`key % num_shards` picks the owner in ascending GPU order, local row
= key // num_shards).  All tables of one collection share `ev_size` here.
"""
from __future__ import annotations

import ctypes
import os
from dataclasses import dataclass
from typing import List, Optional

import torch
import torch.distributed as dist

from . import _lib
from ._lib import check, lib, ptr, stream_ptr

_DT = {torch.float32: _lib.F32, torch.float16: _lib.F16, torch.bfloat16: _lib.BF16}
# optimizers whose step on a dynamic table runs in the static tables' sparse update, on the flat
# row store (EmbeddingCollection._dynamic_apply)
_FLAT_STEP = (_lib.OPT_SGD, _lib.OPT_ADAGRAD, _lib.OPT_ADAM, _lib.OPT_MOMENTUM_SGD)


@dataclass
class EmbeddingTableConfig:
    name: str
    max_vocabulary_size: int
    ev_size: int
    opt_params: object = None
    init_param: object = None


class EmbeddingCollectionConfig:
    def __init__(self, use_exclusive_keys: bool = True, comm_strategy: str = "Uniform"):
        self.lookups = []  # (table_config, bottom_name, top_name, combiner)
        self.shard_matrix = None
        self.shard_strategy = "mp"
        self.compression = {}  # table name -> "reduction" | "unique" (shard(compression_strategy=))
        self.top_name: Optional[str] = None  # one name for the concatenated output (train.py:398)

    def embedding_lookup(self, table_config, bottom_name, top_name, combiner):
        if isinstance(table_config, (list, tuple)):
            if isinstance(top_name, str):  # all lookups feed one [batch, sum(ev)] tensor
                self.top_name = top_name
                top_name = [top_name] * len(table_config)
            for t, b, tp, c in zip(table_config, bottom_name, top_name, combiner):
                self.lookups.append((t, b, tp, c))
        else:
            self.lookups.append((table_config, bottom_name, top_name, combiner))
        return self

    def shard(self, shard_matrix, shard_strategy="mp", compression_strategy=None):
        """shard_matrix: either [gpu][table] in {0, 1}, or the reference's form
        (embedding_collection_wrapper.hpp / samples/dlrm/train.py:404): per GPU the list of table
        NAMES it holds, with shard_strategy = [("mp", names), ("dp", names)]."""
        self.shard_matrix = [list(r) for r in shard_matrix]
        self.shard_strategy = shard_strategy
        self.compression = self._check_compression(compression_strategy)
        return self

    def _check_compression(self, compression_strategy):
        """compression_strategy: [(CompressionStrategy.Reduction | Unique, [table names]), ...] -- how
        the model-parallel lookups of a table travel (EmbeddingCollectionParam's constructor,
        R/HugeCTR/embedding/common.cpp:280-307, 323-340): a table may appear under one strategy only,
        and when the list is given it must name exactly the model-parallel tables.  Not given: the
        reference's default, Reduction for pooled lookups.  -> {table name: "reduction" | "unique"}"""
        if not compression_strategy:
            return {}
        out = {}
        for kind, names in compression_strategy:
            k = str(getattr(kind, "name", kind)).lower()
            if k not in ("reduction", "unique"):
                raise RuntimeError(f"shard: unknown CompressionStrategy {kind!r}")
            for n in names:
                if str(n) in out:
                    raise RuntimeError("Duplicate table id in different CompressionStrategy")
                out[str(n)] = k
        mp = None
        if isinstance(self.shard_strategy, (list, tuple)):
            mp = {str(n) for kind, names in self.shard_strategy if str(kind).lower() == "mp"
                  for n in names}
        elif str(self.shard_strategy).lower() == "mp":
            mp = {t.name for t, _, _, _ in self.lookups}
        if mp is not None and set(out) != mp:
            raise RuntimeError("Table ids in CompressionStrategy does not match with table ids in "
                               "TablePlacementStrategy")
        return out

    def ownership(self, tables, world) -> List[List[int]]:
        """-> [gpu][table] in {0, 1}.  A table listed on several GPUs is row-sharded over them; a
        "dp" table (replicated in the reference) is held row-sharded over its GPUs as well -- same
        results, no replica to keep in sync."""
        T = len(tables)
        sm = self.shard_matrix
        if sm is None:
            return [[1] * T for _ in range(world)]
        assert len(sm) == world, f"shard_matrix has {len(sm)} rows for {world} GPUs"
        binary = all(len(r) == T and all(isinstance(v, int) and not isinstance(v, bool) and
                                         v in (0, 1) for v in r) for r in sm)
        if binary:
            return [list(r) for r in sm]
        index = {t.name: i for i, t in enumerate(tables)}
        own = [[0] * T for _ in range(world)]
        for g, row in enumerate(sm):
            for name in row:
                own[g][index[str(name)]] = 1
        return own


class EmbeddingCollection:
    """Runtime of one rank.  keys / bucket_range passed to forward are this rank's data-parallel
    share in feature-major order: bucket = lookup * (batch/world) + b_local."""

    def __init__(self, config: EmbeddingCollectionConfig, global_batch: int, lr: float = 0.01,
                 optimizer: int = _lib.OPT_SGD, scaler: float = 1.0, epsilon: float = 1e-7,
                 initial_accu_value: float = 0.0, out_dtype=torch.float32, batch_major: bool = False,
                 key_dtype=torch.int64, max_hotness: int = 1, seed: int = 0, group=None,
                 ftrl=(0.0, 0.0, 0.0), storage: Optional[str] = None, initializer: str = "",
                 init_capacity: int = 1 << 20, key_route: str = "allgather",
                 hotness: Optional[List[int]] = None, **opt_kw):
        self.group = group
        self.world = dist.get_world_size(group) if dist.is_initialized() else 1
        self.rank = dist.get_rank(group) if dist.is_initialized() else 0
        assert key_route in ("allgather", "a2a")
        self.key_route = key_route
        self._setup(config, global_batch, lr, optimizer, scaler, epsilon, initial_accu_value,
                    out_dtype, batch_major, key_dtype, max_hotness, seed, ftrl, storage,
                    initializer, init_capacity, hotness=hotness, **opt_kw)

    @classmethod
    def for_rank(cls, rank, world, *a, **kw):
        """single-process construction of one rank's shard (tests / tools)"""
        self = cls.__new__(cls)
        self.group, self.world, self.rank = None, world, rank
        self.key_route = "allgather"
        self._setup(*a, **kw)
        return self


    # -- multi-hot "concat" (Combiner::Concat, R/HugeCTR/embedding/operators/generic_lookup.cuh:
    #    511-1084 one_to_one_*; CPU reference reference_embedding.hpp:125-139): the output of such a
    #    lookup is max_hotness vectors side by side, key r of the bucket in slot r, missing keys
    #    zero.  That is exactly max_hotness one-key "sum" lookups on the same table, lookup (l, r)
    #    holding key r of every bucket -- so the collection runs on that expanded lookup list (same
    #    routing, pooling, all-to-all, update kernels) and the CSR is re-cut on the way in.  In the
    #    batch-major output the r-th slot of lookup l is column block ev_offset(l) + r * ev, the
    #    reference's layout.
    def _expand_concat_lookups(self, config, hotness, batch_major):
        self._virt = None
        names = [str(c).lower().split(".")[-1] for _, _, _, c in config.lookups]
        if hotness is None or not any(c == "concat" and h > 1 for c, h in zip(names, hotness)):
            self.L_user = len(config.lookups)
            self.virt_span = [(l, 1) for l in range(self.L_user)]
            return config
        if not batch_major:
            raise _lib.HugeCTRAmdError("multi-hot 'concat' lookups need the batch-major output "
                                       "(the layout of EmbeddingCollectionConfig's list form)")
        exp = EmbeddingCollectionConfig()
        exp.__dict__.update({k: v for k, v in config.__dict__.items() if k != "lookups"})
        exp.lookups = []
        self._virt, self.virt_span = [], []
        for l, ((t, bottom, top, c), h) in enumerate(zip(config.lookups, hotness)):
            reps = int(h) if names[l] == "concat" else 1
            self.virt_span.append((len(exp.lookups), reps))
            for r in range(reps):
                exp.lookups.append((t, bottom, top, "sum" if reps > 1 else c))
                self._virt.append((l, r, reps > 1))
        self.L_user = len(config.lookups)
        return exp

    def _expand_csr(self, keys: torch.Tensor, bucket_range: torch.Tensor, batch: int):
        """feature-major CSR over the USER's lookups (bucket = lookup * batch + b) -> the same over
        the expanded lookups: lookup (l, r) of a multi-hot concat lookup gets key r of each bucket"""
        if self._virt is None:
            return keys, bucket_range
        br = bucket_range.to(torch.int64)
        lens = (br[1:] - br[:-1]).view(self.L_user, batch)
        starts = br[:-1].view(self.L_user, batch)
        ks, ls = [], []
        # key range of every user lookup: ONE host read per step (not two per lookup)
        bounds = br[0:self.L_user * batch + 1:batch].tolist() \
            if any(not split for _, _, split in self._virt) else None
        for l, r, split in self._virt:
            if not split:
                ks.append(keys[bounds[l]:bounds[l + 1]])
                ls.append(lens[l])
            else:
                m = lens[l] > r
                ks.append(keys[starts[l][m] + r])
                ls.append(m.to(torch.int64))
        nbr = torch.zeros(len(self._virt) * batch + 1, dtype=torch.int64, device=keys.device)
        torch.cumsum(torch.cat(ls), 0, out=nbr[1:])
        return torch.cat(ks), nbr.to(bucket_range.dtype)

    def _setup(self, config, global_batch, lr=0.01, optimizer=_lib.OPT_SGD, scaler=1.0,
               epsilon=1e-7, initial_accu_value=0.0, out_dtype=torch.float32, batch_major=False,
               key_dtype=torch.int64, max_hotness=1, seed=0, ftrl=(0.0, 0.0, 0.0),
               storage=None, initializer="", init_capacity=1 << 20, beta1=0.9, beta2=0.999,
               momentum_factor=0.9, rmsprop_beta=0.9, hotness=None, tables_from=None):
        # tables_from: another collection of the same config (the training one) whose tables and
        # optimizer state this one uses -- the evaluation runtime of a model owns per-batch
        # scratch only (a second copy of 100+ GB tables would not even fit for a moment)
        assert global_batch % self.world == 0
        # CompressionStrategy.Unique selects the reference's unique-compressed model-parallel
        # operator (distinct rows travel, the receiver pools;
        # R/HugeCTR/embedding/dense_model_parallel_embedding.cpp:1-279).  This runtime has ONE
        # model-parallel operator, Reduction (the owner pools, pooled vectors travel).  On one GPU
        # nothing travels and the two give the same output, so the request is honoured as is; on
        # several GPUs it is refused rather than silently run as Reduction.
        uniq = sorted(n for n, k in getattr(config, "compression", {}).items() if k == "unique")
        if uniq and self.world > 1:
            raise _lib.HugeCTRAmdError(
                "EmbeddingCollectionConfig.shard: CompressionStrategy.Unique for tables "
                f"{uniq} on {self.world} GPUs is not available (model-parallel lookups run with "
                "CompressionStrategy.Reduction only)")
        config = self._expand_concat_lookups(config, hotness, batch_major)
        if storage is None:  # max_vocabulary_size < 0 means dynamic (embedding_storage/common.hpp:78,
            # embedding_table.cpp:27-34: one dynamic table makes the whole group dynamic)
            storage = "dynamic" if any(t.max_vocabulary_size < 0 for t, _, _, _ in config.lookups) \
                else "static"
        assert storage in ("static", "dynamic")
        self.dynamic = storage == "dynamic"
        self.training = True
        self.dev = torch.device("cuda", torch.cuda.current_device())
        self.B, self.bpg = global_batch, global_batch // self.world
        self.lr, self.optimizer, self.scaler, self.epsilon = lr, optimizer, scaler, epsilon
        self.beta1, self.beta2, self.momentum_factor = beta1, beta2, momentum_factor
        self.out_dtype, self.batch_major, self.key_dtype = out_dtype, batch_major, key_dtype
        if not self.dynamic and optimizer not in (_lib.OPT_SGD, _lib.OPT_ADAGRAD, _lib.OPT_FTRL):
            # static EBC tables: SGD / AdaGrad / Ftrl only
            raise _lib.HugeCTRAmdError("EBC static tables support SGD, AdaGrad and Ftrl")
        self.ftrl = tuple(float(x) for x in ftrl)  # (lambda1, lambda2, beta)
        tables: List[EmbeddingTableConfig] = []
        for t, _, _, _ in config.lookups:
            if t not in tables:
                tables.append(t)
        self.tables = tables
        self.ev = tables[0].ev_size
        assert all(t.ev_size == self.ev for t in tables), "one ev_size per collection"
        self.L = len(config.lookups)
        self.lookup_table = [tables.index(t) for t, _, _, _ in config.lookups]
        # "concat" keeps every key's vector; with one key per bucket (the one-hot configurations that
        # use it, R/test/embedding_collection_test/dgx_a100_one_hot.py:287) it equals "sum";
        # multi-hot concat lookups were split into one-key lookups by _expand_concat_lookups
        names = [str(c).lower().split(".")[-1] for _, _, _, c in config.lookups]
        self.concat_lookups = [l for l, c in enumerate(names) if c == "concat"]
        self.combiner = [0 if c in ("sum", "0", "concat") else 1 for c in names]
        sm = config.ownership(tables, self.world)
        # owners of every table, ascending GPU order (shard id = position in that list)
        self.owners = [[g for g in range(self.world) if sm[g][t]] for t in range(len(tables))]
        assert all(self.owners), "every table needs at least one owner"
        # local flat table: shards of the tables this rank owns, ceil(vocab / num_shards) rows each
        self.local_tables = [t for t in range(len(tables)) if self.rank in self.owners[t]]
        self.row_start_of_table = {}
        rows = 0
        for t in self.local_tables:
            self.row_start_of_table[t] = rows
            ns = len(self.owners[t])
            rows += -(-max(tables[t].max_vocabulary_size, 0) // ns)
        self.local_rows = max(rows, 1)
        self.accum = self.ftrl_z = self.table = None
        if tables_from is not None:
            src = tables_from
            assert src.dynamic == self.dynamic and src.local_rows == self.local_rows
            self.table, self.accum, self.ftrl_z = src.table, src.accum, src.ftrl_z
            if self.dynamic:
                self.class_of_table = src.class_of_table
                self.det, self.det_opt = src.det, src.det_opt
                self.local_rows = 1
        elif self.dynamic:
            # one class of the dynamic table per local table shard; max_vocabulary_size is only a
            # hint here, the maps grow on demand (dynamic_embedding.cu:41-75)
            from .dynamic_table import DynamicEmbeddingTable, DynamicTableOptimizer
            self.class_of_table = {t: c for c, t in enumerate(self.local_tables)}
            ncls = max(len(self.local_tables), 1)
            self.det = DynamicEmbeddingTable([self.ev] * ncls, initializer, init_capacity,
                                             torch.int64, seed=seed * 1000003 + self.rank)
            # (a step that runs on the flat row store keeps its state there: no state table)
            flat_step = (os.environ.get("HCTR_DYNAMIC_FLAT", "1") != "0" and optimizer in _FLAT_STEP)
            self.det_opt = DynamicTableOptimizer(
                self.det, optimizer, lr, beta1, beta2, epsilon, momentum_factor, rmsprop_beta,
                self.ftrl[0], self.ftrl[1], self.ftrl[2], scaler, init_capacity,
                with_states=not flat_step)
            if flat_step and optimizer != _lib.OPT_SGD:
                # (allocated now, not inside the first step: set-up inside a step was not free
                #  with hundreds of GB resident)
                self.det.state_store(2 if optimizer == _lib.OPT_ADAM else 1)
            self.local_rows = 1
        else:
            self.table = torch.empty((self.local_rows, self.ev), dtype=torch.float32,
                                     device=self.dev)
            g = torch.Generator(device=self.dev)
            g.manual_seed(seed * 1000003 + self.rank)
            for t in self.local_tables:  # U(+-sqrt(1/vocab)) per table (ragged_static_embedding.cu:499-509)
                ns = len(self.owners[t])
                n = -(-tables[t].max_vocabulary_size // ns)
                b = (1.0 / tables[t].max_vocabulary_size) ** 0.5
                s0 = self.row_start_of_table[t]
                self.table[s0:s0 + n].uniform_(-b, b, generator=g)
            if optimizer == _lib.OPT_ADAGRAD:
                self.accum = torch.full_like(self.table, initial_accu_value)
            if optimizer == _lib.OPT_FTRL:  # n and z start at zero (ragged_static_embedding.cu:484-496)
                self.accum = torch.zeros_like(self.table)
                self.ftrl_z = torch.zeros_like(self.table)
        # lookups resolved on this rank (ascending global lookup id) and their descriptors
        self.local_lookups = [l for l in range(self.L) if self.rank in self.owners[self.lookup_table[l]]]
        desc, rs = [], []
        for l in self.local_lookups:
            t = self.lookup_table[l]
            desc += [l, len(self.owners[t]), self.owners[t].index(self.rank)]
            rs.append(-1 if self.dynamic else self.row_start_of_table[t])  # < 0: keep the key
        self.n_local = len(self.local_lookups)
        if self.dynamic:  # class of every (peer, local lookup) segment of the routed keys
            self.seg_class = [self.class_of_table[self.lookup_table[l]]
                              for _ in range(self.world) for l in self.local_lookups]
        self.d_desc = torch.tensor(desc or [0, 1, 0], dtype=torch.int32, device=self.dev)
        self.d_row_start = torch.tensor(rs or [0], dtype=torch.int64, device=self.dev)
        # receive-side block table: blocks are ordered [source rank][its local lookups]
        n_local_of = []
        for r in range(self.world):
            n_local_of.append([l for l in range(self.L) if r in self.owners[self.lookup_table[l]]])
        self.n_local_of = [len(x) for x in n_local_of]
        # key-route all-to-all: what every destination GPU resolves (its lookups, its shard ids)
        self._dest_desc = []
        for r in range(self.world):
            d = []
            for l in n_local_of[r]:
                t = self.lookup_table[l]
                d += [l, len(self.owners[t]), self.owners[t].index(r)]
            self._dest_desc.append(torch.tensor(d or [0, 1, 0], dtype=torch.int32, device=self.dev))
        self._neg = torch.full((max(self.L, 1),), -1, dtype=torch.int64, device=self.dev)
        self.max_shards = max(len(o) for o in self.owners)
        blk_base, acc = [], 0
        for r in range(self.world):
            blk_base.append(acc)
            acc += len(n_local_of[r])
        self.total_blocks = acc
        src = [-1] * (self.L * self.max_shards)
        for l in range(self.L):
            for s, r in enumerate(self.owners[self.lookup_table[l]]):
                src[l * self.max_shards + s] = blk_base[r] + n_local_of[r].index(l)
        self.d_src_blocks = torch.tensor(src, dtype=torch.int32, device=self.dev)
        self.d_combiner = torch.tensor(self.combiner, dtype=torch.int32, device=self.dev)
        self.send_counts = [self.n_local * self.bpg * self.ev] * self.world
        self.recv_counts = [n * self.bpg * self.ev for n in self.n_local_of]
        # scratch
        # key capacity of one batch: the declared hotness of every lookup when the caller knows it
        # (a 100-hot table beside one-hot ones must not size all 26 lookups for 100 keys)
        self.max_nnz = self.B * (max(1, sum(int(h) for h in hotness)) if hotness
                                 else max(1, self.L) * max(1, max_hotness))
        nb = self.world * max(self.n_local, 0) * self.bpg
        self.nb = nb
        self.ws = torch.empty(lib.hctr_ebc_route_workspace_bytes(self.B, max(self.n_local, 1)) + 64,
                              dtype=torch.uint8, device=self.dev)
        self.out_range = torch.zeros(nb + 1, dtype=torch.int64, device=self.dev)
        self.indices = torch.empty(self.max_nnz, dtype=torch.int64, device=self.dev)
        self.d_nnz = torch.zeros(1, dtype=torch.int64, device=self.dev)
        self.counts = torch.zeros(self.L * self.bpg, dtype=torch.int64, device=self.dev)
        self._upd = ctypes.c_void_p()
        # (dynamic: the table grows -- every update names the rows handed out so far,
        #  hctr_updater_set_row_bound; the unique-row reduce numbers its rows < max_nnz)
        check(lib.hctr_updater_create(self.max_nnz, 0xFFFFFFEF if self.dynamic else self.local_rows,
                                      self.ev, ctypes.byref(self._upd)))
        # HCTR_DYNAMIC_FLAT=0: the pointer-per-key lookup and the unique-key optimizer step for
        # every optimizer (what Nesterov / RMSProp / Ftrl take in any case)
        self._dyn_flat = self.dynamic and os.environ.get("HCTR_DYNAMIC_FLAT", "1") != "0"
        if optimizer == _lib.OPT_FTRL and not self.dynamic:
            check(lib.hctr_updater_set_ftrl(self._upd, *self.ftrl))
        self._times = 0
        self._dyn_times = 0
        self._nnz_host = 0
        # One GPU, no Average lookup (static or dynamic tables): the send layout [peer][lookup][b][ev] IS the
        # feature-major output and there is nothing to exchange, so the two reorder passes around
        # the all-to-all (network_forward / network_backward) drop out; the batch-major output is
        # the same buckets stored (and their gradients read) through a transposed address
        # (hctr_forward_pool_mapped, hctr_updater_set_grad_map).  HCTR_EBC_DIRECT=0 keeps the
        # staged path (tests compare the two).
        self._multi_hot = max_hotness > 1
        self._direct = (self.world == 1 and self.n_local == self.L
                        and os.environ.get("HCTR_EBC_DIRECT", "1") != "0")
        # Average lookups on the direct path: the gather stores SUMS, the receiver arithmetic of
        # network_forward / network_backward is applied in place to the Average
        # lookups' vectors (hctr_ebc_scale_average)
        self._direct_avg = self._direct and any(c == 1 for c in self.combiner)
        self._map_on = False
        self.d_one_hot = torch.zeros(1, dtype=torch.int32, device=self.dev)

    def __del__(self):
        u = getattr(self, "_upd", None)
        if u is not None and u.value:
            lib.hctr_updater_destroy(u)
            self._upd = ctypes.c_void_p()

    # -- collectives (identity for world == 1; the single-process tests drive them by hand) -------
    def _allgather_keys(self, keys, bucket_range):
        """data-parallel CSR of every rank -> global feature-major CSR (keys, bucket_range[L*B+1]).
        Key counts are exchanged first and read on the host (one sync), as the reference's key
        routing does (sparse_data_distribution_op_impl.cu:303-307)."""
        if self.world == 1 or not dist.is_initialized():
            return keys, bucket_range
        lens = (bucket_range[1:] - bucket_range[:-1]).to(torch.int64).contiguous()
        staged = dist.get_backend(self.group) == "gloo"  # host staging: tests only
        cdev = torch.device("cpu") if staged else self.dev
        all_lens = torch.empty(self.world * lens.numel(), dtype=torch.int64, device=cdev)
        dist.all_gather_into_tensor(all_lens, lens.to(cdev), group=self.group)
        all_lens = all_lens.to(self.dev).view(self.world, self.L, self.bpg)
        n_all = all_lens.sum(dim=(1, 2)).tolist()  # host sync (counts)
        # ranks hold different numbers of keys: gather buffers padded to the longest share
        m = max(max(n_all), 1)
        mine = torch.zeros(m, dtype=keys.dtype, device=cdev)
        mine[:keys.numel()] = keys.to(cdev)
        gathered = torch.empty(self.world * m, dtype=keys.dtype, device=cdev)
        dist.all_gather_into_tensor(gathered, mine, group=self.group)
        gathered = gathered.to(self.dev)
        parts = [gathered[r * m:r * m + n_all[r]] for r in range(self.world)]
        # per-rank offsets of each lookup's key segment
        seg = all_lens.sum(dim=2)                       # [world, L] keys per (rank, lookup)
        seg_off = torch.zeros((self.world, self.L + 1), dtype=torch.int64, device=self.dev)
        torch.cumsum(seg, 1, out=seg_off[:, 1:])
        seg_off = seg_off.tolist()
        out = [parts[r][seg_off[r][l]:seg_off[r][l + 1]]
               for l in range(self.L) for r in range(self.world)]
        glens = all_lens.permute(1, 0, 2).reshape(-1)   # [l][rank][b_local] == [l][b]
        gbr = torch.zeros(self.L * self.B + 1, dtype=bucket_range.dtype, device=self.dev)
        torch.cumsum(glens, 0, out=gbr[1:])
        return torch.cat(out), gbr

    def _a2a_async(self, buf, send_counts, recv_counts):
        """-> (receive buffer, work or None): the collective runs on the communicator's stream while
        the caller keeps launching; work.wait() orders the current stream behind it"""
        if self.world == 1 or not dist.is_initialized():
            return buf, None
        out = torch.empty(sum(recv_counts), dtype=buf.dtype, device=self.dev)
        from .parallel import all_to_all_single
        work = all_to_all_single(out, buf.reshape(-1), recv_counts, send_counts, group=self.group,
                                 async_op=True)
        return out, work

    # -- the model-parallel exchange in two halves (hugectr.Model, train_intra_iteration_overlap:
    #    ebc_mp_model_forward / ebc_mp_network_forward on their own stream while the bottom MLP
    #    runs, R/HugeCTR/src/pybind/model_pipeline.cpp:299-346) -----------------------------------
    def forward_global_begin(self, gkeys: torch.Tensor, gbucket_range: torch.Tensor):
        send = self.route_and_pool(gkeys, gbucket_range, False)
        recv, work = self._a2a_async(send, self.send_counts, self.recv_counts)
        return recv, work, send  # (send stays alive until the collective has read it)

    def forward_global_finish(self, recv: torch.Tensor, work) -> torch.Tensor:
        if work is not None:
            work.wait()
        return self.network_forward(recv)

    def backward_begin(self, grad: torch.Tensor):
        send = self.network_backward(grad)
        top, work = self._a2a_async(send, self.recv_counts, self.send_counts)
        return top, work, send

    def backward_finish(self, top: torch.Tensor, work):
        if work is not None:
            work.wait()
        self.apply_gradients(top)

    def _a2a(self, buf, send_counts, recv_counts):
        if self.world == 1 or not dist.is_initialized():
            return buf
        out = torch.empty(sum(recv_counts), dtype=buf.dtype, device=self.dev)
        # (parallel.all_to_all_single stages through the host under gloo: the 2-rank tests)
        from .parallel import all_to_all_single
        all_to_all_single(out, buf.reshape(-1), recv_counts, send_counts, group=self.group)
        return out

    # -- stages (public so that tests can emulate the collectives in one process) ------------------
    def route_and_pool(self, gkeys: torch.Tensor, gbucket_range: torch.Tensor,
                       direct: bool = False) -> torch.Tensor:
        """global CSR -> pooled partial vectors in the all-to-all send layout
        [peer][local lookup][b_local][ev] (direct: straight into the output, see _setup)"""
        if self._virt is not None and gbucket_range.numel() == self.L_user * self.B + 1:
            gkeys, gbucket_range = self._expand_csr(gkeys, gbucket_range, self.B)
        kt = _lib.KEY_I64 if gkeys.dtype == torch.int64 else _lib.KEY_U32
        if not direct or self._direct_avg:  # bucket lengths of my samples: the Average divisor
            check(lib.hctr_ebc_bucket_counts(self.B, self.world, self.rank, self.L,
                                             ptr(gbucket_range), kt, ptr(self.counts),
                                             stream_ptr()))
        if direct:
            shape = (self.bpg, self.L, self.ev) if self.batch_major else (self.L, self.bpg, self.ev)
            send = torch.empty(shape, dtype=self.out_dtype, device=self.dev)
        else:
            send = torch.empty((max(self.nb, 1), self.ev), dtype=self.out_dtype, device=self.dev)
        if self.n_local == 0:
            return send[:0]
        if direct:
            # every lookup whole on this rank: one pass (hctr_ebc_route_whole), and the gather keys
            # its offset-free loop on the one-hot flag that pass leaves on the device
            check(lib.hctr_ebc_route_whole(self.B, self.L, ptr(self.d_row_start), ptr(gkeys),
                                           ptr(gbucket_range), kt, ptr(self.out_range),
                                           ptr(self.indices), ptr(self.d_nnz), ptr(self.d_one_hot),
                                           int(gkeys.numel()), stream_ptr()))
            if self.dynamic:  # (row_start < 0: the pass kept the raw keys)
                return self._scale_average(self._dynamic_pool(send, True), True)
            self._nnz_host = int(gkeys.numel())
            bm = self.batch_major
            check(lib.hctr_forward_pool_mapped(self.nb, self.ev, 0, ptr(self.out_range),
                                               _lib.KEY_I64, ptr(self.indices), ptr(self.table),
                                               ptr(send), _DT[self.out_dtype],
                                               1 if self._multi_hot else 0, self.bpg if bm else 0,
                                               self.L if bm else 0, ptr(self.d_one_hot),
                                               stream_ptr()))
            return self._scale_average(send, True)
        check(lib.hctr_ebc_route_keys(self.B, self.world, self.n_local, ptr(self.d_desc),
                                      ptr(self.d_row_start), ptr(gkeys), ptr(gbucket_range), kt,
                                      ptr(self.out_range), ptr(self.indices), ptr(self.d_nnz),
                                      ptr(self.ws), stream_ptr()))
        if self.dynamic:
            return self._dynamic_pool(send)
        self._nnz_host = int(gkeys.numel())  # upper bound; the live count stays on the device
        check(lib.hctr_forward_pool(self.nb, self.ev, 0, ptr(self.out_range), _lib.KEY_I64,
                                    ptr(self.indices), ptr(self.table), ptr(send),
                                    _DT[self.out_dtype], stream_ptr()))
        return send

    def _scale_average(self, data: torch.Tensor, forward: bool) -> torch.Tensor:
        if self._direct_avg and data.numel():
            check(lib.hctr_ebc_scale_average(self.bpg, self.L, self.ev, ptr(self.d_combiner),
                                             ptr(self.counts), 1 if self.batch_major else 0,
                                             ptr(data), _DT[self.out_dtype], 1 if forward else 0,
                                             stream_ptr()))
        return data

    def forward_global(self, gkeys: torch.Tensor, gbucket_range: torch.Tensor) -> torch.Tensor:
        """replicated global feature-major CSR -> this rank's output rows"""
        send = self.route_and_pool(gkeys, gbucket_range, self._direct)
        if self._direct:
            return send
        recv = self._a2a(send, self.send_counts, self.recv_counts)
        return self.network_forward(recv)

    # -- the reference's key route for data-parallel input: two all-to-alls -------------------------
    def route_send(self, keys: torch.Tensor, bucket_range: torch.Tensor):
        """this rank's share of the batch (feature-major, bucket = lookup * batch/world + b_local)
        -> per destination GPU: (lengths of its buckets [n_local_of[p] * batch/world], its keys)"""
        kt = _lib.KEY_I64 if keys.dtype == torch.int64 else _lib.KEY_U32
        br = bucket_range
        # total keys of my buckets: the Average divisor on the receiving side (network_forward)
        self.counts.copy_((br[1:] - br[:-1]).to(torch.int64))
        if getattr(self, "_ws_local", None) is None:
            self._ws_local = torch.empty(lib.hctr_ebc_route_workspace_bytes(self.bpg, max(self.L, 1))
                                         + 64, dtype=torch.uint8, device=self.dev)
        lens, out = [], []
        for p in range(self.world):
            nl = self.n_local_of[p]
            rng = torch.zeros(nl * self.bpg + 1, dtype=torch.int64, device=self.dev)
            ks = torch.empty(max(keys.numel(), 1), dtype=torch.int64, device=self.dev)
            if nl:
                check(lib.hctr_ebc_route_keys(self.bpg, 1, nl, ptr(self._dest_desc[p]), ptr(self._neg),
                                              ptr(keys), ptr(br), kt, ptr(rng), ptr(ks), None,
                                              ptr(self._ws_local), stream_ptr()))
            lens.append(rng[1:] - rng[:-1])
            out.append((ks, rng))
        ends = torch.stack([r[-1] for _, r in out]).tolist()  # host sync (key counts, as the reference)
        return lens, [ks[:n] for (ks, _), n in zip(out, ends)]

    def route_recv(self, lens_all: torch.Tensor, keys_all: torch.Tensor):
        """lengths / keys received from every source GPU, source-major = my bucket order
        [source][local lookup][b_local] -> the routed CSR the pooling and the update work on"""
        n = int(keys_all.numel())
        self._nnz_host = n
        self.out_range[0] = 0
        torch.cumsum(lens_all, 0, out=self.out_range[1:self.nb + 1])
        self.indices[:n] = keys_all
        self.d_nnz.fill_(n)
        check(lib.hctr_ebc_routed_keys_to_indices(self.bpg, self.world, self.n_local,
                                                  ptr(self.d_desc), ptr(self.d_row_start),
                                                  ptr(self.out_range), ptr(self.indices),
                                                  stream_ptr()))

    def pool_routed(self) -> torch.Tensor:
        send = torch.empty((max(self.nb, 1), self.ev), dtype=self.out_dtype, device=self.dev)
        if self.n_local == 0:
            return send[:0]
        if self.dynamic:
            return self._dynamic_pool(send)
        check(lib.hctr_forward_pool(self.nb, self.ev, 0, ptr(self.out_range), _lib.KEY_I64,
                                    ptr(self.indices), ptr(self.table), ptr(send),
                                    _DT[self.out_dtype], stream_ptr()))
        return send

    def _forward_a2a_route(self, keys: torch.Tensor, bucket_range: torch.Tensor) -> torch.Tensor:
        from .parallel import all_to_all_single
        lens, ks = self.route_send(keys, bucket_range)
        send_l = torch.cat(lens)
        recv_l = torch.empty(self.world * self.n_local * self.bpg, dtype=torch.int64, device=self.dev)
        all_to_all_single(recv_l, send_l, [self.n_local * self.bpg] * self.world,
                          [n * self.bpg for n in self.n_local_of], group=self.group)
        recv_counts = recv_l.view(self.world, -1).sum(1).tolist()  # host sync (counts)
        recv_k = torch.empty(sum(recv_counts), dtype=torch.int64, device=self.dev)
        all_to_all_single(recv_k, torch.cat(ks), recv_counts, [int(k.numel()) for k in ks],
                          group=self.group)
        self.route_recv(recv_l, recv_k)
        return self.pool_routed()

    def _dynamic_pool(self, send: torch.Tensor, direct: bool = False) -> torch.Tensor:
        """self.indices holds the routed raw keys in [peer][local lookup][b_local] bucket order:
        one (peer, lookup) segment = one id space of the dynamic table.  The segment offsets are
        read on the host, like the id_space_offset of the reference's lookup
        (dynamic_embedding.cu:139-150)."""
        seg = self.out_range[0:self.nb + 1:self.bpg].tolist()  # host sync (offsets)
        nnz = seg[-1]
        self._nnz_host = nnz
        if nnz == 0:
            return send.zero_()
        keys = self.indices[:nnz]
        # one ev_size per group: the classes' rows are ONE flat table and the row numbers index it
        # (hctr_det_row_store) -- the static tables' gather runs on them, no pointer per key
        flat = self._dyn_flat
        ptrs, rows, base = self.det.lookup_rows(keys, self.seg_class, seg, insert=self.training,
                                                want_ptrs=not flat)
        self._dyn_rows, self._dyn_base = rows, base
        bm = direct and self.batch_major  # one GPU: pooled straight into the batch-major output
        if flat:
            store, _ = self.det.row_store()
            if direct:
                check(lib.hctr_forward_pool_mapped(self.nb, self.ev, 0, ptr(self.out_range),
                                                   _lib.KEY_I64, ptr(rows), store, ptr(send),
                                                   _DT[self.out_dtype], 1 if self._multi_hot else 0,
                                                   self.bpg if bm else 0, self.L if bm else 0,
                                                   ptr(self.d_one_hot), stream_ptr()))
            else:
                check(lib.hctr_forward_pool(self.nb, self.ev, 0, ptr(self.out_range), _lib.KEY_I64,
                                            ptr(rows), store, ptr(send), _DT[self.out_dtype],
                                            stream_ptr()))
            return send
        check(lib.hctr_forward_pool_ptrs_mapped(self.nb, self.ev, 0, ptr(self.out_range), ptr(ptrs),
                                                ptr(send), _DT[self.out_dtype],
                                                self.bpg if bm else 0, self.L if bm else 0,
                                                stream_ptr()))
        return send

    def _dynamic_apply(self, top_grad: torch.Tensor):
        nnz = self._nnz_host
        if nnz == 0:
            return
        if self._dyn_flat and self.optimizer in _FLAT_STEP:
            # sort by row, sum per row in bucket order and apply -- *_update_grad_kernel +
            # scatter_add (dynamic_embedding.cu:227-330, optimizers.cuh:29-140) in the segmented
            # reduce of the static tables' update, on the flat row store: no unique-key list, no
            # wgrad buffer, no second probe of the keys -- and, AdaGrad / Adam / MomentumSGD, no
            # third one either: the state lies at the weights' row numbers (hctr_det_state_store;
            # the reference probes a second table keyed like the first).  Same bits as the
            # unique-key flow below: the same sums in the same order through the same formulas.
            # (Nesterov / Ftrl are written differently there -- delta first, then w += delta -- and
            # RMSProp is not a static optimizer at all: they keep the unique-key flow)
            store, total = self.det.row_store()
            s0 = s1 = None
            if self.optimizer != _lib.OPT_SGD:
                s0, s1 = self.det.state_store(2 if self.optimizer == _lib.OPT_ADAM else 1)
            # (Adam's step count: updates that met keys -- `++adam.times` sits inside the
            #  reference's `if (num_unique_keys_cpu > 0)`, dynamic_embedding.cu:187, 239)
            self._dyn_times += 1
            check(lib.hctr_updater_set_row_bound(self._upd, total))
            check(lib.hctr_updater_update(self._upd, self.nb, nnz, ptr(self.out_range),
                                          ptr(self._dyn_rows), ptr(top_grad), _DT[self.out_dtype],
                                          self.optimizer, _lib.UPDATE_LOCAL, self.lr, self.beta1,
                                          self.beta2, self.epsilon, self.momentum_factor,
                                          self.scaler, self._dyn_times, store, s0, s1,
                                          stream_ptr()))
            return
        urow = torch.empty(nnz, dtype=torch.int64, device=self.dev)
        ukey = torch.empty(nnz, dtype=torch.int64, device=self.dev)
        wgrad = torch.empty((nnz, self.ev), dtype=torch.float32, device=self.dev)
        nu = ctypes.c_size_t()
        check(lib.hctr_ebc_local_reduce(self._upd, self.nb, nnz, ptr(self.out_range),
                                        ptr(self._dyn_rows), self._dyn_base[-1], ptr(self.indices),
                                        ptr(top_grad), _DT[self.out_dtype], ctypes.byref(nu),
                                        ptr(urow), ptr(ukey), ptr(wgrad), stream_ptr()))
        n = nu.value
        # unique rows ascend, so the classes are contiguous: their borders in the unique list
        ncls = len(self.det.dims)
        base = torch.tensor(self._dyn_base, dtype=torch.int64, device=self.dev)
        off = torch.searchsorted(urow[:n], base).tolist()  # host sync (table ranges)
        ev_start = torch.arange(0, n * self.ev, self.ev, dtype=torch.int32, device=self.dev)
        self.det_opt.set_learning_rate(self.lr)
        self.det_opt.update(ukey[:n], ev_start, wgrad[:n].view(-1), list(range(ncls)), off)

    def network_forward(self, recv: torch.Tensor) -> torch.Tensor:
        shape = (self.bpg, self.L, self.ev) if self.batch_major else (self.L, self.bpg, self.ev)
        out = torch.empty(shape, dtype=self.out_dtype, device=self.dev)
        check(lib.hctr_ebc_network_forward(self.bpg, self.L, self.ev, self.max_shards,
                                           ptr(self.d_src_blocks), ptr(self.d_combiner),
                                           ptr(self.counts), 1 if self.batch_major else 0,
                                           ptr(recv), ptr(out), _DT[self.out_dtype], stream_ptr()))
        return out

    def network_backward(self, grad: torch.Tensor) -> torch.Tensor:
        # every (lookup, shard) block is written exactly once by the kernel: no zero fill needed
        send = torch.empty((max(self.total_blocks, 1) * self.bpg, self.ev), dtype=self.out_dtype,
                           device=self.dev)
        grad = grad.contiguous()  # (named: the pointer must outlive the argument list)
        check(lib.hctr_ebc_network_backward(self.bpg, self.L, self.ev, self.max_shards,
                                            ptr(self.d_src_blocks), ptr(self.d_combiner),
                                            ptr(self.counts), 1 if self.batch_major else 0,
                                            ptr(grad), ptr(send), _DT[self.out_dtype],
                                            stream_ptr()))
        return send

    def apply_gradients(self, top_grad: torch.Tensor, direct: bool = False):
        """top_grad: [peer][local lookup][b_local][ev] gradients of my pooled partial vectors
        (direct: the gradient of the output itself, in the output's layout)"""
        if self.n_local == 0:
            return
        self._times += 1
        mapped = direct and self.batch_major
        if mapped != self._map_on:
            check(lib.hctr_updater_set_grad_map(self._upd, self.bpg if mapped else 0,
                                                self.L if mapped else 0))
            self._map_on = mapped
        top_grad = top_grad.contiguous()
        if self.dynamic:
            return self._dynamic_apply(top_grad)
        check(lib.hctr_updater_update(self._upd, self.nb, self._nnz_host, ptr(self.out_range),
                                      ptr(self.indices), ptr(top_grad),
                                      _DT[self.out_dtype], self.optimizer, _lib.UPDATE_LOCAL,
                                      self.lr, 0.9, 0.999, self.epsilon, 0.0, self.scaler,
                                      self._times, ptr(self.table), ptr(self.accum),
                                      ptr(self.ftrl_z), stream_ptr()))

    # -- whole passes --------------------------------------------------------------------------------
    def forward(self, keys: torch.Tensor, bucket_range: torch.Tensor) -> torch.Tensor:
        if self._virt is not None:  # this rank's share of the batch, user lookups -> expanded
            batch = (bucket_range.numel() - 1) // self.L_user
            keys, bucket_range = self._expand_csr(keys, bucket_range, batch)
        if self.key_route == "a2a" and self.world > 1 and dist.is_initialized():
            send = self._forward_a2a_route(keys, bucket_range)
        else:
            gk, gbr = self._allgather_keys(keys, bucket_range)
            send = self.route_and_pool(gk, gbr, self._direct)
        if self._direct:
            return send
        recv = self._a2a(send, self.send_counts, self.recv_counts)
        return self.network_forward(recv)

    def backward_and_update(self, grad: torch.Tensor):
        if self._direct:  # the gradient of the output is the gradient of my buckets
            if self._direct_avg:  # scaled in a copy: the caller's gradient tensor stays as it was
                grad = self._scale_average(grad.contiguous().clone(), False)
            return self.apply_gradients(grad, True)
        send = self.network_backward(grad)
        top = self._a2a(send, self.recv_counts, self.send_counts)
        self.apply_gradients(top)


class DataParallelCollection:
    """Replicated ("dp") tables of an embedding_collection
    (R/HugeCTR/embedding/data_parallel_embedding.cpp; shard_strategy ("dp", [...]) of
    EmbeddingCollectionConfig.shard): every GPU holds the whole table -- identically initialised,
    SURVEY q14 -- and resolves the lookups of ITS OWN samples, so nothing of these lookups crosses
    the all-to-all; the replicas stay in step through one all-reduce of the (dense) per-row gradient
    sums (communication.cpp:145-157), after which every GPU applies the same sparse optimizer
    step to the rows any GPU touched.  Small tables only: the all-reduce moves rows x ev floats.

    forward(gkeys, gbucket_range): the replicated global batch, feature-major over the collection's
    own lookups (bucket = lookup * batch + b) -> [batch/world, lookups, ev]."""

    def __init__(self, config: EmbeddingCollectionConfig, global_batch: int, lr: float = 0.01,
                 optimizer: int = _lib.OPT_SGD, scaler: float = 1.0, epsilon: float = 1e-7,
                 initial_accu_value: float = 0.0, out_dtype=torch.float32, max_hotness: int = 1,
                 seed: int = 0, group=None, ftrl=(0.0, 0.0, 0.0), rank=None, world=None):
        self.group = group
        self.world = world if world is not None else (
            dist.get_world_size(group) if dist.is_initialized() else 1)
        self.rank = rank if rank is not None else (dist.get_rank(group) if dist.is_initialized() else 0)
        assert global_batch % self.world == 0
        if optimizer not in (_lib.OPT_SGD, _lib.OPT_ADAGRAD, _lib.OPT_FTRL):
            raise _lib.HugeCTRAmdError("EBC static tables support SGD, AdaGrad and Ftrl")
        self.dev = torch.device("cuda", torch.cuda.current_device())
        self.B, self.bpg = global_batch, global_batch // self.world
        self.lr, self.optimizer, self.scaler, self.epsilon = lr, optimizer, scaler, epsilon
        self.out_dtype, self.ftrl = out_dtype, tuple(float(x) for x in ftrl)
        tables: List[EmbeddingTableConfig] = []
        for t, _, _, _ in config.lookups:
            if t not in tables:
                tables.append(t)
        self.tables, self.ev = tables, tables[0].ev_size
        assert all(t.ev_size == self.ev for t in tables), "one ev_size per collection"
        self.L = len(config.lookups)
        names = [str(c).lower().split(".")[-1] for _, _, _, c in config.lookups]
        self.combiner = [0 if c in ("sum", "0", "concat") else 1 for c in names]
        starts, rows = [], 0
        for t in tables:
            starts.append(rows)
            rows += t.max_vocabulary_size
        self.rows = max(rows, 1)
        self.table = torch.empty((self.rows, self.ev), dtype=torch.float32, device=self.dev)
        g = torch.Generator(device=self.dev)
        g.manual_seed(seed * 1000003 + 99991)  # replica-uniform: the SAME stream on every rank
        for t, s0 in zip(tables, starts):
            b = (1.0 / t.max_vocabulary_size) ** 0.5
            self.table[s0:s0 + t.max_vocabulary_size].uniform_(-b, b, generator=g)
        self.accum = self.ftrl_z = None
        if optimizer == _lib.OPT_ADAGRAD:
            self.accum = torch.full_like(self.table, initial_accu_value)
        if optimizer == _lib.OPT_FTRL:
            self.accum, self.ftrl_z = torch.zeros_like(self.table), torch.zeros_like(self.table)
        desc, rs = [], []
        for l, (t, _, _, _) in enumerate(config.lookups):
            desc += [l, 1, 0]
            rs.append(starts[tables.index(t)])
        self.d_desc = torch.tensor(desc, dtype=torch.int32, device=self.dev)
        self.d_row_start = torch.tensor(rs, dtype=torch.int64, device=self.dev)
        self.mean_ids = [l for l in range(self.L) if self.combiner[l] == 1]
        self.max_nnz = self.B * self.L * max(1, max_hotness)
        nb = self.world * self.L * self.bpg
        self.ws = torch.empty(lib.hctr_ebc_route_workspace_bytes(self.B, self.L) + 64,
                              dtype=torch.uint8, device=self.dev)
        self.out_range = torch.zeros(nb + 1, dtype=torch.int64, device=self.dev)
        self.indices = torch.empty(self.max_nnz, dtype=torch.int64, device=self.dev)
        self._reduce = ctypes.c_void_p()   # per-row gradient sums of my samples
        check(lib.hctr_updater_create(self.max_nnz, self.rows, self.ev, ctypes.byref(self._reduce)))
        self._apply = ctypes.c_void_p()    # optimizer step on the touched rows
        check(lib.hctr_updater_create(self.rows, self.rows, self.ev, ctypes.byref(self._apply)))
        if optimizer == _lib.OPT_FTRL:
            check(lib.hctr_updater_set_ftrl(self._apply, *self.ftrl))
        self._times = 0

    def __del__(self):
        for name in ("_reduce", "_apply"):
            u = getattr(self, name, None)
            if u is not None and u.value:
                lib.hctr_updater_destroy(u)
                setattr(self, name, ctypes.c_void_p())

    def forward(self, gkeys: torch.Tensor, gbucket_range: torch.Tensor) -> torch.Tensor:
        kt = _lib.KEY_I64 if gkeys.dtype == torch.int64 else _lib.KEY_U32
        check(lib.hctr_ebc_route_keys(self.B, self.world, self.L, ptr(self.d_desc),
                                      ptr(self.d_row_start), ptr(gkeys), ptr(gbucket_range), kt,
                                      ptr(self.out_range), ptr(self.indices), None, ptr(self.ws),
                                      stream_ptr()))
        n = self.L * self.bpg
        # buckets are ordered [peer][lookup][b_local]: my own samples are the segment peer == rank
        self._seg = self.out_range[self.rank * n:(self.rank + 1) * n + 1]
        # Average divides the fp32 sum by the bucket's key count when > 0 and rounds ONCE to the
        # output type (multi_to_one_*_kernel, generic_lookup.cuh:336-348): with a 16-bit output and
        # an Average lookup the sums are pooled in fp32 (a Sum lookup's vector is the same either
        # way: the rounded fp32 sum)
        wide = bool(self.mean_ids) and self.out_dtype != torch.float32
        pdt = torch.float32 if wide else self.out_dtype
        pooled = torch.empty((n, self.ev), dtype=pdt, device=self.dev)
        check(lib.hctr_forward_pool(n, self.ev, 0, ptr(self._seg), _lib.KEY_I64, ptr(self.indices),
                                    ptr(self.table), ptr(pooled), _DT[pdt], stream_ptr()))
        pooled = pooled.view(self.L, self.bpg, self.ev)
        self._cnt = None
        if self.mean_ids:
            cnt = (self._seg[1:] - self._seg[:-1]).view(self.L, self.bpg).clamp(min=1)
            self._cnt = cnt[self.mean_ids].unsqueeze(-1).to(torch.float32)
            pooled[self.mean_ids] = pooled[self.mean_ids].float() / self._cnt
        return pooled.permute(1, 0, 2).contiguous().to(self.out_dtype)

    def backward_local(self, grad: torch.Tensor):
        """grad [batch/world, lookups, ev] -> (dense per-row gradient sums [rows, ev] fp32 of MY
        samples, touched-row flags [rows]) -- the operands of the all-reduce"""
        g = grad.permute(1, 0, 2).contiguous()
        if self.mean_ids:
            # AverageCombiner (data_parallel_embedding.cpp:226-243): the divided gradient stays
            # fp32 (float_emb_vec_) on its way into the local reduce -- it is not rounded back
            g = g.float()
            g[self.mean_ids] = g[self.mean_ids] / self._cnt
        n = self.L * self.bpg
        seg0 = int(self._seg[0])  # host sync: where my segment starts in the routed key list
        nnz = int(self._seg[-1]) - seg0
        dense = torch.zeros((self.rows, self.ev), dtype=torch.float32, device=self.dev)
        touched = torch.zeros(self.rows, dtype=torch.float32, device=self.dev)
        if nnz:
            rng = (self._seg - seg0).contiguous()
            rows = self.indices[seg0:seg0 + nnz]
            urow = torch.empty(nnz, dtype=torch.int64, device=self.dev)
            wg = torch.empty((nnz, self.ev), dtype=torch.float32, device=self.dev)
            nu = ctypes.c_size_t()
            check(lib.hctr_ebc_local_reduce(self._reduce, n, nnz, ptr(rng), ptr(rows), self.rows, None,
                                            ptr(g), _DT[g.dtype], ctypes.byref(nu), ptr(urow), None,
                                            ptr(wg), stream_ptr()))
            u = urow[:nu.value]
            dense[u] = wg[:nu.value]
            touched[u] = 1.0
        return dense, touched

    def apply_reduced(self, dense: torch.Tensor, touched: torch.Tensor):
        """the same optimizer step on every GPU: rows any GPU touched, summed gradients"""
        self._times += 1
        rows = torch.nonzero(touched > 0).view(-1)  # host sync (count)
        n = int(rows.numel())
        if n == 0:
            return
        ro = torch.arange(n + 1, dtype=torch.int64, device=self.dev)
        picked = dense[rows]  # (named: a temporary would be released before the launch reads it)
        check(lib.hctr_updater_update(self._apply, n, n, ptr(ro), ptr(rows), ptr(picked), _lib.F32,
                                      self.optimizer, _lib.UPDATE_LOCAL, self.lr, 0.9, 0.999,
                                      self.epsilon, 0.0, self.scaler, self._times, ptr(self.table),
                                      ptr(self.accum), ptr(self.ftrl_z), stream_ptr()))

    def backward_and_update(self, grad: torch.Tensor):
        dense, touched = self.backward_local(grad)
        if self.world > 1 and dist.is_initialized():
            from .parallel import all_reduce
            all_reduce(dense, group=self.group)      # communication.cpp:145-157
            all_reduce(touched, group=self.group)
        self.apply_reduced(dense, touched)


def static_lookup(indices: torch.Tensor, num_keys_per_table_offset: torch.Tensor,
                  table_id_list: torch.Tensor, local_table_ids: torch.Tensor,
                  table_index_start: torch.Tensor, emb_table: torch.Tensor,
                  table_ev_offset: torch.Tensor, local_ev_sizes: torch.Tensor):
    """embedding::ILookup::lookup for a static (ragged) table shard -- the SOK plug-in point
    (R/HugeCTR/embedding/embedding_table.hpp:22-33, ragged_static_embedding.cu:33-51): returns
    (ptrs, error_flags) where ptrs[i] (int64 view of float*) is the device address of position
    i's fp32 vector inside `emb_table`; feed ptrs to `hctr_forward_pool_ptrs`.  All arguments are
    device tensors: indices uint64-valued int64 (hctr_ebc_keys_to_indices numbering) or int32 /
    int64 keys, offsets uint32-valued int32, id lists int32, index / element offsets int64."""
    n = indices.numel()
    ptrs = torch.zeros(max(n, 1), dtype=torch.int64, device=indices.device)
    err = torch.zeros(1, dtype=torch.int32, device=indices.device)
    kt = 2 if indices.dtype == torch.int64 else 0
    check(lib.hctr_static_lookup(ptr(indices), kt, n, ptr(num_keys_per_table_offset),
                                 num_keys_per_table_offset.numel(), ptr(table_id_list),
                                 ptr(local_table_ids), local_table_ids.numel(),
                                 ptr(table_index_start), ptr(emb_table), ptr(table_ev_offset),
                                 ptr(local_ev_sizes), ptr(ptrs), ptr(err), stream_ptr()))
    return ptrs[:n], err
