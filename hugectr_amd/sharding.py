"""Embedding-table sharding plans for embedding_collection: which GPUs own (a row shard of) which
table.  Same inputs, options and results as the planner the reference's MLPerf DLRM-DCNv2 sample
uses (R/samples/dlrm/sharding/generate_plan.py:23-131, planner.py:23-327; called from
R/samples/dlrm/train.py:288-290): `round_robin`, `uniform`, and the cost-model driven `auto` /
`hier_auto` search -- greedy placement of the hottest shard into the cheapest bin, splitting the
hottest (or the non-fitting) table in two until nothing improves.  The reference implementation is
importable Python, so this module is pinned against its outputs (tests/golden/sharding_plans.json,
made by tests/golden/make_sharding_golden.py).

cost of a GPU = sum over its shards of hotness / #shards  +  (#tables on it) * table_cost, where
table_cost = mem_comm_bw_ratio / mem_comm_work_ratio prices the all-to-all share of one table
against one row read; `mi355x_args()` fills those ratios for MI355X (HBM3E vs xGMI).
"""
from __future__ import annotations

import logging
import time
from argparse import Namespace
from typing import List, Sequence, Tuple

import numpy as np


def mi355x_args(**overrides) -> Namespace:
    """planner options with MI355X numbers: 8 TB/s HBM3E against 7 xGMI links x 153 GB/s per GPU,
    the reference's 8 / 2 memory-to-communication work ratio, 240 of the 288 GB for embeddings"""
    a = dict(sharding_plan="auto", optimizer="sgd", ev_size=128, dp_sharding_threshold=0.0,
             num_gpus_per_node=8, mem_comm_bw_ratio=8.0e12 / (7 * 153e9), mem_comm_work_ratio=8 / 2,
             memory_cap_for_embedding=240.0)
    a.update(overrides)
    return Namespace(**a)


class _Shards:
    """the shards still to be placed (hotness descending) and the bins they are placed into"""

    def __init__(self, hotness: np.ndarray, n_bins: int, dp_tables=np.array([], dtype=int)):
        self.table_hotness = hotness
        mp_tables = np.setdiff1d(np.arange(hotness.size), dp_tables)
        order = np.argsort(hotness[mp_tables])[::-1]
        self.hot = hotness[mp_tables][order]        # hotness of every shard
        self.table = mp_tables[order]               # table of every shard
        self.n_bins = n_bins
        self.n_split = np.zeros(hotness.size, dtype=int)
        self.n_split[mp_tables] = 1
        self.bins: List[list] = [[] for _ in range(n_bins)]

    def _resort(self):
        order = np.argsort(self.hot)[::-1]
        self.hot, self.table = self.hot[order], self.table[order]

    def _double(self, t) -> bool:
        """replace the shards of table t by twice as many, each half as hot"""
        if self.n_split[t] * 2 > self.n_bins:
            return False
        others = self.table != t
        self.n_split[t] *= 2
        k = self.n_split[t]
        self.hot = np.concatenate((self.hot[others], np.ones(k) * (self.table_hotness[t] / k)))
        self.table = np.concatenate((self.table[others], np.ones(k, dtype=int) * t))
        return True

    def split_hottest(self):
        for s in range(self.table.size):  # the hottest shard whose table can still be split
            if self._double(self.table[s]):
                break
        self._resort()

    def split_table(self, t) -> bool:
        if not self._double(t):
            return False
        self._resort()
        return True

    def recount_splits(self):
        self.n_split = np.zeros_like(self.table_hotness)
        for b in self.bins:
            for t in b:
                self.n_split[t] += 1


class CostModel:
    def __init__(self, hotness_cost: float, table_cost: float, mem_cost: float,
                 mem_capacity: float, table_size: Sequence[int]):
        self.unit_hotness_cost, self.unit_table_cost = hotness_cost, table_cost
        self.unit_mem_cost, self.mem_capacity = mem_cost, mem_capacity
        self.array_table_size = np.array(table_size)

    def evaluate(self, st: _Shards):
        """(total, hotness, table, memory) cost per bin and whether a bin exceeds its memory"""
        hot, tab, mem = [], [], []
        for b in st.bins:
            share = np.array(st.n_split)[b]
            hot.append(self.unit_hotness_cost * (st.table_hotness[b] / share).sum())
            tab.append(self.unit_table_cost * len(b))
            mem.append(self.unit_mem_cost * (self.array_table_size[b] / share).sum())
        hot, tab, mem = np.array(hot), np.array(tab), np.array(mem)
        return (hot + tab, hot, tab, mem), max(mem) > self.mem_capacity

    def reserve_replicated(self, dp_tables):
        self.mem_capacity -= self.array_table_size[dp_tables].sum() * self.unit_mem_cost
        if self.mem_capacity < 0:
            raise Exception("OOM due to DP. Please considering increase the DP threshold")


class Planner:
    def __init__(self, list_hotness: Sequence[int], num_bucket: int, cost_model: CostModel,
                 dp_threshold: float = 0, max_search_iter: int = 20, log_result: bool = False):
        self.hotness = np.array(list_hotness)
        self.n_bins, self.cm = num_bucket, cost_model
        self.max_iter, self.log = max_search_iter, log_result
        # fall-back candidate: every table row-sharded over every bin (the smallest footprint)
        st = _Shards(self.hotness, num_bucket)
        st.bins = [list(range(self.hotness.size)) for _ in range(num_bucket)]
        st.recount_splits()
        cost, oom = self.cm.evaluate(st)
        if oom:
            raise Exception("OOM even with the most memory-efficient sharding plan")
        self.candidates = [(cost[0].max(), cost[1], cost[2], cost[3], st.bins)]
        # small tables are replicated (data parallel) when a threshold is given
        self.dp_tables = np.where(cost_model.array_table_size <
                                  dp_threshold / cost_model.unit_mem_cost)[0]
        self.mp_tables = np.setdiff1d(np.arange(self.hotness.size), self.dp_tables)
        self.state = _Shards(self.hotness, num_bucket, self.dp_tables)
        self.cm.reserve_replicated(self.dp_tables)

    def _place_greedily(self, st: _Shards):
        """shards in hotness order, each into the cheapest bin that has no shard of its table yet
        and still fits; returns the table that fits nowhere (or None) and the last cost"""
        bin_cost = np.zeros(st.n_bins)
        st.bins = [[] for _ in range(st.n_bins)]
        cost = None
        for s in range(st.hot.size):
            t, placed = st.table[s], False
            for b in np.argsort(bin_cost):
                if t in st.bins[b]:
                    continue
                st.bins[b].append(t)
                cost, oom = self.cm.evaluate(st)
                if not oom:
                    placed, bin_cost = True, cost[0]
                    break
                st.bins[b].pop()
            if not placed:
                return t, cost
        return None, cost

    def plan(self):
        t0 = time.time()
        for _ in range(self.max_iter):
            stuck, cost = self._place_greedily(self.state)
            if stuck is None:
                self.candidates.append((cost[0].max(), cost[1], cost[2], cost[3], self.state.bins))
                self.state.split_hottest()
            elif not self.state.split_table(stuck):
                break
        self.candidates.sort(key=lambda c: c[0])
        best = self.candidates[0]
        strategy = [("mp", self.mp_tables.tolist()), ("dp", self.dp_tables.tolist())]
        matrix = best[-1]
        for t in self.dp_tables:
            for b in matrix:
                b.append(t)
        if self.log:
            logging.info("Planner took %f sec" % (time.time() - t0))
            logging.info(strategy)
            logging.info(matrix)
            logging.info("hotness / table / memory cost per GPU: %s %s %s", best[1], best[2], best[3])
        return strategy, matrix


def generate_plan(slot_size_array: List[int], multi_hot_sizes: List[int], num_nodes: int,
                  num_gpus: int, args: Namespace, log_result: bool = False
                  ) -> Tuple[List[List[str]], List[Tuple[str, List[str]]]]:
    """-> (shard_matrix[gpu] = table names, shard_strategy = [("mp", names), ("dp", names)]) for
    EmbeddingCollectionConfig.shard()"""
    n_tables = len(slot_size_array)
    plan = args.sharding_plan
    if plan == "round_robin":
        matrix = [[t for t in range(n_tables) if t % num_gpus == g] for g in range(num_gpus)]
        strategy = [("mp", list(range(n_tables)))]
    elif plan == "uniform":
        matrix = [list(range(n_tables)) for _ in range(num_gpus)]
        strategy = [("mp", list(range(n_tables)))]
    elif plan in ("auto", "hier_auto"):
        bytes_per_elem = {"adagrad": 8, "sgd": 4}[args.optimizer]  # weight (+ accumulator), fp32
        hier = plan == "hier_auto"
        if hier and num_nodes <= 1:
            raise Exception("hier_auto plan is only applicable to configs with more than one node")
        cap = args.memory_cap_for_embedding * (args.num_gpus_per_node if hier else 1)
        cm = CostModel(1, args.mem_comm_bw_ratio / args.mem_comm_work_ratio,
                       args.ev_size * bytes_per_elem * 1e-9, cap, slot_size_array)
        planner = Planner(multi_hot_sizes, num_nodes if hier else num_gpus, cm,
                          dp_threshold=args.dp_sharding_threshold, log_result=log_result)
        strategy, bins = planner.plan()
        # hier_auto plans per node; every GPU of a node gets the node's list
        matrix = [b for b in bins for _ in range(args.num_gpus_per_node)] if hier else bins
    else:
        raise Exception("unknown sharding plan")

    covered = set(t for row in matrix for t in row)
    assert covered == set(range(n_tables)), "Not all tables covered in the sharding plan"
    assert set(t for _, ts in strategy for t in ts) == set(range(n_tables)), \
        "Not all tables covered in the sharding plan"
    if any(len(row) == 0 for row in matrix):
        raise Exception("Currently no empty shard list is allowed")
    shard_matrix = [[str(t) for t in row] for row in matrix]
    shard_strategy = [(kind, [str(t) for t in ts]) for kind, ts in strategy if len(ts) != 0]
    if log_result:
        logging.info("shard_matrix: %s", shard_matrix)
    return shard_matrix, shard_strategy
