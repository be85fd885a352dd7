#!/usr/bin/env python
"""bench.py -- DLRM (Criteo-1TB shape) training step on the MI355X-native embedding hot path.

A "step" is one full pass of the hot path over one synthetic batch: hash/index -> per-slot gather
+ pooling -> (N > 1: all-to-all + reorder) -> bottom MLP -> dot interaction -> top MLP -> BCE
loss -> backward -> (N > 1: reorder + all-to-all) -> sort + segmented gradient reduce + sparse
SGD update -> dense SGD step.  Nothing is skipped inside the timed region.

Contract: `python bench.py --gpus N --steps K --warmup W` (N > 1 under torch.distributed.run);
rank 0 prints ONE JSON line.  See DESIGN.md "Measurement".
"""
import argparse
import json
import os
import sys
import time

import numpy as np
import torch
import torch.distributed as dist

ROOT = os.path.dirname(os.path.abspath(__file__))


def _counter_file(name):
    """the newest round's counter file under profiles/ (tools/measure_round.sh writes rN_<name>)"""
    for r in (6, 5):
        p = os.path.join(ROOT, "profiles", f"r{r}_{name}")
        if os.path.exists(p):
            return p
    return os.path.join(ROOT, "profiles", f"r6_{name}")
sys.path.insert(0, ROOT)

# R/test/embedding_collection_test/dgx_a100_one_hot.py:24-51 -- Criteo-1TB slot_size_array
CRITEO_1TB = [39884406, 39043, 17289, 7420, 20263, 3, 7120, 1543, 63, 38532951, 2953546, 403346,
              10, 2208, 11938, 155, 4, 976, 14, 39979771, 25641295, 39664984, 585935, 12972, 108,
              36]
DENSE_DIM = 13
BOTTOM = [512, 256, 128]          # R/samples/dlrm/train.py:415-458 bottom MLP
TOP = [1024, 1024, 512, 256, 1]   # top MLP
HBM_PEAK_GBPS = 8000.0            # MI355X_MICROARCH.md: HBM3E 8.0 TB/s spec (6.29 TB/s copy)


LINE_CAP = 4096  # bytes; round 4's 21 KB line outgrew the driver's channel


def _r(x, sig=5):
    """numbers to `sig` significant digits (the line is a summary; the file keeps full precision)"""
    if isinstance(x, bool) or x is None:
        return x
    if isinstance(x, int):
        return x
    if isinstance(x, float):
        if x != x or x in (float("inf"), float("-inf")):
            return None
        return float(f"{x:.{sig}g}")
    return x


def _pick(d, keys):
    return {k: _r(d[k]) for k in keys if isinstance(d, dict) and k in d and d[k] is not None}


def _roof_short(r, extra=()):
    if not isinstance(r, dict):
        return None
    o = _pick(r, ("bound", "achieved", "peak", "unit", "frac", "traffic", "frac_compulsory",
                  "frac_traffic", "us", "avg_launch_us") + tuple(extra))
    o.setdefault("traffic", None)
    return o


def compact_line(full, extra_file):
    """the ONE stdout line of the contract, <= LINE_CAP bytes: the contract's keys, the roofline
    objects as numbers and the cpu baseline; everything else lives in `extra_file` (the full
    record, also printed to stderr).  tests/test_bench_line_cpu.py holds the cap."""
    cfg = full.get("config", {}) if isinstance(full.get("config"), dict) else {}
    line = _pick(full, ("metric", "value", "unit", "n_gpus", "steps", "warmup", "ms_per_step",
                        "higher_is_better", "scaling"))
    line["vs_baseline"] = full.get("vs_baseline")
    line["dtype"] = full.get("precision") or str(full.get("dtype", ""))[:16]
    line["data"] = "synthetic"
    line["config"] = _pick(cfg, ("workload", "batch_per_gpu", "global_batch", "slots", "emb_dim",
                                 "table_rows_total", "parallelism", "exchange", "new_keys_per_step",
                                 "distinct_rows_per_batch", "gather_fused_into_interaction",
                                 "final_loss", "key_distribution"))
    if isinstance(line["config"].get("workload"), str):
        line["config"]["workload"] = line["config"]["workload"][:200]
    r = full.get("roofline")
    if isinstance(r, dict):
        ro = _roof_short(r, ("algorithmic_bytes_per_launch", "launches_timed_region",
                             "launches_inline", "avg_launch_us_timed_region", "traffic_stale"))
        ro["kernel"] = str(r.get("kernel", "")).split(" ")[0]
        line["roofline"] = ro
    for k in ("roofline_update", "roofline_index", "roofline_uniform", "roofline_fp32"):
        if isinstance(full.get(k), dict):
            line[k] = _roof_short(full[k], ("algorithmic_bytes", "algorithmic_bytes_per_launch",
                                            "traffic_ratio", "us_grouping_ahead",
                                            "frac_grouping_ahead"))
            line[k].pop("peak", None)
            line[k].pop("unit", None)
            line[k].pop("bound", None)
    cb = full.get("cpu_baseline")
    if isinstance(cb, dict):
        c = _pick(cb, ("value", "unit", "cores", "kind", "error"))
        if "sample" in cb:
            c["sample"] = str(cb["sample"]).split(";")[0][:300]
        if isinstance(cb.get("port"), dict):
            c["port"] = _pick(cb["port"], ("value", "cores"))
        line["cpu_baseline"] = c
    st = full.get("strong")
    if isinstance(st, dict):
        line["strong"] = _pick(st, ("value", "ms_per_step", "scaling", "error"))
        if isinstance(st.get("config"), dict):
            line["strong"]["global_batch"] = st["config"].get("global_batch")
    # one number per extra leg (the legs themselves: extra_file)
    ex = full.get("extra")
    if isinstance(ex, dict):
        summ = {}
        for name, leg in ex.items():
            if not isinstance(leg, dict):
                continue
            if "error" in leg:
                summ[name] = "error"
            elif "ms_per_step" in leg:
                summ[name] = {"ms_per_step": _r(leg["ms_per_step"], 4)}
            elif "forward_backward_update_us" in leg:
                summ[name] = {"fwd_bwd_upd_us": _r(leg["forward_backward_update_us"], 4)}
            elif "lookup_us" in leg:
                summ[name] = {"lookup_us": _r(leg["lookup_us"], 4), "update_us": _r(leg.get("update_us"), 4)}
            elif "summary" in leg:
                summ[name] = leg["summary"]
        line["extra_summary"] = summ
    line["extra_file"] = extra_file
    s = json.dumps(line, separators=(",", ":"))
    # never exceed the cap: shed the optional parts in order of (un)importance
    for k in ("extra_summary", "strong", "roofline_fp32", "roofline_uniform"):
        if len(s) <= LINE_CAP:
            break
        line.pop(k, None)
        s = json.dumps(line, separators=(",", ":"))
    if len(s) > LINE_CAP:
        line["config"] = {"workload": str(cfg.get("workload", ""))[:120]}
        if isinstance(line.get("cpu_baseline"), dict):
            line["cpu_baseline"]["sample"] = line["cpu_baseline"].get("sample", "")[:80]
        s = json.dumps(line, separators=(",", ":"))
    assert len(s) <= LINE_CAP and "\n" not in s, len(s)
    return s


def emit(full, extra_file):
    """full record -> extra_file (and stderr under HCTR_BENCH_VERBOSE=1); the compact line ->
    stdout, last, flushed"""
    path = extra_file
    try:
        os.makedirs(os.path.dirname(os.path.abspath(path)) or ".", exist_ok=True)
        with open(path, "w") as f:
            json.dump(full, f, indent=1)
    except OSError as e:
        path = f"(not written: {e})"
    if os.environ.get("HCTR_BENCH_VERBOSE"):
        sys.stderr.write(json.dumps(full, indent=1) + "\n")
    sys.stderr.write(f"[bench] full record: {path}\n")
    sys.stderr.flush()
    sys.stdout.write(compact_line(full, path) + "\n")
    sys.stdout.flush()


def _csrc_hash():
    """tools/csrc_hash.py: sha256 over the kernel sources, the stamp of the counter files"""
    import glob
    import hashlib
    h = hashlib.sha256()
    d = os.path.join(ROOT, "hugectr_amd", "csrc")
    for f in sorted(glob.glob(os.path.join(d, "*.hip")) + glob.glob(os.path.join(d, "*.h")) +
                    glob.glob(os.path.join(d, "*.cpp"))):
        h.update(os.path.basename(f).encode())
        h.update(open(f, "rb").read())
    return h.hexdigest()[:16]


def powerlaw(rng, n, vocab, alpha):
    """IntPowerLawDataSimulator (R/HugeCTR/include/data_generator.hpp:108-129), vectorised."""
    if alpha <= 0:
        return rng.integers(0, vocab, size=n).astype(np.int64)
    u = rng.random(n, dtype=np.float32).astype(np.float64)
    a = 1.0 - alpha
    y = ((float(vocab) ** a - 1.0) * u + 1.0) ** (1.0 / a)
    return np.clip(np.round(y) - 1, 0, vocab - 1).astype(np.int64)


def make_keys(rng, batch, sizes, alpha):
    offs = np.concatenate([[0], np.cumsum(sizes)[:-1]]).astype(np.int64)
    keys = np.empty((batch, len(sizes)), dtype=np.int64)
    for s, v in enumerate(sizes):
        keys[:, s] = powerlaw(rng, batch, v, alpha) + offs[s]
    return keys.reshape(-1)


def cpu_baseline(sizes, alpha, D, seed, budget_s=10.0):
    """CPU baselines on this box's host cores, on bounded samples (DESIGN.md "Measurement").

    kind "reference": the REFERENCE'S OWN CPU path -- oracle/_ref/libref_embedding.so, i.e.
    `SparseEmbeddingHashCpu` (R/test/utest/embedding/sparse_embedding_hash_cpu.hpp) compiled from
    the reference checkout; its forward() starts with read_a_batch (:343-377), the reference's own
    parse of the Norm dataset records, followed by hash lookup, pooling, backward and
    update_params (:920-1015).  Single-threaded, as that code is.  Timed on (a) the bench workload's
    shape -- 26 Criteo-1TB slots, D = 128, SGD, power-law one-hot keys, tables scaled 1/64,
    B = 1024 -- which is `value`, and (b) BASELINE configs[0] / SURVEY C1 exactly: README DCN
    slot sizes, B = 1024, D = 16, Adam (Global), up to 20 warm-up + 200 timed iterations inside a
    time bound.  Its `cpu_csr_sort` (:541-561) is an O(nnz^2) odd-even transposition sort:
    ~7e8 compare-swaps per 1024-sample batch dominate every iteration (and make B = 8192, 64x
    that, impractical).

    kind "port" (key `port`): oracle/hctr_oracle.c, the line-by-line C restatement with a stable
    O(n log n) sort and OpenMP, embedding forward + backward + SGD only, B = 8192, tables 1/64."""
    from oracle import pyoracle as orc
    from oracle import ref_baseline as rb
    scale = 64
    ssz = [max(1, v // scale) for v in sizes]
    cores = os.cpu_count() or 1
    out = {}
    # ---- the reference's own CPU path (reader + embedding), 1 thread --------------------------
    if rb.available():
        c3 = rb.time_reference_cpu(ssz, 1024, D, "sgd", 0, alpha, 2, 40, budget_s, seed=seed,
                                   lr=0.01)
        c1 = rb.time_reference_cpu(rb.C1_SLOTS, 1024, 16, "adam", 1, 1.3, 20, 200, 4.0 * budget_s,
                                   seed=seed + 1, lr=0.001)
        out.update({
            "value": c3["samples_per_s"], "unit": "samples/s", "cores": 1, "kind": "reference",
            "sample": f"reference SparseEmbeddingHashCpu (read_a_batch + hash + forward + backward "
                      f"+ SGD update, single-threaded as written): {c3['iters']} timed iterations "
                      f"after {c3['warmup_iters']} warm-up, {c3['seconds']:.1f} s of CPU work, "
                      f"B=1024, 26 Criteo-1TB slots one-hot power-law alpha={alpha}, D={D}, tables "
                      f"scaled 1/{scale} ({c3['rows']} rows); O(nnz^2) odd-even sort inside "
                      f"({c3['s_per_iter'] * 1e3:.0f} ms / iteration); host has {cores} logical cpus",
            # the reference's three calls timed separately: forward() =
            # read_a_batch + hash get/insert + pooling, backward(), update_params() = sort /
            # unduplicate + optimizer
            "stage_ms_per_iteration": c3["stage_ms_per_iter"],
            "c1_dcn_readme": {
                "value": c1["samples_per_s"], "unit": "samples/s", "cores": 1, "kind": "reference",
                "sample": f"BASELINE configs[0] (SURVEY C1): README DCN slot sizes "
                          f"({c1['rows']} rows), B=1024, 26 slots one-hot power-law alpha=1.3, "
                          f"D=16, Adam Global, {c1['warmup_iters']} warm-up + {c1['iters']} timed "
                          f"iterations ({c1['seconds']:.1f} s; the target 20 + 200 is cut by the "
                          f"time bound), reader + embedding fwd/bwd/update, no dense tower",
                "ms_per_iteration": c1["s_per_iter"] * 1e3,
                "stage_ms_per_iteration": c1["stage_ms_per_iter"]},
        })
    # ---- the port (restated oracle), embedding only, 1 and many threads -------------------------
    V, S, B = sum(ssz), len(ssz), 8192
    rng = np.random.default_rng(seed)
    table = (rng.random((V, D), dtype=np.float32) - 0.5) * 0.1
    ro = np.arange(B * S + 1, dtype=np.int64)
    g = rng.standard_normal((B * S, D)).astype(np.float32)
    opt = orc.OptParamsC()
    opt.optimizer, opt.update_type, opt.lr, opt.scaler, opt.times = orc.OPT_SGD, 0, 0.01, 1.0, 1
    threads = min(cores, 64)
    res = {}
    for label, th in (("1t", 1), ("mt", threads)):
        ht = orc.HashTable(V, 8)
        ht.get_insert(make_keys(rng, B, ssz, alpha))  # warm
        t0, n = time.perf_counter(), 0
        while time.perf_counter() - t0 < budget_s / 2 and n < 100:
            keys = make_keys(rng, B, ssz, alpha)
            t1 = time.perf_counter()
            vi = ht.get_insert(keys)
            orc.forward(ro, vi, table, D, 0, threads=th)
            wg = orc.backward(ro, g, D, 0)
            orc.update_params(ro, vi, wg, opt, table, threads=th)
            res.setdefault(label, []).append(time.perf_counter() - t1)
            n += 1
    best = {k: B / float(np.median(v)) for k, v in res.items()}
    port = {"value": best["mt"], "unit": "samples/s", "cores": threads, "kind": "port",
            "value_1_thread": best["1t"],
            "sample": f"{len(res['1t'])} + {len(res['mt'])} batches (1 thread + {threads} threads, "
                      f"{sum(res['1t']) + sum(res['mt']):.1f} s of CPU work): "
                      f"embedding fwd+bwd+SGD update only (no reader, no dense tower), B={B}, 26 "
                      f"Criteo-1TB slots one-hot power-law alpha={alpha}, D={D}, tables scaled "
                      f"1/{scale} ({V} rows), host has {cores} logical cpus"}
    if out:
        out["port"] = port
        return out
    return port  # oracle/_ref absent (it is built where the reference checkout is present)


def gen_keys(gen, n, sizes, alpha, dev):
    """[n, len(sizes)] int64 keys with the readers' cumulative slot offsets added
    (R/HugeCTR/src/pybind/add_input.cpp:315-317), drawn on the GPU: IntPowerLawDataSimulator's
    inverse CDF (R/HugeCTR/include/data_generator.hpp:108-129), uniform when alpha <= 0.  Every
    rank seeds the same generator, so all ranks hold the same full-batch CSR (reader semantics)."""
    cols, off = [], 0
    for v in sizes:
        u = torch.rand(n, device=dev, generator=gen, dtype=torch.float32).double()
        if alpha <= 0:
            k = (u * v).to(torch.int64).clamp_(0, v - 1)
        else:
            e = 1.0 - alpha
            y = ((float(v) ** e - 1.0) * u + 1.0) ** (1.0 / e)
            k = (torch.round(y) - 1).clamp_(0, v - 1).to(torch.int64)
        cols.append(k + off)
        off += v
    return torch.stack(cols, 1).reshape(-1)


def build_dlrm(B, sizes, D, precision, world, lr=0.01, overlap=True):
    """BASELINE configs[2] written against the `hugectr` surface: DLRM (bottom MLP 512-256-128,
    dot Interaction, top MLP 1024-1024-512-256-1, BinaryCrossEntropyLoss) over ONE
    LocalizedSlotSparseEmbeddingHash with the Criteo-1TB slot_size_array, SGD, global batch B over
    `world` GPUs -- the layer list of R/samples/dlrm/train.py:406-470 on the legacy embedding the
    configuration names."""
    import hugectr_amd.hugectr as hugectr
    S = len(sizes)
    mixed = precision == "fp16"
    solver = hugectr.CreateSolver(
        max_eval_batches=1, batchsize_eval=B, batchsize=B, lr=lr, vvgpu=[list(range(world))],
        repeat_dataset=True, i64_input_key=True, use_mixed_precision=mixed,
        scaler=1024.0 if mixed else 1.0,
        train_intra_iteration_overlap=overlap, train_inter_iteration_overlap=overlap)
    optimizer = hugectr.CreateOptimizer(optimizer_type=hugectr.Optimizer_t.SGD,
                                        update_type=hugectr.Update_t.Local, atomic_update=True)
    reader = hugectr.DataReaderParams(data_reader_type=hugectr.DataReaderType_t.Parquet,
                                      source=["(batches resident in HBM)"], eval_source="",
                                      check_type=hugectr.Check_t.Non, slot_size_array=sizes)
    m = hugectr.Model(solver, reader, optimizer)
    L, T, A = hugectr.DenseLayer, hugectr.Layer_t, hugectr.Activation_t
    m.add(hugectr.Input(label_dim=1, label_name="label", dense_dim=DENSE_DIM, dense_name="dense",
                        data_reader_sparse_param_array=[
                            hugectr.DataReaderSparseParam("data1", 1, True, S)]))
    m.add(hugectr.SparseEmbedding(
        embedding_type=hugectr.Embedding_t.LocalizedSlotSparseEmbeddingHash,
        slot_size_array=sizes, embedding_vec_size=D, combiner="sum",
        sparse_embedding_name="sparse_embedding1", bottom_name="data1", optimizer=optimizer))
    m.add(L(layer_type=T.MLP, bottom_names=["dense"], top_names=["mlp1"], num_outputs=BOTTOM,
            act_type=A.Relu))
    m.add(L(layer_type=T.Interaction, bottom_names=["mlp1", "sparse_embedding1"],
            top_names=["interaction1"]))
    m.add(L(layer_type=T.MLP, bottom_names=["interaction1"], top_names=["mlp2"], num_outputs=TOP,
            activations=[A.Relu] * (len(TOP) - 1) + [A.Non]))
    m.add(L(layer_type=T.BinaryCrossEntropyLoss, bottom_names=["mlp2", "label"],
            top_names=["loss"]))
    return m


def dlrm_leg(a, precision, steps, warmup, world, rank, dev, scaling="weak", alpha=None,
             sizes=None, label=None):
    """one measurement of the DLRM Criteo-1TB step through hugectr.Model.train() -- the product
    surface, on 1 or N GPUs -- at `precision` (see --precision).  scaling = "weak": a.batch
    samples per GPU (global batch N x a.batch); "strong": global batch a.batch (per GPU a.batch/N),
    BASELINE configs[2] read literally.  Every step sees a batch it has not seen before (keys are
    inserted live).  Builds and releases its own model: the 89.5 GiB table exists once at a time."""
    from hugectr_amd import _lib
    alpha = a.alpha if alpha is None else alpha
    sizes = sizes or [max(1, int(v * a.table_scale)) for v in CRITEO_1TB]
    S, D = len(sizes), a.dim
    Bl = a.batch if scaling == "weak" else a.batch // world
    B = Bl * world
    esz = 2 if precision == "fp16" else 4
    m = build_dlrm(B, sizes, D, precision, world, overlap=not a.no_overlap)
    # ---- synthetic batches, resident in HBM before the timed region; one per step ----------------
    sel = 0
    if world > 1 and os.environ.get("HCTR_EXCHANGE", "auto") == "auto" and not a.no_overlap:
        sel = m._SEL_WARM + 2 * m._SEL_TIMED + m._SEL_SWITCH + 1  # the exchange selection's steps
    inline = 8 if world == 1 else 0  # unseen batches for the in-line stage times (see below)
    ov_prev = os.environ.get("HCTR_UPDATE_OVERLAP")
    nb = a.nbatches if a.nbatches > 0 else sel + warmup + steps + inline + 1
    gk = torch.Generator(device=dev)
    gk.manual_seed(1234)                     # the same full-batch CSR on every rank
    gd = torch.Generator(device=dev)
    gd.manual_seed(99 + rank)
    ro = torch.arange(0, B * S + 1, dtype=torch.int64, device=dev)
    batches, uniq = [], []
    for _ in range(nb):
        keys = gen_keys(gk, B, sizes, alpha, dev)
        dense = torch.rand((Bl, DENSE_DIM), device=dev, generator=gd)
        lab = (torch.rand((Bl, 1), device=dev, generator=gd) < 0.5).float()
        batches.append({"dense": dense, "label": lab, "sparse": {"data1": (ro, keys)}})
    for bt in batches[-4:]:  # distinct rows per batch on this rank's slots: the compulsory reads
        k = bt["sparse"]["data1"][1].view(B, S)[:, rank::world]
        uniq.append(int(torch.unique(k).numel()))
    m.reader_override = _CycleReader(batches)
    m.compile()
    se, p, emb, ex, _ = m._emb["sparse_embedding1"]
    spr = emb.slots_on_rank
    my_rows = sum(v for i, v in enumerate(sizes) if i % world == rank)

    def sync():
        m._drain_prefetch()
        torch.cuda.synchronize()

    # (N > 1: the payload selection's steps run the per-sample payload, i.e. the gather kernel,
    #  whichever payload is kept: its launches are timed here in case the timed region has none)
    emb.profiling(True)
    for _ in range(sel):
        m.train()
    pool_prof = (0.0, 0)
    if sel:
        sync()
        pool_prof = emb.profile().get("gather_pool", (0.0, 0))
    emb.profiling(True)
    for _ in range(warmup):
        m.train()
    sync()
    rows_before = emb.get_vocabulary_size()
    emb.profiling(True)  # (re-arms the counters: the stage times cover the timed steps only)
    if world > 1:
        dist.barrier()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(steps):
        m.train()
    sync()
    if world > 1:
        dist.barrier()
    torch.cuda.synchronize()
    elapsed = time.perf_counter() - t0
    if world > 1:
        t = torch.tensor([elapsed], dtype=torch.float64)
        if dist.get_backend() != "gloo":
            t = t.to(dev)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        elapsed = float(t.item())
    prof_timed = emb.profile()
    emb.profiling(False)
    upd_overlapped = bool(getattr(m, "_upd_overlap", False) and getattr(m, "_upd_overlap_on", False)
                          and (ov_prev or "auto") != "0")
    m.check_overflow()
    new_keys = (emb.get_vocabulary_size() - rows_before) / max(steps, 1)
    # The timed region runs the product's schedule: on one GPU the sparse update starts from the
    # gradient hook on a side stream, under the bottom MLP's backward (hugectr.py), so its events
    # there bracket shared-chip time.  The stage times the rooflines are computed from are taken
    # right after, on `inline` further unseen batches with the update in line (its own time).
    prof = prof_timed
    prof_ahead = None
    if world == 1 and inline > 0:
        os.environ["HCTR_UPDATE_OVERLAP"] = "0"
        # (a) HCTR_PREWORK=1: the update's grouping kernels (hot rows' chunk sort, cold rows'
        # count / base / scatter) right behind the index stage on side streams, the update stage
        # itself = the two reduces + join / apply
        pw_prev = os.environ.get("HCTR_PREWORK")
        os.environ["HCTR_PREWORK"] = "1"
        emb.profiling(True)
        for _ in range(inline // 2):
            m.train()
        sync()
        prof_ahead = emb.profile()
        # (b) everything of the update inside the update stage (HCTR_PREWORK=0, the default): the
        # stage time `roofline_update` is computed from
        os.environ["HCTR_PREWORK"] = "0"
        emb.profiling(True)
        for _ in range(inline - inline // 2):
            m.train()
        sync()
        prof = emb.profile()
        emb.profiling(False)
        if pw_prev is None:
            os.environ.pop("HCTR_PREWORK", None)
        else:
            os.environ["HCTR_PREWORK"] = pw_prev
    # the same stages with NO unseen key (batches the tables have met): the index stage's floor
    steady_us = None
    if world == 1:
        m.reader = _CycleReader(batches[-4:])
        m._lookahead = None
        for _ in range(4):
            m.train()
        sync()
        emb.profiling(True)
        for _ in range(8):
            m.train()
        sync()
        steady_us = {k: (v[0] / max(v[1], 1)) * 1e3 for k, v in emb.profile().items()}
        emb.profiling(False)
        if ov_prev is None:
            os.environ.pop("HCTR_UPDATE_OVERLAP", None)
        else:
            os.environ["HCTR_UPDATE_OVERLAP"] = ov_prev
    xrep = m.exchange_report()["sparse_embedding1"]

    # ---- roofline of the gather+pool kernel (DESIGN.md section 5 / SURVEY 8d) -------------------
    nnz_g = B * spr  # one-hot: one key per (sample, slot on this rank)
    fused = bool(xrep.get("gather_fused_into_interaction"))
    U = sum(uniq) / max(len(uniq), 1)
    n_ins = S + 1
    out_len = D + n_ins * (n_ins - 1) // 2 + 1
    if fused:
        # one GPU: the gather rides in the interaction kernel (hctr_emb_forward_interaction).  Per
        # launch: row index + fp32 row per key, the bottom-MLP row, the pooled vectors written
        # once, the interaction output -- SURVEY 8(d)'s gather bytes (without the key read, which
        # is the index stage's) + the interaction's own input / output, minus the pooled re-read
        io = Bl * D * esz + Bl * out_len * esz
        alg_bytes = nnz_g * 8 + nnz_g * D * 4 + B * spr * D * esz + io
        comp_bytes = nnz_g * 8 + U * D * 4 + B * spr * D * esz + io
    else:
        alg_bytes = nnz_g * 8 + nnz_g * 8 + nnz_g * D * 4 + B * spr * D * esz
        # compulsory = what must cross the HBM interface even with a perfect cache: every key and
        # row index once, every DISTINCT row once, every output element once
        comp_bytes = nnz_g * 8 + nnz_g * 8 + U * D * 4 + B * spr * D * esz
    pool_ms, pool_n = prof["gather_pool"]
    if pool_n == 0:  # unique-row payload kept: the gather kernel only ran during the selection
        pool_ms, pool_n = pool_prof
    pool_s = pool_ms / max(pool_n, 1) * 1e-3
    achieved = alg_bytes / pool_s / 1e9 if pool_ms > 0 else 0.0
    # HBM bytes per launch from the PMC counters (rocprofv3, separate passes, tools/measure_round.sh
    # -> profiles/r4_pmc_hbm_traffic_<label>.json, stamped with the kernel sources' hash; a bench
    # process cannot read the counters of its own kernels)
    pmc, pmc_src, pmc_upd, pmc_idx, pmc_stale, mfma_busy = None, None, None, None, None, None
    tag = label or ("fp32" if esz == 4 else "fp16")
    here = _csrc_hash()
    if world == 1 and D == 128 and a.batch == 65536 and a.table_scale == 1.0:
        pmc_path = _counter_file(f"pmc_hbm_traffic_{tag}.json")
        if not os.path.exists(pmc_path):
            pmc_path = os.path.join(ROOT, "profiles", f"r4_pmc_hbm_traffic_{tag}.json")
        try:
            j = json.load(open(pmc_path))
            if abs(float(j.get("alpha", 1.1)) - alpha) < 1e-9:
                pmc_src = os.path.relpath(pmc_path, ROOT) + (
                    f" @ {j['commit']}" if j.get("commit") else "")
                # counters are quoted only while the kernel sources are the ones they were taken
                # on (tools/csrc_hash.py; tools/measure_round.sh regenerates the files)
                pmc_stale = j.get("csrc_hash") != here
                if not pmc_stale:
                    ks = j["kernels"]
                    pmc = ks["interaction_fwd16_gather_kernel" if fused
                             else "pool_vec4_kernel"]["hbm_bytes_per_launch"]

                    def per_step(names, once):
                        # bytes per step of a group of kernels: per-launch average x launches per
                        # step (= its launch count over that of a kernel launched once a step)
                        n1 = ks[once]["launches_averaged"][0]
                        return sum(ks[k]["hbm_bytes_per_launch"] *
                                   ks[k]["launches_averaged"][0] / n1 for k in names if k in ks)
                    upd_once = ("cold_reduce_kernel" if "cold_reduce_kernel" in ks
                                else "seg_reduce_kernel")
                    if upd_once in ks:
                        pmc_upd = per_step(("expand_pairs_kernel", "rs_hist_kernel",
                                            "rs_colscan_kernel", "rs_scatter_kernel",
                                            "hot_sort_kernel", "hot_reduce_kernel",
                                            "hot_join_kernel", "hot_apply_kernel",
                                            "cold_count_kernel", "cold_base_kernel",
                                            "cold_scatter_kernel", "cold_reduce_kernel",
                                            "seg_reduce_kernel", "seg_combine_kernel",
                                            "seg_combine_big_kernel"), upd_once)
                    if "ht_probe_insert_kernel" in ks:
                        pmc_idx = per_step(("ht_probe_insert_kernel", "ht_finish_kernel"),
                                           "ht_probe_insert_kernel")
        except Exception:
            pmc = None
        # matrix-pipe utilisation of the interaction kernels (SQ pass of tools/measure_round.sh)
        try:
            sq_path = _counter_file("pmc_sq_counters.json")
            if not os.path.exists(sq_path):
                sq_path = os.path.join(ROOT, "profiles", "r4_pmc_sq_counters.json")
            j = json.load(open(sq_path))
            if j.get("csrc_hash") == here and tag == "fp16":
                mfma_busy = {k: {"mfma_util_of_1024_simds": v.get("mfma_util_of_1024_simds"),
                                 "wave_cycles_split": v.get("wave_cycles_split")}
                             for k, v in j["kernels"].items() if k.startswith("interaction_")}
                mfma_busy["source"] = os.path.relpath(sq_path, ROOT) + (
                    f" @ {j['commit']}" if j.get("commit") else "")
        except Exception:
            mfma_busy = None
    stage_us = {k: (v[0] / max(v[1], 1)) * 1e3 for k, v in prof.items()}
    # update (a12): SURVEY 8(d) bytes = nnz (8 + K) + B S_g D E (gradients read) + U D 4 x 2 (rows)
    upd_bytes = nnz_g * 16 + B * spr * D * esz + U * D * 4 * 2
    upd_s = (stage_us.get("sort", 0.0) + stage_us.get("segmented_update", 0.0)) * 1e-6
    upd_ahead_s = None
    if prof_ahead is not None:
        sa = {k: (v[0] / max(v[1], 1)) * 1e3 for k, v in prof_ahead.items()}
        upd_ahead_s = (sa.get("sort", 0.0) + sa.get("segmented_update", 0.0)) * 1e-6
    idx_bytes = nnz_g * (8 + 16 + 8)
    idx_s = stage_us.get("hash_index", 0.0) * 1e-6
    per_rank = None
    if world > 1:
        mine = {"rank": rank, "device": torch.cuda.current_device(), "backend": dist.get_backend(),
                "slots": spr, "table_rows": my_rows, "stage_us": stage_us,
                "exchange": xrep, "new_keys_per_step": new_keys}
        per_rank = [None] * world
        dist.all_gather_object(per_rank, mine)
    out = {
        "metric": "samples/sec (whole node) + embedding-gather HBM GB/s, DLRM Criteo-1TB",
        "value": B * steps / elapsed, "unit": "samples/s", "n_gpus": world, "steps": steps,
        "warmup": warmup, "ms_per_step": elapsed / steps * 1e3, "higher_is_better": True,
        "scaling": scaling, "vs_baseline": None,
        "dtype": {"fp16": "fp16 = the reference's mixed precision (use_mixed_precision, scaler 1024): "
                          "fp16 pooled vectors + top gradients + dense GEMMs (fp32 accumulate), "
                          "fp32 tables / pooling accumulation / sparse SGD, loss scaling",
                  "fp32": "fp32 = the reference's default: fp32 tables, pooled vectors, gradients, "
                          "sparse SGD and dense GEMMs (interaction: fp32 I/O, 3x bf16-split MFMA)"
                  }[precision],
        "precision": precision,
        "data": f"synthetic power-law alpha={alpha} (uniform if 0), one-hot, resident in HBM, a "
                f"new batch every step ({nb} batches)",
        "config": {"workload": "BASELINE configs[2]: DLRM Criteo-1TB slot_size_array, "
                               "LocalizedSlotSparseEmbeddingHash, emb_dim=128, SGD, bs=65536 " +
                               ("per GPU (weak scaling)" if scaling == "weak" else
                                "global (strong scaling)"),
                   "surface": "hugectr.Model.train() (hugectr_amd/hugectr.py): CreateSolver, "
                              "SparseEmbedding, DenseLayer MLP / Interaction / BinaryCrossEntropyLoss",
                   "batch_per_gpu": Bl, "global_batch": B,
                   "exchange": xrep["payload"],
                   "exchange_selection_ms_per_step": xrep.get("selection_ms_per_step"),
                   "intra_iteration_overlap": xrep["intra_iteration_overlap"],
                   "inter_iteration_overlap": xrep["inter_iteration_overlap"],
                   "slots": S, "emb_dim": D, "table_rows_total": sum(sizes),
                   "table_rows_this_rank": my_rows,
                   "parallelism": f"slot-sharded x{world} + dp{world}",
                   "new_keys_per_step": new_keys, "distinct_rows_per_batch": U,
                   "gather_fused_into_interaction": fused,
                   "final_loss": m.get_current_loss(),
                   "dense_gemm_selection": getattr(m, "_gemm_selection", "off")},
        "roofline": {"bound": "hbm",
                     "kernel": ("interaction_fwd16_gather_kernel (gather + pooling fused into the "
                                "dot interaction: table rows -> LDS tile -> MFMA, pooled vectors "
                                "written once)" if fused else
                                "pool_vec4_kernel (gather + intra-slot pooling)"),
                     "achieved": achieved, "peak": HBM_PEAK_GBPS, "unit": "GB/s",
                     "frac": achieved / HBM_PEAK_GBPS, "traffic": pmc,
                     "traffic_source": pmc_src, "traffic_stale": pmc_stale,
                     # SQ_VALU_MFMA_BUSY_CYCLES over the 1024 SIMDs' cycles, and where the waves'
                     # cycles go, for the interaction kernels (north_star: "MFMA utilisation on the
                     # interaction"); null when the counter file is not of these sources
                     "mfma_busy": mfma_busy,
                     "algorithmic_bytes_per_launch": alg_bytes,
                     # `achieved` / `avg_launch_us` are of the in-line launches (the update in line,
                     # the kernel alone on the chip); the timed region's launches share the chip
                     # with the overlapped update on one GPU -- both averages are given
                     "launches_inline": pool_n, "avg_launch_us": pool_s * 1e6,
                     "launches_timed_region": prof_timed.get("gather_pool", (0.0, 0))[1],
                     "avg_launch_us_timed_region":
                         (prof_timed["gather_pool"][0] / max(prof_timed["gather_pool"][1], 1) * 1e3)
                         if prof_timed.get("gather_pool", (0, 0))[1] else None,
                     "frac_traffic": (pmc / pool_s / 1e9 / HBM_PEAK_GBPS)
                     if pmc and pool_ms > 0 else None,
                     # SURVEY 8(d) counts duplicate rows; power-law keys repeat hot rows, which L2
                     # / Infinity Cache serve, so `frac` can reach 1 without saying much about the
                     # kernel.  compulsory_bytes counts every DISTINCT row once: the bytes no cache
                     # can remove; `traffic` is what the counters saw cross the HBM interface
                     "compulsory_bytes": comp_bytes,
                     "frac_compulsory": (comp_bytes / pool_s / 1e9 / HBM_PEAK_GBPS)
                     if pool_ms > 0 else None,
                     "hbm_traffic_gbps": (pmc / pool_s / 1e9) if pmc and pool_ms > 0 else None},
        "roofline_update": {"bound": "hbm", "kernels": "hot rows (per-chunk LDS sort, tile reduce, "
                            "chunk-ordered apply) beside the cold rows' chain (counted per row: "
                            "count / base / scatter, then singles + sorted short runs + long runs "
                            "with the optimizer folded in); `us` = fork .. join of the two chains "
                            "with ALL of it inside the update stage (HCTR_PREWORK=0); "
                            "`us_grouping_ahead` = the same stage under HCTR_PREWORK=1, the "
                            "grouping kernels started behind the index stage on side streams "
                            "(step time within noise of the default, so off by default)",
                            "us_grouping_ahead": (upd_ahead_s * 1e6) if upd_ahead_s else None,
                            "frac_grouping_ahead": (upd_bytes / upd_ahead_s / 1e9 / HBM_PEAK_GBPS)
                            if upd_ahead_s else None,
                            "achieved": (upd_bytes / upd_s / 1e9) if upd_s > 0 else None,
                            "peak": HBM_PEAK_GBPS, "unit": "GB/s",
                            "frac": (upd_bytes / upd_s / 1e9 / HBM_PEAK_GBPS) if upd_s > 0 else None,
                            "algorithmic_bytes": upd_bytes, "us": upd_s * 1e6,
                            # HBM bytes per step of these kernels (same PMC passes as above)
                            "traffic": pmc_upd, "traffic_source": pmc_src if pmc_upd else None,
                            "traffic_ratio": (pmc_upd / upd_bytes) if pmc_upd else None},
        "roofline_index": {"bound": "hbm", "kernels": "hash index stage (filter + probe / insert)",
                           "achieved": (idx_bytes / idx_s / 1e9) if idx_s > 0 else None,
                           "peak": HBM_PEAK_GBPS, "unit": "GB/s",
                           "frac": (idx_bytes / idx_s / 1e9 / HBM_PEAK_GBPS) if idx_s > 0 else None,
                           "algorithmic_bytes": idx_bytes, "us": idx_s * 1e6,
                           "traffic": pmc_idx, "traffic_source": pmc_src if pmc_idx else None,
                           "traffic_ratio": (pmc_idx / idx_bytes) if pmc_idx else None},
        "stage_us_per_step": stage_us,
        # the same stages as the timed region saw them (one GPU: the update overlapped with the
        # bottom MLP's backward -- shared-chip time, not the kernels' own)
        "stage_us_per_step_timed_region": {k: (v[0] / max(v[1], 1)) * 1e3
                                           for k, v in prof_timed.items()},
        # (auto: the product switches it off when the update is a long bandwidth-bound kernel --
        #  hugectr.py _overlap_update_now; this is the mode the timed region ended in)
        "update_overlapped_with_dense_backward": bool(
            upd_overlapped and xrep.get("gather_fused_into_interaction")),
        "stage_us_per_step_no_new_keys": steady_us,
        "embedding_ms_per_step": sum(stage_us.values()) * 1e-3,
    }
    if per_rank is not None:
        out["per_rank"] = per_rank
    m._drain_prefetch()
    del m, emb, ex, batches
    import gc
    gc.collect()
    torch.cuda.empty_cache()
    return out


# R/README.md:72-74 (DCN quick start) and R/samples/deepfm/deepfm_parquet.py:33-60
C1_SLOTS = [39884, 39043, 17289, 7420, 20263, 3, 7120, 1543, 39884, 39043, 17289, 7420, 20263, 3,
            7120, 1543, 63, 63, 39884, 39043, 17289, 7420, 20263, 3, 7120, 1543]
C2_SLOTS = [203931, 18598, 14092, 7012, 18977, 4, 6385, 1245, 49, 186213, 71328, 67288, 11, 2168,
            7338, 61, 4, 932, 15, 204515, 141526, 199433, 60919, 9137, 71, 34]


class _CycleReader:
    """batches already resident in HBM (the bench contract), handed out round robin"""

    def __init__(self, batches):
        self.b, self.i = batches, 0

    def next_batch(self, train: bool):
        self.i += 1
        return self.b[(self.i - 1) % len(self.b)]

    def has_eval(self):
        return False


def small_config_leg(cfg, steps, warmup, dev):
    """BASELINE configs[0] (c1: the README's DCN on its synthetic Parquet data, bs 1024) and
    configs[1] (c2: DeepFM, Criteo-Kaggle slot sizes, DistributedSlotSparseEmbeddingHash, D = 16,
    bs 16384) -- the reference's own scripts (R/README.md:59-150, R/samples/deepfm/
    deepfm_parquet.py) written against the `hugectr` surface of this repo; data generated by
    hugectr.tools.DataGenerator, read once through the Parquet reader and then served from HBM.
    One step = Model.train(): embedding forward, dense tower forward + backward, sparse Adam
    (Global) update, dense Adam step."""
    import shutil
    import tempfile
    import hugectr_amd.hugectr as hugectr
    c1 = cfg == "c1"
    slots = C1_SLOTS if c1 else C2_SLOTS
    B, D, nb = (1024, 16, 16) if c1 else (16384, 16, 8)
    i64 = not c1  # the README generates u32 keys (i64_input_key = False), the DeepFM sample i64
    tmp = tempfile.mkdtemp(prefix=f"hctr_bench_{cfg}_")
    try:
        hugectr.tools.DataGenerator(hugectr.tools.DataGeneratorParams(
            format=hugectr.DataReaderType_t.Parquet, label_dim=1, dense_dim=13, num_slot=26,
            i64_input_key=i64, source=os.path.join(tmp, "train", "_file_list.txt"),
            eval_source="", slot_size_array=slots, dist_type=hugectr.Distribution_t.PowerLaw,
            power_law_type=hugectr.PowerLaw_t.Short, num_files=1, eval_num_files=0,
            num_samples_per_file=B * nb, num_samples=B * nb, eval_num_samples=0)).generate()
        solver = hugectr.CreateSolver(max_eval_batches=1, batchsize_eval=B, batchsize=B, lr=0.001,
                                      vvgpu=[[0]], repeat_dataset=True, i64_input_key=i64)
        reader = hugectr.DataReaderParams(
            data_reader_type=hugectr.DataReaderType_t.Parquet,
            source=[os.path.join(tmp, "train", "_file_list.txt")], eval_source="",
            slot_size_array=slots, check_type=hugectr.Check_t.Non)
        optimizer = hugectr.CreateOptimizer(optimizer_type=hugectr.Optimizer_t.Adam,
                                            update_type=hugectr.Update_t.Global, beta1=0.9,
                                            beta2=0.999, epsilon=1e-7)
        m = hugectr.Model(solver, reader, optimizer)
        L, T = hugectr.DenseLayer, hugectr.Layer_t
        m.add(hugectr.Input(label_dim=1, label_name="label", dense_dim=13, dense_name="dense",
                            data_reader_sparse_param_array=[
                                hugectr.DataReaderSparseParam("data1", 1, True, 26)]))
        m.add(hugectr.SparseEmbedding(
            embedding_type=hugectr.Embedding_t.DistributedSlotSparseEmbeddingHash,
            workspace_size_per_gpu_in_mb=75 if c1 else 300, embedding_vec_size=D, combiner="sum",
            sparse_embedding_name="sparse_embedding1", bottom_name="data1", optimizer=optimizer))
        if c1:  # R/README.md:116-146
            m.add(L(layer_type=T.Reshape, bottom_names=["sparse_embedding1"], top_names=["reshape1"],
                    leading_dim=416))
            m.add(L(layer_type=T.Concat, bottom_names=["reshape1", "dense"], top_names=["concat1"]))
            m.add(L(layer_type=T.MultiCross, bottom_names=["concat1"], top_names=["multicross1"],
                    num_layers=6))
            m.add(L(layer_type=T.InnerProduct, bottom_names=["concat1"], top_names=["fc1"],
                    num_output=1024))
            m.add(L(layer_type=T.ReLU, bottom_names=["fc1"], top_names=["relu1"]))
            m.add(L(layer_type=T.Dropout, bottom_names=["relu1"], top_names=["dropout1"],
                    dropout_rate=0.5))
            m.add(L(layer_type=T.Concat, bottom_names=["dropout1", "multicross1"],
                    top_names=["concat2"]))
            m.add(L(layer_type=T.InnerProduct, bottom_names=["concat2"], top_names=["fc2"],
                    num_output=1))
            m.add(L(layer_type=T.BinaryCrossEntropyLoss, bottom_names=["fc2", "label"],
                    top_names=["loss"]))
        else:  # R/samples/deepfm/deepfm_parquet.py:111-300 with embedding_vec_size 16
            m.add(L(layer_type=T.Reshape, bottom_names=["sparse_embedding1"], top_names=["reshape1"],
                    leading_dim=D))
            m.add(L(layer_type=T.Slice, bottom_names=["reshape1"], top_names=["slice11", "slice12"],
                    ranges=[(0, D - 1), (D - 1, D)]))
            m.add(L(layer_type=T.Reshape, bottom_names=["slice11"], top_names=["reshape2"],
                    leading_dim=26 * (D - 1)))
            m.add(L(layer_type=T.Reshape, bottom_names=["slice12"], top_names=["reshape3"],
                    leading_dim=26))
            m.add(L(layer_type=T.WeightMultiply, bottom_names=["dense"],
                    top_names=["weight_multiply1"], weight_dims=[13, D - 1]))
            m.add(L(layer_type=T.WeightMultiply, bottom_names=["dense"],
                    top_names=["weight_multiply2"], weight_dims=[13, 1]))
            m.add(L(layer_type=T.Concat, bottom_names=["reshape2", "weight_multiply1"],
                    top_names=["concat1"]))
            prev = "concat1"
            for i in (1, 2, 3):
                m.add(L(layer_type=T.InnerProduct, bottom_names=[prev], top_names=[f"fc{i}"],
                        num_output=400))
                m.add(L(layer_type=T.ReLU, bottom_names=[f"fc{i}"], top_names=[f"relu{i}"]))
                m.add(L(layer_type=T.Dropout, bottom_names=[f"relu{i}"], top_names=[f"dropout{i}"],
                        dropout_rate=0.5))
                prev = f"dropout{i}"
            m.add(L(layer_type=T.InnerProduct, bottom_names=[prev], top_names=["fc4"], num_output=1))
            m.add(L(layer_type=T.FmOrder2, bottom_names=["concat1"], top_names=["fmorder2"],
                    out_dim=D - 1))
            m.add(L(layer_type=T.ReduceSum, bottom_names=["fmorder2"], top_names=["reducesum1"],
                    axis=1))
            m.add(L(layer_type=T.Concat, bottom_names=["reshape3", "weight_multiply2"],
                    top_names=["concat2"]))
            m.add(L(layer_type=T.ReduceSum, bottom_names=["concat2"], top_names=["reducesum2"],
                    axis=1))
            m.add(L(layer_type=T.Add, bottom_names=["fc4", "reducesum1", "reducesum2"],
                    top_names=["add"]))
            m.add(L(layer_type=T.BinaryCrossEntropyLoss, bottom_names=["add", "label"],
                    top_names=["loss"]))
        m.compile()
        batches = [m.reader.next_batch(True) for _ in range(nb)]
        m.reader = _CycleReader(batches)
    finally:
        shutil.rmtree(tmp, ignore_errors=True)
    h = list(m._emb.values())[0][2]
    for _ in range(warmup):
        m.train()
    torch.cuda.synchronize()
    h.profiling(True)
    t0 = time.perf_counter()
    for _ in range(steps):
        m.train()
    torch.cuda.synchronize()
    el = time.perf_counter() - t0
    prof = h.profile()
    h.profiling(False)
    m.check_overflow()
    K = 8 if i64 else 4
    nnz = B * 26
    alg = nnz * (K + 8 + D * 4) + nnz * D * 4
    pool_ms, pool_n = prof["gather_pool"]
    ach = alg / (pool_ms / max(pool_n, 1) * 1e-3) / 1e9 if pool_ms > 0 else 0.0
    return {
        "metric": "samples/sec, " + ("DCN README synthetic (BASELINE configs[0])" if c1 else
                                     "DeepFM Criteo-Kaggle shape, D=16 (BASELINE configs[1])"),
        "value": B * steps / el, "unit": "samples/s", "n_gpus": 1, "steps": steps,
        "warmup": warmup, "ms_per_step": el / steps * 1e3, "higher_is_better": True,
        "scaling": "weak", "vs_baseline": None,
        "dtype": "fp32 (reference default: fp32 tables, vectors, dense tower)",
        "data": "synthetic power-law alpha=1.3 (hugectr.tools.DataGenerator, PowerLaw_t.Short), "
                "one-hot, resident in HBM",
        "config": {"workload": ("BASELINE configs[0]: DCN, README synthetic slot sizes, "
                                "DistributedSlotSparseEmbeddingHash, D=16, bs=1024, Adam Global"
                                if c1 else
                                "BASELINE configs[1]: DeepFM, Criteo-Kaggle slot sizes, "
                                "DistributedSlotSparseEmbeddingHash, D=16, bs=16384, Adam Global"),
                   "surface": "hugectr_amd.hugectr Model.train()", "batch": B, "slots": 26,
                   # solver.use_cuda_graph (default on): dense tower replayed from a HIP graph
                   "hip_graph": getattr(m, "_graph", None) is not None,
                   "emb_dim": D, "table_rows_total": int(sum(slots)),
                   "max_vocabulary_size_per_gpu": h.get_max_vocabulary_size(),
                   "final_loss": m.get_current_loss()},
        "roofline": {"bound": "hbm", "kernel": "gather + pool (64-byte rows)",
                     "achieved": ach, "peak": HBM_PEAK_GBPS, "unit": "GB/s",
                     "frac": ach / HBM_PEAK_GBPS, "traffic": None,
                     "algorithmic_bytes_per_launch": alg, "launches": pool_n,
                     "avg_launch_us": pool_ms / max(pool_n, 1) * 1e3},
        "stage_us_per_step": {k: (v[0] / max(v[1], 1)) * 1e3 for k, v in prof.items()},
    }


def dlrm_ebc_model_leg(steps, warmup, dev, B=65536, alpha=1.1):
    """The main line's model (DLRM, Criteo-1TB tables, D = 128, SGD, mixed precision) written the
    way the reference's embedding_collection script writes it
    (R/test/embedding_collection_test/dgx_a100_one_hot.py:223-330) against the `hugectr` surface of
    this repo: RawAsync input (one u32 key per table), EmbeddingTableConfig /
    EmbeddingCollectionConfig.embedding_lookup(...).shard(...), bottom MLP, Interaction, top MLP,
    BCE -- one step = Model.train().  What a dropped-in script gets, next to the hand-driven main
    line."""
    import shutil
    import tempfile
    import hugectr_amd.hugectr as hugectr
    nb = 4
    tmp = tempfile.mkdtemp(prefix="hctr_bench_ebc_model_")
    try:
        rng = np.random.default_rng(77)
        a = np.zeros((B * nb, 1 + 13 + 26), dtype="<u4")
        keys = np.stack([powerlaw(rng, B * nb, v, alpha) for v in CRITEO_1TB], 1)
        a[:, 0] = (keys[:, 2] % 2).astype("<i4").view("<u4")
        a[:, 1:14] = rng.random((B * nb, 13), dtype=np.float32).view("<u4")
        a[:, 14:] = keys.astype("<u4")
        f = os.path.join(tmp, "train_data.bin")
        a.tofile(f)
        solver = hugectr.CreateSolver(max_eval_batches=1, batchsize_eval=B, batchsize=B, lr=0.5,
                                      vvgpu=[[0]], repeat_dataset=True, i64_input_key=False,
                                      use_mixed_precision=True, scaler=1024.0,
                                      use_embedding_collection=True)
        optimizer = hugectr.CreateOptimizer(optimizer_type=hugectr.Optimizer_t.SGD,
                                            update_type=hugectr.Update_t.Local, atomic_update=True)
        reader = hugectr.DataReaderParams(
            data_reader_type=hugectr.DataReaderType_t.RawAsync, source=[f], eval_source="",
            check_type=hugectr.Check_t.Non, num_samples=B * nb, eval_num_samples=0,
            slot_size_array=CRITEO_1TB,
            async_param=hugectr.AsyncParam(1, 4, 512000, 4, 512, True, hugectr.Alignment_t.Non,
                                           multi_hot_reader=True, is_dense_float=True))
        m = hugectr.Model(solver, reader, optimizer)
        m.add(hugectr.Input(label_dim=1, label_name="label", dense_dim=13, dense_name="dense",
                            data_reader_sparse_param_array=[
                                hugectr.DataReaderSparseParam(f"data{i}", 1, True, 1)
                                for i in range(26)]))
        tables = [hugectr.EmbeddingTableConfig(name=str(i), max_vocabulary_size=v, ev_size=128)
                  for i, v in enumerate(CRITEO_1TB)]
        ebc = hugectr.EmbeddingCollectionConfig(use_exclusive_keys=True)
        ebc.embedding_lookup(table_config=tables, bottom_name=[f"data{i}" for i in range(26)],
                             top_name="sparse_embedding", combiner=["concat"] * 26)
        names = [str(i) for i in range(26)]
        ebc.shard(shard_matrix=[names], shard_strategy=[("mp", names)])
        m.add(ebc)
        L, T, A = hugectr.DenseLayer, hugectr.Layer_t, hugectr.Activation_t
        m.add(L(layer_type=T.MLP, bottom_names=["dense"], top_names=["mlp1"],
                num_outputs=BOTTOM, act_type=A.Relu))
        m.add(L(layer_type=T.Reshape, bottom_names=["sparse_embedding"],
                top_names=["sparse_embedding1"], shape=[-1, 26, 128]))
        m.add(L(layer_type=T.Interaction, bottom_names=["mlp1", "sparse_embedding1"],
                top_names=["interaction1"]))
        m.add(L(layer_type=T.MLP, bottom_names=["interaction1"], top_names=["mlp2"],
                num_outputs=TOP, activations=[A.Relu] * 4 + [A.Non]))
        m.add(L(layer_type=T.BinaryCrossEntropyLoss, bottom_names=["mlp2", "label"],
                top_names=["loss"]))
        m.compile()
        batches = [m.reader.next_batch(True) for _ in range(nb)]
        m.reader = _CycleReader(batches)
    finally:
        shutil.rmtree(tmp, ignore_errors=True)
    for _ in range(warmup):
        m.train()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(steps):
        m.train()
    torch.cuda.synchronize()
    el = time.perf_counter() - t0
    m.check_overflow()
    return {"workload": "DLRM Criteo-1TB (the main line's model) as the reference's "
                        "embedding_collection script builds it, through hugectr.Model.train(): "
                        f"B={B}, D=128, SGD, use_mixed_precision (scaler 1024), static tables, "
                        "batches resident in HBM",
            "surface": "hugectr_amd.hugectr Model.train() + EmbeddingCollectionConfig",
            "value": B * steps / el, "unit": "samples/s", "ms_per_step": el / steps * 1e3,
            "steps": steps, "warmup": warmup, "final_loss": m.get_current_loss(),
            "direct_one_gpu_path": bool(m._ebc[0]["train"]._direct)}


def _timed_us(fn, it=20, warm=3):
    for _ in range(warm):
        fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(it):
        fn()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / it * 1e3


MFMA_PEAK_16 = 2500.0   # TFLOP/s dense, fp16 / bf16 (MI355X_MICROARCH.md)
MFMA_PEAK_F32 = 157.3   # TFLOP/s, f32-input matrix instructions


def interaction_leg(dev):
    """the dot-interaction layer alone (a19) -- B in {8192, 65536}, 26 embeddings +
    the bottom-MLP row, W = 128, N(0, 1) inputs (R/test/utest/layers/interaction_layer_test.cpp:96),
    fp16 and fp32, forward and backward.  HBM-bound at these shapes (27 x 128 inputs, 8 MFMAs per
    sample): bytes = B (n_ins W + W + n_ins (n_ins - 1) / 2 + 1) E forward, ~2x backward."""
    from hugectr_amd import _lib
    from hugectr_amd.layers import _DT
    lib, ptr, check, sp = _lib.lib, _lib.ptr, _lib.check, _lib.stream_ptr
    res = {}
    n, W = 26, 128
    n_ins = n + 1
    for B in (8192, 65536):
        for name, dt, E in (("fp16", torch.float16, 2), ("fp32", torch.float32, 4)):
            # the C ABI directly, launches back to back (through autograd the host's ~100 us per
            # call would be what is timed at B = 8192)
            mlp = torch.randn(B, W, device=dev).to(dt)
            emb = torch.randn(B, n, W, device=dev).to(dt)
            out = torch.empty((B, W + n_ins * (n_ins - 1) // 2 + 1), dtype=dt, device=dev)
            g = torch.randn(out.shape, device=dev).to(dt)
            mg, eg = torch.empty_like(mlp), torch.empty_like(emb)
            fwd = _timed_us(lambda: check(lib.hctr_interaction_fwd(
                B, n, W, ptr(mlp), ptr(emb), ptr(out), _DT[dt], sp())), it=50)
            bwd = _timed_us(lambda: check(lib.hctr_interaction_bwd(
                B, n, W, ptr(mlp), ptr(emb), ptr(g), ptr(mg), ptr(eg), _DT[dt], sp())), it=50)
            out_len = W + n_ins * (n_ins - 1) // 2 + 1
            fb_bytes = B * (n_ins * W + out_len) * E
            bw_bytes = B * (2 * n_ins * W + out_len) * E  # inputs + their gradients + dOut
            flops = B * 2 * n_ins * n_ins * W
            res[f"B{B}_{name}"] = {
                "forward_us": fwd, "backward_us": bwd,
                "roofline": {"bound": "hbm", "achieved": fb_bytes / fwd / 1e3, "peak": HBM_PEAK_GBPS,
                             "unit": "GB/s", "frac": fb_bytes / fwd / 1e3 / HBM_PEAK_GBPS,
                             "algorithmic_bytes": fb_bytes},
                "roofline_backward": {"bound": "hbm", "achieved": bw_bytes / bwd / 1e3,
                                      "peak": HBM_PEAK_GBPS, "unit": "GB/s",
                                      "frac": bw_bytes / bwd / 1e3 / HBM_PEAK_GBPS,
                                      "algorithmic_bytes": bw_bytes},
                "mfma_tflops_forward": flops / fwd / 1e6,
                "mfma_frac_of_peak": flops / fwd / 1e6 / (MFMA_PEAK_16 if E == 2 else MFMA_PEAK_F32)}
            del mlp, emb, out, g, mg, eg
    return {"workload": "SURVEY I1: InteractionLayer alone, n_emb=26, W=128, N(0,1) inputs",
            "kernels": "interaction_fwd16 / bwd16 (v_mfma_f32_32x32x16_f16), fp32: 3x bf16-split MFMA",
            "cases": res}


def cross_leg(dev):
    """MultiCrossLayer alone (a20).  v1 (projection_dim 0): B=1024, w=429, 6 layers
    (the README DCN shape; B=16384 beside it, where the launch is no longer the cost) -- HBM-bound,
    B w E 4 bytes per layer against the fused kernel's own B w 4 (1 + L).  v2: B=8192,
    w=3456, p=512, 3 layers (the MLPerf DCNv2 tower, R/samples/dlrm/train.py:425-441) -- MFMA-bound,
    4 B w p flops per layer forward, twice that backward."""
    from hugectr_amd.layers import MultiCrossLayer
    res = {}
    for B in (1024, 16384):
        w, L = 429, 6
        layer = MultiCrossLayer(w, L, 0).to(dev)
        x = torch.randn(B, w, device=dev, requires_grad=True)
        out = layer(x)
        g = torch.randn_like(out)
        fwd = _timed_us(lambda: layer(x), it=50)

        def both():
            o = layer(x)
            o.backward(g)
            x.grad = None
            layer.zero_grad(set_to_none=True)
        fb = _timed_us(both, it=50)
        alg = L * B * w * 4 * 4
        own = B * w * 4 * (1 + L)
        # (the fraction is quoted on the fused kernel's own bytes -- x0 read once, L outputs
        #  written -- not on SURVEY's 4 arrays per layer, which a kernel that keeps the row in
        #  registers never moves: that figure exceeds 1 at B = 16384)
        res[f"v1_B{B}"] = {"forward_us": fwd, "forward_backward_us": fb,
                           "roofline": {"bound": "hbm", "achieved": own / fwd / 1e3,
                                        "peak": HBM_PEAK_GBPS, "unit": "GB/s",
                                        "frac": own / fwd / 1e3 / HBM_PEAK_GBPS,
                                        "kernel_bytes": own, "survey_bytes_4_arrays_per_layer": alg}}
    B, w, pdim, L = 8192, 3456, 512, 3
    for name, dt, peak in (("fp16", torch.float16, MFMA_PEAK_16), ("fp32", torch.float32, MFMA_PEAK_F32)):
        layer = MultiCrossLayer(w, L, pdim).to(dev)
        x = torch.randn(B, w, device=dev).to(dt).requires_grad_(True)
        out = layer(x)
        g = torch.randn_like(out)
        fwd = _timed_us(lambda: layer(x), it=10)

        def both2():
            o = layer(x)
            o.backward(g)
            x.grad = None
            layer.zero_grad(set_to_none=True)
        fb = _timed_us(both2, it=10)
        flops = L * 4 * B * w * pdim
        res[f"v2_B{B}_{name}"] = {
            "forward_us": fwd, "forward_backward_us": fb, "activation_dtype": name,
            "roofline": {"bound": "mfma", "achieved": flops / fwd / 1e6, "peak": peak,
                         "unit": "TFLOP/s", "frac": flops / fwd / 1e6 / peak, "flops": flops},
            "roofline_forward_backward": {"bound": "mfma", "achieved": 3 * flops / fb / 1e6,
                                          "peak": peak, "unit": "TFLOP/s",
                                          "frac": 3 * flops / fb / 1e6 / peak, "flops": 3 * flops}}
        del layer, x, out, g
    return {"workload": "SURVEY X1: MultiCrossLayer alone", "cases": res}


def dcnv2_model_leg(steps, warmup, dev, B=65536, alpha=1.1):
    """The model the reference ships for MLPerf (R/samples/dlrm/train.py:406-470): one multi-hot
    input per table (214 keys per sample), an embedding_collection over the 26 MLPerf tables
    (sum), bottom MLP, concat, MultiCross with projection_dim 512 x 3 layers, top MLP, BCE --
    through hugectr.Model.train(), mixed precision, SGD.  One step = Model.train()."""
    import shutil
    import tempfile
    import hugectr_amd.hugectr as hugectr
    nb = 3
    hot, sizes = MLPERF_HOTNESS, MLPERF_TABLES
    tmp = tempfile.mkdtemp(prefix="hctr_bench_dcnv2_")
    try:
        rng = np.random.default_rng(78)
        nk = sum(hot)
        a = np.zeros((B * nb, 1 + 13 + nk), dtype="<u4")
        col = 14
        for v, h in zip(sizes, hot):
            a[:, col:col + h] = powerlaw(rng, B * nb * h, v, alpha).reshape(B * nb, h).astype("<u4")
            col += h
        a[:, 0] = (a[:, 16] % 2).astype("<i4").view("<u4")
        a[:, 1:14] = rng.random((B * nb, 13), dtype=np.float32).view("<u4")
        f = os.path.join(tmp, "train_data.bin")
        a.tofile(f)
        del a
        solver = hugectr.CreateSolver(max_eval_batches=1, batchsize_eval=B, batchsize=B, lr=0.005,
                                      vvgpu=[[0]], repeat_dataset=True, i64_input_key=False,
                                      use_mixed_precision=True, scaler=1024.0,
                                      use_embedding_collection=True)
        optimizer = hugectr.CreateOptimizer(optimizer_type=hugectr.Optimizer_t.SGD,
                                            update_type=hugectr.Update_t.Local, atomic_update=True)
        reader = hugectr.DataReaderParams(
            data_reader_type=hugectr.DataReaderType_t.RawAsync, source=[f], eval_source="",
            check_type=hugectr.Check_t.Non, num_samples=B * nb, eval_num_samples=0,
            slot_size_array=sizes,
            async_param=hugectr.AsyncParam(1, 4, 512000, 4, 512, True, hugectr.Alignment_t.Non,
                                           multi_hot_reader=True, is_dense_float=True))
        m = hugectr.Model(solver, reader, optimizer)
        m.add(hugectr.Input(label_dim=1, label_name="label", dense_dim=13, dense_name="dense",
                            data_reader_sparse_param_array=[
                                hugectr.DataReaderSparseParam(f"data{i}", hot[i], True, 1)
                                for i in range(26)]))
        tables = [hugectr.EmbeddingTableConfig(name=str(i), max_vocabulary_size=v, ev_size=128)
                  for i, v in enumerate(sizes)]
        ebc = hugectr.EmbeddingCollectionConfig(use_exclusive_keys=True)
        ebc.embedding_lookup(table_config=tables, bottom_name=[f"data{i}" for i in range(26)],
                             top_name="sparse_embedding", combiner=["sum"] * 26)
        names = [str(i) for i in range(26)]
        ebc.shard(shard_matrix=[names], shard_strategy=[("mp", names)])
        m.add(ebc)
        L, T, A = hugectr.DenseLayer, hugectr.Layer_t, hugectr.Activation_t
        m.add(L(layer_type=T.MLP, bottom_names=["dense"], top_names=["mlp1"],
                num_outputs=BOTTOM, act_type=A.Relu))
        m.add(L(layer_type=T.Concat, bottom_names=["sparse_embedding", "mlp1"],
                top_names=["concat1"]))
        m.add(L(layer_type=T.MultiCross, bottom_names=["concat1"], top_names=["interaction1"],
                projection_dim=512, num_layers=3))
        m.add(L(layer_type=T.MLP, bottom_names=["interaction1"], top_names=["mlp2"],
                num_outputs=TOP, activations=[A.Relu] * 4 + [A.Non]))
        m.add(L(layer_type=T.BinaryCrossEntropyLoss, bottom_names=["mlp2", "label"],
                top_names=["loss"]))
        m.compile()
        batches = [m.reader.next_batch(True) for _ in range(nb)]
        m.reader = _CycleReader(batches)
    finally:
        shutil.rmtree(tmp, ignore_errors=True)
    for _ in range(warmup):
        m.train()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(steps):
        m.train()
    torch.cuda.synchronize()
    el = time.perf_counter() - t0
    m.check_overflow()
    w = 26 * 128 + 128
    cross_flops = 3 * 3 * 4 * B * w * 512  # forward + backward of the three cross layers
    return {"workload": "MLPerf DLRM-DCNv2 as R/samples/dlrm/train.py:406-470 builds it, through "
                        f"hugectr.Model.train(): B={B}, 26 tables ({sum(sizes)} rows), 214 keys / "
                        "sample (multi-hot, sum), D=128, bottom MLP 512-256-128, MultiCross "
                        "projection 512 x 3 layers over 3456 columns, top MLP 1024-1024-512-256-1, "
                        "SGD, use_mixed_precision (scaler 1024), batches resident in HBM",
            "surface": "hugectr_amd.hugectr Model.train() + EmbeddingCollectionConfig + MultiCross",
            "value": B * steps / el, "unit": "samples/s", "ms_per_step": el / steps * 1e3,
            "steps": steps, "warmup": warmup, "final_loss": m.get_current_loss(),
            "keys_per_batch": B * sum(hot), "cross_flops_per_step": cross_flops,
            "direct_one_gpu_path": bool(m._ebc[0]["train"]._direct)}


# R/samples/dlrm/train.py:29-83 (MLPerf DLRM-DCNv2): tables capped at 40 M rows, keys per sample
MLPERF_TABLES = [40000000, 39060, 17295, 7424, 20265, 3, 7122, 1543, 63, 40000000, 3067956, 405282,
                 10, 2209, 11938, 155, 4, 976, 14, 40000000, 40000000, 40000000, 590152, 12973,
                 108, 36]
MLPERF_HOTNESS = [3, 2, 1, 2, 6, 1, 1, 1, 1, 7, 3, 8, 1, 6, 9, 5, 1, 1, 1, 12, 100, 27, 10, 3, 1, 1]


def dynamic_optimizer_leg(steps, warmup, dev, alpha):
    """stateful optimizers on dynamic tables (embedding::DynamicEmbeddingTable::update,
    R/HugeCTR/embedding_storage/dynamic_embedding.cu:227-317): the step on the flat row store with
    the state at the weights' row numbers (one probe per key, the forward's) beside the
    reference's flow (HCTR_DYNAMIC_FLAT=0: unique keys -> wgrad -> probe of the weights + inserting
    probe of a state table -> optimizer kernel), same collection and batches as
    ebc_dynamic_multi_hot"""
    import gc
    r = {}
    prev = os.environ.get("HCTR_DYNAMIC_FLAT")
    try:
        for opt in ("adagrad", "adam"):
            for flat in ("1", "0"):
                os.environ["HCTR_DYNAMIC_FLAT"] = flat
                gc.collect()
                torch.cuda.empty_cache()
                name = f"{opt}_{'flat_row_store' if flat == '1' else 'unique_key_flow'}"
                free0 = torch.cuda.mem_get_info()[0]
                try:
                    leg = ebc_leg("multi_hot", steps, warmup, dev, alpha, dynamic=True,
                                  optimizer=opt)
                    r[name] = {k: leg[k] for k in ("forward_us", "backward_update_us",
                                                   "forward_backward_update_us", "value", "unit")}
                    del leg
                except Exception as e:
                    r[name] = {"error": repr(e)}
                r[name]["hbm_free_gb_before"] = free0 / 1e9
    finally:
        if prev is None:
            os.environ.pop("HCTR_DYNAMIC_FLAT", None)
        else:
            os.environ["HCTR_DYNAMIC_FLAT"] = prev
    r["workload"] = ("embedding_collection, MLPerf DLRM-DCNv2 tables and hotness on DYNAMIC hash "
                     "tables, B=65536, D=128, fp16 output; AdaGrad and Adam")
    return r


def c5_model_leg(steps, warmup, dev, B=16384):
    """BASELINE configs[4]: Wide & Deep + MMoE, multi-hot, embedding_collection API on DYNAMIC hash
    tables (max_vocabulary_size = -1), fp16 dense tower -- through hugectr.Model.train().  The graph
    is the reference's MMoE sample (R/samples/mmoe/mmoe_parquet.py: 3 experts 256-128, two softmax
    gates, two towers, two BinaryCrossEntropyLoss weighted 0.5 / 0.5) fed by an
    EmbeddingCollectionConfig as R/samples/wdl and R/samples/ftrl/dlrm_train_ftrl.py:222-245 write
    it -- deep tables ev = 16 (hotness 1..5) into the experts and gates, wide tables ev = 1 whose sum
    joins tower A's logit (R/samples/wdl/wdl_1gpu.py's wide branch) -- AdaGrad, Parquet input read
    once and served from HBM.  One step = Model.train()."""
    import shutil
    import tempfile
    import hugectr_amd.hugectr as hugectr
    nb = 6
    sizes = [73622, 91, 17, 1425, 3, 24, 15, 5, 10, 2, 3, 6, 8, 133, 114, 1675, 6, 6, 51, 38, 8, 47, 10,
             9, 10, 3, 4, 7, 5, 2, 52, 9]  # the census tables of the MMoE sample
    hot = [5, 3, 1, 2] + [1] * 28
    wide, deep = [0, 3, 15], list(range(32))
    tmp = tempfile.mkdtemp(prefix="hctr_bench_c5_")
    try:
        hugectr.tools.DataGenerator(hugectr.tools.DataGeneratorParams(
            format=hugectr.DataReaderType_t.Parquet, label_dim=2, dense_dim=0, num_slot=32,
            i64_input_key=True, source=os.path.join(tmp, "train", "_file_list.txt"),
            eval_source="", slot_size_array=sizes, nnz_array=hot,
            dist_type=hugectr.Distribution_t.PowerLaw, power_law_type=hugectr.PowerLaw_t.Short,
            num_files=1, eval_num_files=0, num_samples_per_file=B * nb, num_samples=B * nb,
            eval_num_samples=0)).generate()
        solver = hugectr.CreateSolver(max_eval_batches=1, batchsize_eval=B, batchsize=B, lr=0.01,
                                      vvgpu=[[0]], repeat_dataset=True, i64_input_key=True,
                                      use_mixed_precision=True, scaler=1024.0,
                                      use_embedding_collection=True)
        reader = hugectr.DataReaderParams(
            data_reader_type=hugectr.DataReaderType_t.Parquet,
            source=[os.path.join(tmp, "train", "_file_list.txt")], eval_source="",
            slot_size_array=sizes, check_type=hugectr.Check_t.Non)
        optimizer = hugectr.CreateOptimizer(optimizer_type=hugectr.Optimizer_t.AdaGrad,
                                            update_type=hugectr.Update_t.Global)
        m = hugectr.Model(solver, reader, optimizer)
        L, T = hugectr.DenseLayer, hugectr.Layer_t
        m.add(hugectr.Input(label_dims=[1, 1], label_names=["labelA", "labelB"], dense_dim=0,
                            dense_name="dense",
                            data_reader_sparse_param_array=[
                                hugectr.DataReaderSparseParam(f"data{i}", hot[i], True, 1)
                                for i in range(32)]))
        ebc = hugectr.EmbeddingCollectionConfig()
        for i in wide:
            ebc.embedding_lookup(table_config=hugectr.EmbeddingTableConfig(f"w{i}", -1, 1),
                                 bottom_name=f"data{i}", top_name=f"wide{i}", combiner="sum")
        for i in deep:
            ebc.embedding_lookup(table_config=hugectr.EmbeddingTableConfig(f"d{i}", -1, 16),
                                 bottom_name=f"data{i}", top_name=f"deep{i}", combiner="sum")
        names = [f"w{i}" for i in wide] + [f"d{i}" for i in deep]
        ebc.shard(shard_matrix=[names], shard_strategy=[("mp", names)])
        m.add(ebc)
        m.add(L(layer_type=T.Concat, bottom_names=[f"deep{i}" for i in deep], top_names=["emb"]))
        m.add(L(layer_type=T.Slice, bottom_names=["emb"],
                top_names=["e0_in", "e1_in", "e2_in", "gateA_in", "gateB_in"],
                ranges=[(0, 512)] * 5))
        for e in range(3):
            m.add(L(layer_type=T.MLP, bottom_names=[f"e{e}_in"], top_names=[f"e{e}_out"],
                    num_outputs=[256, 128], act_type=hugectr.Activation_t.Relu))
            m.add(L(layer_type=T.Slice, bottom_names=[f"e{e}_out"],
                    top_names=[f"e{e}_out_A", f"e{e}_out_B"], ranges=[(0, 128), (0, 128)]))
        for t in "AB":
            m.add(L(layer_type=T.InnerProduct, bottom_names=[f"gate{t}_in"], top_names=[f"g{t}_dense"],
                    num_output=3))
            m.add(L(layer_type=T.Softmax, bottom_names=[f"g{t}_dense"], top_names=[f"g{t}_softmax"]))
            m.add(L(layer_type=T.Slice, bottom_names=[f"g{t}_softmax"],
                    top_names=[f"g{t}_e0", f"g{t}_e1", f"g{t}_e2"], ranges=[(0, 1), (1, 2), (2, 3)]))
            for e in range(3):
                m.add(L(layer_type=T.Scale, bottom_names=[f"g{t}_e{e}"],
                        top_names=[f"g{t}_e{e}_scaled"], axis=0, factor=128))
                m.add(L(layer_type=T.ElementwiseMultiply,
                        bottom_names=[f"e{e}_out_{t}", f"g{t}_e{e}_scaled"],
                        top_names=[f"e{e}_{t}_gated"]))
            m.add(L(layer_type=T.Add, bottom_names=[f"e{e}_{t}_gated" for e in range(3)],
                    top_names=[f"tower_{t}_input"]))
            m.add(L(layer_type=T.MLP, bottom_names=[f"tower_{t}_input"], top_names=[f"{t}_fc2"],
                    num_outputs=[64, 1],
                    activations=[hugectr.Activation_t.Relu, hugectr.Activation_t.Non]))
        m.add(L(layer_type=T.Concat, bottom_names=[f"wide{i}" for i in wide], top_names=["wide"]))
        m.add(L(layer_type=T.ReduceSum, bottom_names=["wide"], top_names=["wide_sum"], axis=1))
        m.add(L(layer_type=T.Add, bottom_names=["A_fc2", "wide_sum"], top_names=["A_logit"]))
        m.add(L(layer_type=T.BinaryCrossEntropyLoss, bottom_names=["A_logit", "labelA"],
                top_names=["lossA"]))
        m.add(L(layer_type=T.BinaryCrossEntropyLoss, bottom_names=["B_fc2", "labelB"],
                top_names=["lossB"]))
        m.compile(loss_names=["labelA", "labelB"], loss_weights=[0.5, 0.5])
        # (as the other legs: read once through the Parquet reader, then served from HBM)
        batches = [m.reader.next_batch(True) for _ in range(nb)]
        m.reader = _CycleReader(batches)
        for _ in range(warmup):
            m.train()
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(steps):
            m.train()
        torch.cuda.synchronize()
        el = time.perf_counter() - t0
        nnz = B * sum(hot[i] for i in deep) + B * sum(hot[i] for i in wide)
        held = sum(rt["train"].det.size() for rt in m._ebc)
        return {
            "workload": "BASELINE configs[4]: Wide & Deep + MMoE (3 experts, 2 gates, 2 tasks), "
                        f"embedding_collection over {len(deep)} deep (ev 16) + {len(wide)} wide (ev 1) "
                        f"DYNAMIC hash tables, multi-hot ({nnz // B} keys per sample), mixed "
                        f"precision (fp16 tower), AdaGrad, bs {B}, Model.train()",
            "ms_per_step": el / steps * 1e3, "value": B * steps / el, "unit": "samples/s",
            "keys_per_step": nnz, "keys_held_by_the_dynamic_tables": held,
            "loss": m.get_current_loss(),
        }
    finally:
        shutil.rmtree(tmp, ignore_errors=True)


def ebc_leg(kind, steps, warmup, dev, alpha=1.1, B=65536, D=128, dynamic=False, optimizer="sgd"):
    """embedding_collection on one GPU (the reference's current-generation path):
    forward and backward + update of the collection alone, through EmbeddingCollection.forward /
    backward_and_update.  kind = "one_hot": Criteo-1TB tables, one key per table (the shape of the
    main line, R/test/embedding_collection_test/dgx_a100_one_hot.py); "multi_hot": the MLPerf
    DLRM-DCNv2 tables and hotness (214 keys per sample)."""
    from hugectr_amd import _lib
    from hugectr_amd.embedding_collection import (EmbeddingCollection, EmbeddingCollectionConfig,
                                                  EmbeddingTableConfig)
    sizes = CRITEO_1TB if kind == "one_hot" else MLPERF_TABLES
    hot = [1] * 26 if kind == "one_hot" else MLPERF_HOTNESS
    cfg = EmbeddingCollectionConfig()
    tabs = [EmbeddingTableConfig(f"t{i}", v, D) for i, v in enumerate(sizes)]
    cfg.embedding_lookup(tabs, [f"b{i}" for i in range(26)], "sparse_embedding", ["sum"] * 26)
    kw = {}
    if dynamic:
        # BASELINE configs[4]'s embedding half: the same collection on DYNAMIC hash tables
        # (EmbeddingTableConfig(max_vocabulary_size=-1), R/HugeCTR/embedding_storage/
        # dynamic_embedding.cu:130-330): keys are inserted on first sight, rows found by probing
        kw = dict(storage="dynamic", init_capacity=1 << 22)
    opt_code = {"sgd": _lib.OPT_SGD, "adagrad": _lib.OPT_ADAGRAD, "adam": _lib.OPT_ADAM,
                "momentum": _lib.OPT_MOMENTUM_SGD}[optimizer]
    ebc = EmbeddingCollection(cfg, B, lr=0.01, optimizer=opt_code, scaler=1024.0,
                              out_dtype=torch.float16, batch_major=True, max_hotness=max(hot),
                              hotness=hot, **kw)
    g = torch.Generator(device=dev)
    g.manual_seed(99)
    batches = []
    nbat = 2 if not dynamic else 2 + warmup  # (dynamic tables: the inserts happen in warm-up)
    for _ in range(nbat):  # feature-major CSR: bucket = table * B + sample, hot[t] keys each
        ks = []
        for v, h in zip(sizes, hot):
            u = torch.rand(B * h, device=dev, generator=g, dtype=torch.float32).double()
            a = 1.0 - alpha
            y = ((float(v) ** a - 1.0) * u + 1.0) ** (1.0 / a)
            ks.append((torch.round(y) - 1).clamp_(0, v - 1).to(torch.int64))
        br = torch.zeros(26 * B + 1, dtype=torch.int64, device=dev)
        torch.cumsum(torch.tensor(hot, device=dev).repeat_interleave(B), 0, out=br[1:])
        batches.append((torch.cat(ks), br))
    out = ebc.forward(*batches[0])
    grad = (torch.randn(out.shape, device=dev) * 1e-3).to(out.dtype)

    def timed(fn):
        # (dynamic tables: every batch of the cycle is met once before the clock starts -- its
        #  unseen keys are inserted, the table grows -- so that the timed calls are steady ones)
        for i in range(max(warmup, nbat) if dynamic else warmup):
            fn(i)
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for i in range(steps):
            fn(i)
        e1.record()
        torch.cuda.synchronize()
        return e0.elapsed_time(e1) / steps * 1e3

    fwd_us = timed(lambda i: ebc.forward(*batches[i % nbat]))

    def train(i):
        ebc.forward(*batches[i % nbat])
        ebc.backward_and_update(grad)
    both_us = timed(train)
    nnz = B * sum(hot)
    pmc, pmc_src, pmc_stale = None, None, None
    pmc_path = _counter_file("pmc_hbm_traffic_ebc.json")
    if not os.path.exists(pmc_path):
        pmc_path = os.path.join(ROOT, "profiles", "r4_pmc_hbm_traffic_ebc.json")
    if os.path.exists(pmc_path) and B == 65536 and D == 128 and abs(alpha - 1.1) < 1e-9:
        try:  # the gather kernel of this leg (rocprofv3 --pmc passes, tools/measure_round.sh)
            j = json.load(open(pmc_path))
            # (dynamic tables of one dimension are one flat row store: the static gather runs on them)
            kern = ("pool_ptrs_vec4_kernel" if dynamic and not ebc._dyn_flat else
                    "pool_flat_kernel" if kind == "multi_hot" else "pool_vec4_kernel")
            pmc_src = f"{os.path.relpath(pmc_path, ROOT)} [{kern}] @ {j.get('commit', '?')}"
            pmc_stale = j.get("csrc_hash") != _csrc_hash()  # (quoted for these sources only)
            if not pmc_stale:
                pmc = j["kernels"][kern]["hbm_bytes_per_launch"]
        except Exception:
            pmc = None
    # dynamic tables: + the 16-byte hash probe per key in place of the static index arithmetic
    alg = nnz * (8 + 8 + D * 4) + B * 26 * D * 2 + (nnz * 16 if dynamic else 0)
    # compulsory bytes: every DISTINCT row of the batch once (what no cache can remove).  With 214
    # keys per sample SURVEY's duplicates-counted bytes exceed what 8 TB/s can carry in the
    # measured time (round 4 quoted 1.2 "of peak"): for these legs `frac` is the compulsory
    # fraction, the counter fraction stands beside it and the duplicates-counted figure is a note
    ks0 = batches[0][0]
    off0 = torch.tensor(np.concatenate([[0], np.cumsum(sizes)[:-1]]), device=dev).repeat_interleave(
        torch.tensor([B * h for h in hot], device=dev))
    distinct = int(torch.unique(ks0 + off0).numel())
    comp = nnz * (8 + 8) + distinct * D * 4 + B * 26 * D * 2 + (nnz * 16 if dynamic else 0)
    ach = alg / (fwd_us * 1e-6) / 1e9
    ach_comp = comp / (fwd_us * 1e-6) / 1e9
    return {
        "workload": ("embedding_collection, Criteo-1TB tables, one-hot" if kind == "one_hot" else
                     "embedding_collection, MLPerf DLRM-DCNv2 tables and hotness (214 keys / "
                     "sample), R/samples/dlrm/train.py:29-83") +
                    (" on DYNAMIC hash tables (BASELINE configs[4], embedding half)" if dynamic
                     else "") +
                    f", B={B}, D={D}, fp16 output [B][26][D], {optimizer}, power-law alpha={alpha}; the "
                    "collection alone (no dense tower)",
        "direct_one_gpu_path": bool(ebc._direct), "keys_per_batch": nnz,
        "table_rows_total": int(sum(sizes)),
        "forward_us": fwd_us, "forward_backward_update_us": both_us,
        "backward_update_us": both_us - fwd_us,
        "value": B / (both_us * 1e-6), "unit": "samples/s (embedding path only)",
        "roofline": {"bound": "hbm", "kernel": "whole forward (key -> row pass + gather/pool): a "
                                               "lower bound on the gather kernel's own rate",
                     "achieved": ach_comp, "peak": HBM_PEAK_GBPS, "unit": "GB/s",
                     "frac": ach_comp / HBM_PEAK_GBPS,
                     "bytes": "compulsory (every distinct row once)", "compulsory_bytes": comp,
                     "distinct_rows": distinct,
                     "frac_traffic": (pmc / (fwd_us * 1e-6) / 1e9 / HBM_PEAK_GBPS) if pmc else None,
                     "frac_duplicates_counted": ach / HBM_PEAK_GBPS,
                     "traffic": pmc, "traffic_source": pmc_src,
                     "traffic_stale": pmc_stale,
                     # duplicates counted: power-law keys repeat rows, which L2 /
                     # Infinity Cache serve -- `traffic` (PMC counters of the gather kernel) over
                     # the forward's time is what crossed the HBM interface
                     "hbm_traffic_gbps": (pmc / (fwd_us * 1e-6) / 1e9) if pmc else None,
                     "algorithmic_bytes_per_launch": alg},
    }


def tiered_leg(steps, warmup, dev, alpha=1.1, rows=20_000_000, D=128, n=1 << 20, tables=4):
    """BASELINE configs[3] (4 tables with a 10 B-row KEY SPACE each, host-HBM tiered, power-law
    keys) at the size one box's host memory takes: arbitrary int64 keys -- table * 10^10 + a key
    scattered over [0, 10^10) -- resolve through the device index of hctr_uvm_* to rows of a
    [rows, D] fp32 store in pinned host memory, handed out on first touch; the hot rows sit in the
    HBM cache (R/gpu_cache/include/nv_gpu_cache.hpp, uvm_table.hpp:127-174).  A NEW batch every
    call.  lookup = index probe + cache Query + miss fill straight out of host memory + Replace
    (write-back: a dirty victim goes home first); update = per-row gradient sums + SGD on the
    distinct rows, in the cache for the rows that are cached.  Bound by the host link (PCIe 5 x16:
    64 GB/s per direction), not by HBM."""
    from hugectr_amd.cache import UvmEmbedding
    cached = 1 << 22  # 4.2 M cached rows = 2.1 GB of HBM for a 10 GB store
    te = UvmEmbedding(cached, rows, D, n, lr=0.01)
    te.table.host[:] = 0.01
    g = torch.Generator(device=dev)
    g.manual_seed(5)
    per_table = rows // tables  # distinct keys a table can come to hold
    SPACE = 10**10

    def draw():
        ks = []
        for t in range(tables):
            u = torch.rand(n // tables, device=dev, generator=g, dtype=torch.float32).double()
            e = 1.0 - alpha
            y = ((float(per_table) ** e - 1.0) * u + 1.0) ** (1.0 / e) if alpha > 0 else u * per_table + 1
            rank = (torch.round(y) - 1).clamp_(0, per_table - 1).to(torch.int64)
            # the rank-th most frequent key of table t lies anywhere in the table's 10^10 keys
            ks.append(t * SPACE + (rank * 2654435761 + 12345) % SPACE)
        return torch.stack(ks, 1).reshape(-1).contiguous()
    nb = warmup + steps
    batches = [draw() for _ in range(2 * nb + 8)]
    grad = torch.randn((n, D), device=dev) * 1e-3
    for k in batches[:8]:  # the cache settles at the stream's hit rate
        te.forward(k)
    miss = []

    def timed(fn, first):
        for i in range(warmup):
            fn(batches[first + i])
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for i in range(steps):
            fn(batches[first + warmup + i])
            miss.append(te.table._miss.clone())
        e1.record()
        torch.cuda.synchronize()
        return e0.elapsed_time(e1) / steps * 1e3
    look_us = timed(lambda k: te.forward(k), 8)
    miss_rate = float(torch.stack(miss).double().mean()) / n
    miss.clear()

    def train(k):
        te.forward(k)
        te.backward_update(grad)
    both_us = timed(train, 8 + nb)
    miss_train = float(torch.stack(miss).double().mean()) / n
    te.table.check_overflow()
    keys_held = te.table.size()
    t0 = time.perf_counter()
    te.table.flush()
    flush_ms = (time.perf_counter() - t0) * 1e3
    link_bytes = miss_rate * n * D * 4
    LINK = 64.0
    return {
        "workload": f"tiered table of arbitrary keys (BASELINE configs[3] in miniature): {tables} "
                    f"tables x 10^10 key space, {rows} x {D} fp32 rows ({rows * D * 4 / 2**30:.1f} GiB) "
                    f"of pinned host memory handed out on first touch, {cached} cached rows in HBM, "
                    f"{n} one-hot power-law (alpha={alpha}) keys per call, a new batch every call",
        "lookup_us": look_us, "lookup_update_us": both_us, "update_us": both_us - look_us,
        "miss_rate": miss_rate, "miss_rate_while_training": miss_train, "keys_held": keys_held,
        "flush_ms_at_the_end": flush_ms, "value": n / (both_us * 1e-6),
        "unit": "keys/s (lookup + write-back SGD)",
        "roofline": {"bound": "host link (PCIe 5 x16, per direction)", "peak": LINK,
                     "unit": "GB/s", "achieved": link_bytes / (look_us * 1e-6) / 1e9,
                     "frac": link_bytes / (look_us * 1e-6) / 1e9 / LINK,
                     "algorithmic_bytes_per_launch": link_bytes, "traffic": None,
                     "note": "bytes = missed rows x D x 4 crossing the link during the lookup (as "
                             "many dirty victims travel the other way while training); the cache "
                             "hits (HBM) ride along in the same time"},
    }


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=30)
    ap.add_argument("--warmup", type=int, default=8)
    ap.add_argument("--batch", type=int, default=65536,
                    help="BASELINE config 3's 65536: per GPU on the main (weak-scaling) line, "
                         "global on the strong-scaling line")
    ap.add_argument("--scaling", default="both", choices=["both", "weak", "strong"],
                    help="N > 1: weak = 65536 samples per GPU (the main line), strong = global "
                         "batch 65536 (the reference's `batchsize` is global, "
                         "solver_wrapper.hpp:127-150); both = the weak line with the strong one "
                         "under the key `strong` of the same JSON object")
    ap.add_argument("--exchange", default=None, choices=["auto", "rows", "unique", "unique16"],
                    help="multi-GPU payload of the embedding exchange (sets HCTR_EXCHANGE, read by "
                         "hugectr.Model): rows = one pooled vector / gradient per (sample, slot) "
                         "as the reference; unique = every distinct row once per destination + "
                         "per-row gradient sums (hugectr_amd/unique_exchange.py); unique16 = the "
                         "same with 16-bit sums on the wire; auto (default) = Model.train() times "
                         "rows and unique over its first steps and keeps the faster")
    ap.add_argument("--no-overlap", action="store_true",
                    help="train_intra/inter_iteration_overlap = False: blocking collectives in line")
    ap.add_argument("--alpha", type=float, default=1.1, help="power-law exponent; 0 = uniform")
    ap.add_argument("--dim", type=int, default=128)
    ap.add_argument("--table-scale", type=float, default=1.0)
    ap.add_argument("--nbatches", type=int, default=0,
                    help="batches resident in HBM; 0 = one per step (warm-up included): every "
                         "step then meets keys it has not seen (inserts live), as real data does")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--precision", default="fp16", choices=["fp16", "fp32"],
                    help="precision of the MAIN line.  fp16 = the reference's mixed precision "
                         "(use_mixed_precision=True, scaler=1024: fp16 pooled vectors / top "
                         "gradients / dense GEMMs, loss scaling; the MLPerf DLRM configuration of "
                         "the reference), fp32 = the reference's default (everything fp32).  "
                         "Tables, pooling accumulation and the sparse optimizer are fp32 in both.")
    ap.add_argument("--extra", default="auto",
                    choices=["auto", "none", "all", "ebc", "model", "uniform", "next", "dense", "dcnv2", "dynopt", "tiered", "c5"],
                    help="extra legs appended to the JSON line under `extra` (1 GPU only): the "
                         "other precision on the same workload, `uniform_big_tables` (no key "
                         "repeats: the discriminating roofline), BASELINE configs[0] / [1] (DCN "
                         "README, DeepFM Criteo-Kaggle, D = 16) through the hugectr surface, the "
                         "embedding_collection legs (one-hot Criteo-1TB, multi-hot MLPerf DCNv2, "
                         "dynamic tables = configs[4]'s embedding half) and the tiered table "
                         "(configs[3]); auto = all of them when --config c3 runs on one GPU")
    ap.add_argument("--config", default="c3", choices=["c1", "c2", "c3"],
                    help="c3 = BASELINE configs[2], DLRM Criteo-1TB (the metric's configuration); "
                         "c1 / c2 = configs[0] / [1] as the main line")
    ap.add_argument("--extra-steps", type=int, default=10)
    ap.add_argument("--extra-file", default=os.path.join(ROOT, "bench_extra.json"),
                    help="the full record (every leg, stage tables, prose); the stdout line is its "
                         "<= 4 KB summary and names this file under `extra_file`")
    ap.add_argument("--tunable", default="auto", choices=["auto", "off"],
                    help="dense-tower GEMM solution selection: auto = hugectr.Model reads the "
                         "committed hugectr_amd/tuning/tunableop_gfx950.csv (solver."
                         "use_algorithm_search); off = library heuristics")
    a = ap.parse_args()
    if a.exchange:
        os.environ["HCTR_EXCHANGE"] = a.exchange
    if a.tunable == "off":
        os.environ["HCTR_TUNABLEOP"] = "off"

    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if world != a.gpus:
        if world == 1 and a.gpus > 1:
            sys.exit("--gpus N > 1 must be launched with torch.distributed.run (one rank per GPU)")
    if os.environ.get("HCTR_BENCH_BACKEND") == "gloo":
        # functional check of the N > 1 orchestration with all ranks on ONE GPU (collectives
        # staged through the host); never used for measurements
        local_rank = 0
        os.environ["HCTR_DIST_BACKEND"] = "gloo"
        os.environ["HCTR_RANKS_ON_ONE_GPU"] = "1"
    torch.cuda.set_device(local_rank)
    dev = torch.device("cuda", local_rank)
    if world > 1:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        if os.environ.get("HCTR_BENCH_BACKEND") == "gloo":
            dist.init_process_group("gloo")
        else:
            dist.init_process_group("nccl", device_id=dev)
            # a multi-GPU line is an RCCL line or no line at all: no silent fall to a host backend
            if dist.get_backend() != "nccl":
                sys.exit(f"bench.py --gpus {a.gpus}: the process group is on '{dist.get_backend()}', "
                         "not on nccl (= RCCL); refusing to time it")
            if torch.cuda.device_count() < world and os.environ.get("HCTR_RANKS_ON_ONE_GPU") != "1":
                sys.exit(f"bench.py --gpus {a.gpus}: {torch.cuda.device_count()} device(s) visible for "
                         f"{world} ranks")

    if a.config != "c3":
        out = small_config_leg(a.config, a.steps, a.warmup, dev)
        if rank == 0:
            emit(out, a.extra_file)
        return

    import gc
    first = "strong" if a.scaling == "strong" else "weak"
    out = dlrm_leg(a, a.precision, a.steps, a.warmup, world, rank, dev, scaling=first)
    if world > 1 and a.scaling == "both":
        # BASELINE configs[2] read literally: global batch 65536 over N GPUs
        gc.collect()
        torch.cuda.empty_cache()
        try:
            out["strong"] = dlrm_leg(a, a.precision, a.steps, a.warmup, world, rank, dev,
                                     scaling="strong")
        except Exception as e:  # (raised on every rank alike)
            out["strong"] = {"error": repr(e)}
    elif world == 1:
        out["strong"] = "at N = 1 the strong-scaling line (global batch 65536) IS this line"
    if rank == 0 and world == 1 and a.extra != "none":
        extra = {}
        sel = a.extra

        def run(name, kinds, fn):
            if sel not in kinds:
                return
            gc.collect()
            torch.cuda.empty_cache()  # (100 GB tables: the previous leg's must be gone first)
            try:
                extra[name] = fn()
            except Exception as e:  # an extra leg never takes the main line down
                extra[name] = {"error": repr(e)}

        def other_precision():
            prec = "fp32" if a.precision == "fp16" else "fp16"
            leg = dlrm_leg(a, prec, a.extra_steps, 4, world, rank, dev)
            r = {k: leg[k] for k in ("value", "unit", "ms_per_step", "steps", "warmup", "dtype",
                                     "roofline", "roofline_update", "stage_us_per_step")}
            r["final_loss"] = leg["config"]["final_loss"]
            return r

        def uniform_big():
            # no cache-resident table, no repeated row: the six tables of >= 2.9 M rows in all
            # 26 slots, uniform keys -- every row read is a
            # compulsory HBM read, so this leg's roofline fraction says what the KERNEL does
            # (each capped at 16 M rows: 26 x 8 GiB of rows fit one GPU beside the hash index)
            big = sorted([v for v in CRITEO_1TB if v >= 2900000], reverse=True)
            sizes = [min(big[i % len(big)], 16000000) for i in range(26)]
            leg = dlrm_leg(a, a.precision, a.extra_steps, 4, world, rank, dev, alpha=0.0,
                           sizes=sizes, label="uniform_big_tables")
            r = {k: leg[k] for k in ("value", "unit", "ms_per_step", "steps", "warmup",
                                     "roofline", "roofline_update", "roofline_index",
                                     "stage_us_per_step", "data")}
            r["workload"] = ("the main line's model with 26 slots over the six Criteo-1TB tables "
                             f"of >= 2.9 M rows, 16 M rows at most each ({sum(sizes)} rows, "
                             f"{sum(sizes) * 512 / 2**30:.0f} GiB), uniform keys: nothing is "
                             "cache-resident, no row repeats")
            r["distinct_rows_per_batch"] = leg["config"]["distinct_rows_per_batch"]
            r["new_keys_per_step"] = leg["config"]["new_keys_per_step"]
            return r

        run(("fp32" if a.precision == "fp16" else "fp16"), ("auto", "all"), other_precision)
        run("uniform_big_tables", ("auto", "all", "uniform"), uniform_big)
        run("c1", ("auto", "all"), lambda: small_config_leg("c1", 50, 20, dev))
        run("c2", ("auto", "all"), lambda: small_config_leg("c2", 50, 20, dev))
        run("dlrm_ebc_model", ("auto", "all", "model"),
            lambda: dlrm_ebc_model_leg(a.extra_steps, 4, dev, alpha=a.alpha))
        run("interaction", ("auto", "all", "dense"), lambda: interaction_leg(dev))
        run("cross", ("auto", "all", "dense"), lambda: cross_leg(dev))
        run("dcnv2_model", ("auto", "all", "model", "dcnv2"),
            lambda: dcnv2_model_leg(a.extra_steps, 3, dev, alpha=a.alpha))
        run("ebc_one_hot", ("auto", "all", "ebc"),
            lambda: ebc_leg("one_hot", a.extra_steps, 3, dev, a.alpha))
        run("ebc_multi_hot", ("auto", "all", "ebc"),
            lambda: ebc_leg("multi_hot", a.extra_steps, 3, dev, a.alpha))
        run("ebc_dynamic_multi_hot", ("auto", "all", "ebc", "next"),
            lambda: ebc_leg("multi_hot", a.extra_steps, 3, dev, a.alpha, dynamic=True))
        run("c5_wdl_mmoe_dynamic", ("auto", "all", "next", "model", "c5"),
            lambda: c5_model_leg(a.extra_steps * 3, 8, dev))
        run("tiered", ("auto", "all", "next", "tiered"), lambda: tiered_leg(a.extra_steps, 3, dev, a.alpha))
        run("ebc_dynamic_optimizers", ("all", "dynopt"),
            lambda: dynamic_optimizer_leg(a.extra_steps, 3, dev, a.alpha))
        out["extra"] = extra
        # the discriminating figures as top-level keys (a driver that keeps the head of the line
        # sees them): the gather on uniform keys over big tables -- nothing cache-resident, no
        # repeated row -- and the fp32 leg
        if isinstance(extra.get("uniform_big_tables"), dict) and "roofline" in extra["uniform_big_tables"]:
            out["roofline_uniform"] = extra["uniform_big_tables"]["roofline"]
        if isinstance(extra.get("fp32"), dict) and "roofline" in extra["fp32"]:
            out["roofline_fp32"] = extra["fp32"]["roofline"]
    if rank == 0:
        if not a.no_cpu_baseline and world == 1:
            try:
                out["cpu_baseline"] = cpu_baseline(CRITEO_1TB, a.alpha, a.dim, 4321)
            except Exception as e:  # the oracle is a reported baseline, never the product path
                out["cpu_baseline"] = {"error": repr(e)}
        emit(out, a.extra_file)
    if world > 1:
        dist.barrier()
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
