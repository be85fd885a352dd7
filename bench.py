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
sys.path.insert(0, ROOT)

# R/test/embedding_collection_test/dgx_a100_one_hot.py:24-51 -- Criteo-1TB slot_size_array
CRITEO_1TB = [39884406, 39043, 17289, 7420, 20263, 3, 7120, 1543, 63, 38532951, 2953546, 403346,
              10, 2208, 11938, 155, 4, 976, 14, 39979771, 25641295, 39664984, 585935, 12972, 108,
              36]
DENSE_DIM = 13
BOTTOM = [512, 256, 128]          # R/samples/dlrm/train.py:415-458 bottom MLP
TOP = [1024, 1024, 512, 256, 1]   # top MLP
HBM_PEAK_GBPS = 8000.0            # MI355X_MICROARCH.md: HBM3E 8.0 TB/s spec (6.29 TB/s copy)


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


def mlp(dims, last_relu):
    layers = []
    for i in range(len(dims) - 1):
        layers.append(torch.nn.Linear(dims[i], dims[i + 1]))
        if i < len(dims) - 2 or last_relu:
            layers.append(torch.nn.ReLU())
    return torch.nn.Sequential(*layers)


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


def dlrm_leg(a, precision, steps, warmup, world, rank, dev, shared):
    """one measurement of the DLRM Criteo-1TB step at `precision` (see --precision); returns the
    JSON object of that leg.  Builds (and releases) its own embedding: the 89.5 GiB table exists
    once at a time."""
    import hugectr_amd as ha
    from hugectr_amd import _lib
    from hugectr_amd.parallel import LocalizedExchange
    from hugectr_amd.parallel import all_reduce as par_all_reduce

    # precision -> (pooled-vector type, dense-tower type, loss scaler)
    edt, ddt, scaler = {"fp16": (torch.float16, torch.float16, 1024.0),
                        "bf16": (torch.bfloat16, torch.bfloat16, 1.0),
                        "fp32": (torch.float32, torch.float32, 1.0)}[precision]
    esz = 2 if edt != torch.float32 else 4
    amp = ddt != torch.float32
    sizes = [max(1, int(v * a.table_scale)) for v in CRITEO_1TB]
    S, D = len(sizes), a.dim
    Bl = a.batch                     # samples per GPU per step (fixed: weak scaling)
    B = Bl * world                   # global batch every rank resolves its slots for
    C = a.chunks if a.chunks > 0 else 1
    assert Bl % C == 0
    Bc = Bl // C                     # samples per GPU per sub-batch
    Bsub = Bc * world                # one sub-batch = a "global batch" of the reference layout:
    #   global sample g belongs to sub-batch g // Bsub and to rank (g % Bsub) // Bc, so the pooled
    #   vectors [B, S_r, D] in natural order are already [sub-batch][peer][Bc][S_r][D].
    spr = S // world + (1 if rank < S % world else 0)
    my_rows = sum(v for i, v in enumerate(sizes) if i % world == rank)
    max_rows = max(sum(v for i, v in enumerate(sizes) if i % world == r) for r in range(world))

    # ---- the embedding (this rank's slots), SGD as in the reference DLRM samples -----------------
    opt = ha.OptParams(optimizer=_lib.OPT_SGD, lr=0.01, atomic_update=a.sgd_atomic, scaler=scaler)
    emb = ha.SparseEmbeddingHash(_lib.EMB_LOCALIZED, B, 0, max_rows, D, S, S, 0, opt,
                                 slot_size_array=sizes, out_dtype=edt, rank=rank, world=world,
                                 seed=1234)
    emb.init_params()
    exch = LocalizedExchange(Bsub, S, D)
    ux = None
    if world > 1 and C == 1 and a.exchange != "rows":
        from hugectr_amd.unique_exchange import UniqueExchange
        ux = UniqueExchange(emb, Bl, S, D)
    mode = {"name": a.exchange if (ux is not None and a.exchange != "auto") else "rows"}

    def set_mode(name):
        mode["name"] = name
        if ux is not None and name != "rows":
            ux.set_sum_dtype(edt if name == "unique16" else torch.float32)

    set_mode(mode["name"])

    # ---- synthetic data, resident in HBM before the timed region ---------------------------------
    ro = torch.arange(0, B * S + 1, dtype=torch.int64, device=dev)
    key_batches = shared["keys"]
    g = torch.Generator(device=dev)
    g.manual_seed(99 + rank)
    dense_batches = [torch.rand((Bl, DENSE_DIM), device=dev, generator=g) for _ in range(a.nbatches)]
    label_batches = [(torch.rand((Bl, 1), device=dev, generator=g) < 0.5).float()
                     for _ in range(a.nbatches)]

    # ---- dense tower (PyTorch-ROCm / hipBLASLt GEMMs; interaction is our HIP kernel) ---------------
    torch.manual_seed(7)
    n_ins = S + 1
    if amp:
        from hugectr_amd.dense import FusedMLP, bce_with_logits
        bottom = FusedMLP([DENSE_DIM] + BOTTOM, last_relu=True, dtype=ddt).to(dev)
        top = FusedMLP([D + n_ins * (n_ins - 1) // 2 + 1] + TOP, last_relu=False, dtype=ddt).to(dev)
    else:
        bottom = mlp([DENSE_DIM] + BOTTOM, last_relu=True).to(dev)
        top = mlp([D + n_ins * (n_ins - 1) // 2 + 1] + TOP, last_relu=False).to(dev)
    dense_params = list(bottom.parameters()) + list(top.parameters())
    dense_opt = torch.optim.SGD(dense_params, lr=0.01)
    # flat master / gradient / 16-bit buffers: backward writes gradients in place, the SGD step and
    # the shadow refresh are one kernel per MLP, the data-parallel all-reduce needs no packing
    flat_mode = amp and C == 1 and a.graph != "on"
    # HCTR_BENCH_HEAD=0 keeps the logit layer + loss as separate library / HIP calls (A/B runs)
    head_fused = amp and os.environ.get("HCTR_BENCH_HEAD", "1") != "0" and top.can_fuse_bce_head()
    if flat_mode:
        bottom.flatten()
        top.flatten()
    loss_fn = torch.nn.BCEWithLogitsLoss()
    pooled = torch.empty((B, spr, D), dtype=edt, device=dev)
    top_grad = torch.empty((B, spr, D), dtype=edt, device=dev) if world > 1 or C > 1 else None

    tuned = "off"
    if a.tunable == "tune" or (a.tunable == "auto" and os.path.exists(a.tunable_file)):
        import torch.cuda.tunable as tunable
        tunable.enable(True)
        tunable.set_filename(a.tunable_file)
        if a.tunable == "tune":
            tunable.tuning_enable(True)
            tunable.set_max_tuning_duration(30)
            tunable.set_max_tuning_iterations(20)
            tuned = "tuned-now"
        else:
            tunable.tuning_enable(False)
            tunable.read_file(a.tunable_file)
            tuned = "file"
    def dense_chunk(dense_k, label_k, E, get_E=None, on_E_grad=None):
        """bottom MLP -> interaction -> top MLP -> BCE (scaled 1/C) -> backward.  E is a leaf; when
        get_E is given the bottom MLP is launched first (it does not need the embeddings, so it runs
        under the all-to-all) and get_E() waits for / reorders the received vectors.  on_E_grad(g)
        fires as soon as dL/dE exists -- before the bottom MLP's backward -- so the gradient
        all-to-all starts under the rest of the backward pass."""
        xb = bottom(dense_k)
        if get_E is not None:
            E = get_E()
        if on_E_grad is not None:
            E.register_hook(on_E_grad)
        z = ha.interaction(xb.to(edt), E)
        if amp and head_fused:
            # last layer + BCE + their backward in one pass over the last hidden activations
            loss = top.forward_bce(z, label_k, scaler / (Bc * C * world))
            loss.backward()
            return loss.detach() / C
        logit = top(z)
        if amp:  # fused BCE forward + logit gradient (HIP), mean over the step's Bl samples
            loss, dlogit = bce_with_logits(logit, label_k, scaler / (Bc * C * world))
            logit.backward(dlogit)
            return loss / C
        loss = loss_fn(logit.float(), label_k) / C
        (loss / world).backward()
        return loss.detach()

    use_graph = a.graph == "on" or (a.graph == "auto" and C > 1)
    graph = None
    if use_graph:
        # static buffers + whole fwd/bwd capture of one sub-batch; parameter gradients exist before
        # the capture so that backward ACCUMULATES into them across the C replays of a step
        st_dense = torch.zeros((Bc, DENSE_DIM), device=dev)
        st_label = torch.zeros((Bc, 1), device=dev)
        st_E = torch.zeros((Bc, S, D), dtype=edt, device=dev).requires_grad_(True)
        for q in dense_params:
            q.grad = torch.zeros_like(q)
        try:
            side = torch.cuda.Stream()
            side.wait_stream(torch.cuda.current_stream())
            with torch.cuda.stream(side):
                for _ in range(3):
                    st_E.grad = None
                    dense_chunk(st_dense, st_label, st_E)
            torch.cuda.current_stream().wait_stream(side)
            st_E.grad = None
            graph = torch.cuda.CUDAGraph()
            with torch.cuda.graph(graph):
                st_loss = dense_chunk(st_dense, st_label, st_E)
            for q in dense_params:
                q.grad.zero_()
        except Exception as e:  # capture is an optimisation: fall back to eager launches
            if rank == 0:
                print(f"[bench] HIP-graph capture failed ({e!r}); running the dense tower eagerly",
                      file=sys.stderr)
            graph = None
            dense_opt.zero_grad(set_to_none=True)

    def dense_update():
        if flat_mode:  # w -= lr * g / scaler (the loss scaler leaves the gradients here)
            bottom.sgd_step(0.01, 1.0 / scaler)
            top.sgd_step(0.01, 1.0 / scaler)
            return
        if scaler != 1.0:
            for q in dense_params:
                if q.grad is not None:
                    q.grad /= scaler
        dense_opt.step()
        dense_opt.zero_grad(set_to_none=(graph is None))
        if amp:
            bottom.refresh_shadow()
            top.refresh_shadow()

    def finish_step():
        if world > 1 and flat_mode:
            for m in (bottom, top):  # gradients are shares of the global-batch mean: plain sum
                par_all_reduce(m.flat_g)
            return
        if world > 1:
            grads = [p.grad for p in dense_params]
            flat = torch.cat([x.reshape(-1) for x in grads])
            par_all_reduce(flat)
            off = 0
            for x in grads:
                x.copy_(flat[off:off + x.numel()].view_as(x))
                off += x.numel()

    def step_unique(i):
        """same step with the unique-row exchange.  With 16-bit vectors the received distinct rows
        are never expanded: the interaction kernels read them through the (sample, slot) -> row
        table; the gradient sums leave from inside backward."""
        keys = key_batches[i % a.nbatches]
        ux.forward_begin(ro, keys)
        nxt = key_batches[(i + 1) % a.nbatches]
        dense_k, label_k = dense_batches[i % a.nbatches], label_batches[i % a.nbatches]
        if amp and edt != torch.float32:
            xb = bottom(dense_k)                       # runs under the row all-to-all
            rows, row_of = ux.forward_finish(indexed=True)
            # next batch's index stage, plan, counts and (index, bucket) exchange: side stream,
            # under this step's dense tower; issued after this step's row all-to-all so that the
            # communicator serves the critical-path transfer first
            ux.prefetch(ro, nxt)
            z = ha.interaction_indexed(xb.to(edt), rows, row_of, on_emb_grad=ux.backward_begin)
            logit = top(z)
            loss, dlogit = bce_with_logits(logit, label_k, scaler / (Bl * world))
            logit.backward(dlogit)
        else:
            sent = {}

            def get_E():
                sent["E"] = ux.forward_finish().detach().requires_grad_(True)
                ux.prefetch(ro, nxt)
                return sent["E"]

            loss = dense_chunk(dense_k, label_k, None, get_E=get_E, on_E_grad=ux.backward_begin)
        finish_step()
        ux.backward_finish()
        dense_update()
        return loss

    def step(i):
        if mode["name"] != "rows":
            return step_unique(i)
        keys = key_batches[i % a.nbatches]
        dense = dense_batches[i % a.nbatches]
        label = label_batches[i % a.nbatches]
        emb.forward(True, ro, keys, out=pooled)
        recvs, works = [None] * C, [None] * C
        recvs[0], works[0] = exch.forward_async(pooled[0:Bsub])
        back = []
        total = None
        tg = top_grad
        for k in range(C):
            if k + 1 < C:  # next sub-batch's vectors travel while this one's dense tower runs
                recvs[k + 1], works[k + 1] = exch.forward_async(pooled[(k + 1) * Bsub:(k + 2) * Bsub])
            if graph is not None:
                if works[k] is not None:
                    works[k].wait()
                if world > 1:
                    ha.forward_reorder(recvs[k], Bc, S, D, world, out=st_E.detach())
                else:
                    st_E.detach().copy_(recvs[k].view(Bc, S, D))
                st_dense.copy_(dense[k * Bc:(k + 1) * Bc])
                st_label.copy_(label[k * Bc:(k + 1) * Bc])
                graph.replay()
                Eg, loss = st_E.grad, st_loss
            else:
                sent = {}

                def get_E(k=k):
                    if works[k] is not None:
                        works[k].wait()
                    E = (ha.forward_reorder(recvs[k], Bc, S, D, world) if world > 1
                         else recvs[k].view(Bc, S, D))
                    sent["E"] = E.detach().requires_grad_(True)
                    return sent["E"]

                def on_E_grad(g, k=k):
                    # runs inside backward, right after the interaction's backward kernel.  The
                    # hook must not keep `g` alive: autograd then steals it for E.grad instead of
                    # cloning 436 MB
                    gsend = (ha.backward_reorder(g.contiguous(), Bc, S, D, world) if world > 1
                             else g.reshape(-1))
                    sent["w"] = exch.backward_async(
                        gsend, top_grad[k * Bsub:(k + 1) * Bsub].view(-1))
                    sent["buf"] = gsend if world > 1 else None

                loss = dense_chunk(dense[k * Bc:(k + 1) * Bc], label[k * Bc:(k + 1) * Bc], None,
                                   get_E=get_E,
                                   on_E_grad=on_E_grad if top_grad is not None else None)
                if top_grad is None:
                    tg = sent["E"].grad
                else:
                    back.append((sent["w"], sent["buf"]))
                total = loss.clone() if total is None else total + loss
                continue
            total = loss.clone() if total is None else total + loss
            if top_grad is None:
                tg = Eg
            else:
                gsend = ha.backward_reorder(Eg, Bc, S, D, world) if world > 1 else Eg.reshape(-1)
                w = exch.backward_async(gsend, top_grad[k * Bsub:(k + 1) * Bsub].view(-1))
                back.append((w, gsend))
        for w, _ in back:
            if w is not None:
                w.wait()
        finish_step()
        emb.backward(tg)
        emb.update_params()
        dense_update()  # (static graph gradient buffers are zeroed in place, not released)
        return total

    emb.profiling(True)
    if ux is not None and a.exchange == "auto":
        # measure, don't guess: a few steps of each payload during warm-up, keep the faster one
        # (the decision is taken on the max over ranks, so every rank takes the same one)
        timing = {}
        # every cycled batch once first: the tables then hold all keys, so neither candidate is
        # timed on steps that insert (and RCCL's first-call setup is out of the way)
        set_mode("rows")
        for i in range(a.nbatches):
            step(i)
        for name in ("rows", "unique"):  # unique16 changes the wire precision: opt-in only
            set_mode(name)
            try:
                for i in range(2):
                    step(i)
            except Exception as e:  # e.g. keys x peers beyond the 32-bit sort key: same on every rank
                if rank == 0:
                    print(f"[bench] exchange '{name}' unavailable: {e!r}", file=sys.stderr)
                timing[name] = float("inf")
                continue
            torch.cuda.synchronize()
            dist.barrier()
            t0 = time.perf_counter()
            for i in range(2, 5):
                step(i)
            torch.cuda.synchronize()  # (also drains the unique exchange's prefetch stream)
            t = torch.tensor([time.perf_counter() - t0], dtype=torch.float64)
            if dist.get_backend() != "gloo":
                t = t.to(dev)
            dist.all_reduce(t, op=dist.ReduceOp.MAX)
            timing[name] = float(t.item()) / 3
        set_mode(min(timing, key=timing.get))
        mode["timing_ms"] = {k: v * 1e3 for k, v in timing.items()}
    pool_prof = emb.profile().get("gather_pool", (0.0, 0))
    for i in range(warmup):
        step(i)
    torch.cuda.synchronize()
    if a.tunable == "tune":
        tunable.tuning_enable(False)  # keep the chosen solutions, stop searching
        if rank == 0:
            # TunableOp writes its CSV at process exit; also write it now in the same format so
            # the file exists even if the interpreter is torn down abnormally.
            try:
                os.makedirs(os.path.dirname(a.tunable_file), exist_ok=True)
                with open(a.tunable_file + ".now", "w") as f:
                    for k, v in tunable.get_validators():
                        f.write(f"Validator,{k},{v}\n")
                    for r in tunable.get_results():
                        f.write(",".join(str(x) for x in r) + "\n")
            except Exception as e:  # diagnostics only
                print("tunable dump failed:", e, file=sys.stderr)
    emb.profiling(True)
    if world > 1:
        dist.barrier()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    loss = None
    for i in range(warmup, warmup + steps):
        loss = step(i)
    torch.cuda.synchronize()
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
    prof = emb.profile()
    emb.profiling(False)
    emb.check_overflow()

    # ---- roofline of the gather+pool kernel (algorithmic bytes, DESIGN.md / SURVEY 8d) ------------
    nnz_g = B * spr  # one-hot: one key per (sample, slot on this rank)
    alg_bytes = nnz_g * 8 + nnz_g * 8 + nnz_g * D * 4 + B * spr * D * esz
    pool_ms, pool_n = prof["gather_pool"]
    if pool_n == 0:  # unique-row exchange: the pool kernel only ran in the warm-up comparison
        pool_ms, pool_n = pool_prof
    achieved = alg_bytes / (pool_ms / max(pool_n, 1) * 1e-3) / 1e9 if pool_ms > 0 else 0.0
    # HBM bytes per launch from the PMC counters: collected with rocprofv3 in separate passes on
    # this workload (profiles/r2_pmc_hbm_traffic_<precision>.json; a bench process cannot read the
    # counters of its own kernels), attached only to the leg whose output width they were taken on
    pmc = None
    pmc_path = os.path.join(ROOT, "profiles",
                            f"r2_pmc_hbm_traffic_{'fp32' if esz == 4 else 'fp16'}.json")
    if world == 1 and D == 128 and a.batch == 65536 and a.alpha == 1.1 and os.path.exists(pmc_path):
        try:
            pmc = json.load(open(pmc_path))["kernels"]["pool_vec4_kernel"]["hbm_bytes_per_launch"]
        except Exception:
            pmc = None
    stage_us = {k: (v[0] / max(v[1], 1)) * 1e3 for k, v in prof.items()}
    # ---- per-rank diagnosis of a multi-GPU run: what each rank resolved, what it shipped --------
    per_rank = None
    if world > 1:
        if mode["name"] == "rows":
            sent = sum(exch.send) - exch.send[rank]      # elements to the other ranks, one way
            xb = {"payload": "rows", "bytes_out_forward": sent * esz, "bytes_out_backward":
                  (sum(exch.recv) - exch.recv[rank]) * esz}
        else:
            us, ur = ux.u_send or [0] * world, ux.u_recv or [0] * world
            gsz = 2 if mode["name"] == "unique16" else 4
            xb = {"payload": mode["name"], "distinct_rows_out": sum(us) - us[rank],
                  "positions": ux.P,
                  "bytes_out_forward": (sum(us) - us[rank]) * D * esz + (ux.P - ux.P // world) * 8,
                  "bytes_out_backward": (sum(ur) - ur[rank]) * D * gsz}
        mine = {"rank": rank, "slots": spr, "table_rows": my_rows, "stage_us": stage_us,
                "exchange": xb}
        per_rank = [None] * world
        dist.all_gather_object(per_rank, mine)

    out = {
        "metric": "samples/sec (whole node) + embedding-gather HBM GB/s, DLRM Criteo-1TB",
        "value": B * steps / elapsed, "unit": "samples/s", "n_gpus": world, "steps": steps,
        "warmup": warmup, "ms_per_step": elapsed / steps * 1e3, "higher_is_better": True,
        "scaling": "weak", "vs_baseline": None,
        "dtype": {"fp16": "fp16 = the reference's mixed precision (use_mixed_precision, scaler 1024): "
                          "fp16 pooled vectors + top gradients + dense GEMMs (fp32 accumulate), "
                          "fp32 tables / pooling accumulation / sparse SGD, loss scaling",
                  "fp32": "fp32 = the reference's default: fp32 tables, pooled vectors, gradients, "
                          "sparse SGD and dense GEMMs (interaction: fp32 I/O, 3x bf16-split MFMA)",
                  "bf16": "bf16 (not a reference mode): bf16 pooled vectors + top gradients + dense "
                          "GEMMs, fp32 tables / pooling accumulation / sparse SGD"}[precision],
        "precision": precision,
        "data": f"synthetic power-law alpha={a.alpha} (uniform if 0), one-hot, resident in HBM",
        "config": {"workload": "BASELINE configs[2]: DLRM Criteo-1TB slot_size_array, "
                               "LocalizedSlotSparseEmbeddingHash, emb_dim=128, bs=65536 per GPU, SGD",
                   "batch_per_gpu": Bl, "global_batch": B, "sub_batches_per_step": C, "dense_tower_hip_graph": graph is not None,
                   "exchange": mode["name"] if world > 1 else "none (1 GPU)",
                   "exchange_warmup_ms_per_step": mode.get("timing_ms"), "slots": S, "emb_dim": D, "table_rows_total": sum(sizes),
                   "table_rows_this_rank": my_rows, "parallelism": f"slot-sharded x{world} + dp{world}",
                   "final_loss": float(loss.detach()), "dense_gemm_selection": tuned},
        "roofline": {"bound": "hbm", "kernel": "pool_vec4_kernel (gather + intra-slot pooling)",
                     "achieved": achieved, "peak": HBM_PEAK_GBPS, "unit": "GB/s",
                     "frac": achieved / HBM_PEAK_GBPS, "traffic": pmc,
                     "traffic_source": os.path.relpath(pmc_path, ROOT) if pmc else None,
                     "algorithmic_bytes_per_launch": alg_bytes, "launches": pool_n,
                     "avg_launch_us": pool_ms / max(pool_n, 1) * 1e3,
                     # duplicates counted.  Power-law keys repeat hot rows, which L2 /
                     # Infinity Cache serve -- `traffic` is what actually crossed the HBM interface,
                     # so `achieved` can pass the HBM peak while traffic / time stays below it
                     "hbm_traffic_gbps": (pmc / (pool_ms / max(pool_n, 1) * 1e-3) / 1e9)
                     if pmc and pool_ms > 0 else None},
        "stage_us_per_step": stage_us,
    }
    if per_rank is not None:
        out["per_rank"] = per_rank
    del emb, ux, exch, pooled, top_grad, bottom, top, dense_params, dense_opt
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


# R/samples/dlrm/train.py:29-83 (MLPerf DLRM-DCNv2): tables capped at 40 M rows, keys per sample
MLPERF_TABLES = [40000000, 39060, 17295, 7424, 20265, 3, 7122, 1543, 63, 40000000, 3067956, 405282,
                 10, 2209, 11938, 155, 4, 976, 14, 40000000, 40000000, 40000000, 590152, 12973,
                 108, 36]
MLPERF_HOTNESS = [3, 2, 1, 2, 6, 1, 1, 1, 1, 7, 3, 8, 1, 6, 9, 5, 1, 1, 1, 12, 100, 27, 10, 3, 1, 1]


def ebc_leg(kind, steps, warmup, dev, alpha=1.1, B=65536, D=128):
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
    ebc = EmbeddingCollection(cfg, B, lr=0.01, optimizer=_lib.OPT_SGD, scaler=1024.0,
                              out_dtype=torch.float16, batch_major=True, max_hotness=max(hot),
                              hotness=hot)
    g = torch.Generator(device=dev)
    g.manual_seed(99)
    batches = []
    for _ in range(2):  # feature-major CSR: bucket = table * B + sample, hot[t] keys each
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
        for i in range(warmup):
            fn(i)
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for i in range(steps):
            fn(i)
        e1.record()
        torch.cuda.synchronize()
        return e0.elapsed_time(e1) / steps * 1e3

    fwd_us = timed(lambda i: ebc.forward(*batches[i % 2]))

    def train(i):
        ebc.forward(*batches[i % 2])
        ebc.backward_and_update(grad)
    both_us = timed(train)
    nnz = B * sum(hot)
    alg = nnz * (8 + 8 + D * 4) + B * 26 * D * 2
    ach = alg / (fwd_us * 1e-6) / 1e9
    return {
        "workload": ("embedding_collection, Criteo-1TB tables, one-hot" if kind == "one_hot" else
                     "embedding_collection, MLPerf DLRM-DCNv2 tables and hotness (214 keys / "
                     "sample), R/samples/dlrm/train.py:29-83") +
                    f", B={B}, D={D}, fp16 output [B][26][D], SGD, power-law alpha={alpha}; the "
                    "collection alone (no dense tower)",
        "direct_one_gpu_path": bool(ebc._direct), "keys_per_batch": nnz,
        "table_rows_total": int(sum(sizes)),
        "forward_us": fwd_us, "forward_backward_update_us": both_us,
        "backward_update_us": both_us - fwd_us,
        "value": B / (both_us * 1e-6), "unit": "samples/s (embedding path only)",
        "roofline": {"bound": "hbm", "kernel": "whole forward (key -> row pass + gather/pool): a "
                                               "lower bound on the gather kernel's own rate",
                     "achieved": ach, "peak": HBM_PEAK_GBPS, "unit": "GB/s",
                     "frac": ach / HBM_PEAK_GBPS, "traffic": None,
                     "algorithmic_bytes_per_launch": alg},
    }


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=30)
    ap.add_argument("--warmup", type=int, default=8)
    ap.add_argument("--batch", type=int, default=65536,
                    help="batch PER GPU (BASELINE config 3: 65536); weak scaling: global = N * batch")
    ap.add_argument("--graph", default="auto", choices=["auto", "on", "off"],
                    help="capture the per-sub-batch dense tower (bottom MLP, interaction, top MLP, "
                         "loss, backward) in a HIP graph; auto = when chunks > 1")
    ap.add_argument("--chunks", type=int, default=0,
                    help="sub-batches per step whose all-to-all overlaps the dense tower of the "
                         "previous one.  Default 1: measured on MI355X, 4 sub-batches of 16384 cost "
                         "+2.1 ms of dense-tower time per step (smaller GEMMs / reductions), which "
                         "is what the overlap could save at N = 8, so no split is the default")
    ap.add_argument("--exchange", default="auto", choices=["auto", "rows", "unique", "unique16"],
                    help="multi-GPU payload of the embedding exchange: rows = one pooled vector / "
                         "gradient per (sample, slot) as the reference; unique = every distinct row "
                         "once per destination + per-row gradient sums "
                         "(hugectr_amd/unique_exchange.py), sums on the wire in fp32; unique16 = "
                         "the same with the sums in the pooled vectors' 16-bit type (the precision "
                         "class of the per-sample gradients the rows payload ships; opt-in); auto = "
                         "time rows and unique during warm-up and keep the faster")
    ap.add_argument("--alpha", type=float, default=1.1, help="power-law exponent; 0 = uniform")
    ap.add_argument("--dim", type=int, default=128)
    ap.add_argument("--table-scale", type=float, default=1.0)
    ap.add_argument("--nbatches", type=int, default=8)
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--sgd-atomic", action="store_true")
    ap.add_argument("--precision", default="fp16", choices=["fp16", "fp32", "bf16"],
                    help="precision of the MAIN line.  fp16 = the reference's mixed precision "
                         "(use_mixed_precision=True, scaler=1024: fp16 pooled vectors / top "
                         "gradients / dense GEMMs, loss scaling; the MLPerf DLRM configuration of "
                         "the reference), fp32 = the reference's default (everything fp32), bf16 = "
                         "not a reference mode (MI355X-native 16-bit type, no loss scaling).  "
                         "Tables, pooling accumulation and the sparse optimizer are fp32 in all "
                         "three.")
    ap.add_argument("--extra", default="auto", choices=["auto", "none", "all", "ebc", "model"],
                    help="extra legs appended to the JSON line under `extra` (1 GPU only): the "
                         "other two precisions on the same workload and BASELINE configs[0] / [1] "
                         "(DCN README, DeepFM Criteo-Kaggle, D = 16) through the hugectr surface; "
                         "auto = all of them when --config c3 runs on one GPU, plus the "
                         "embedding_collection legs (one-hot Criteo-1TB, multi-hot MLPerf DCNv2); "
                         "ebc = only those")
    ap.add_argument("--config", default="c3", choices=["c1", "c2", "c3"],
                    help="c3 = BASELINE configs[2], DLRM Criteo-1TB (the metric's configuration); "
                         "c1 / c2 = configs[0] / [1] as the main line")
    ap.add_argument("--extra-steps", type=int, default=10)
    ap.add_argument("--tunable", default="auto", choices=["auto", "tune", "off"],
                    help="dense-tower GEMM solution selection through PyTorch TunableOp: auto = use "
                         "the committed hugectr_amd/tuning/tunableop_gfx950.csv if present (no "
                         "tuning at run time); tune = search during warm-up (outside the timed "
                         "region) and write --tunable-file; off = library heuristics")
    ap.add_argument("--tunable-file", default=os.path.join(ROOT, "hugectr_amd", "tuning",
                                                           "tunableop_gfx950.csv"))
    a = ap.parse_args()

    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if world != a.gpus:
        if world == 1 and a.gpus > 1:
            sys.exit("--gpus N > 1 must be launched with torch.distributed.run (one rank per GPU)")
    if os.environ.get("HCTR_BENCH_BACKEND") == "gloo":
        local_rank = 0
    torch.cuda.set_device(local_rank)
    dev = torch.device("cuda", local_rank)
    if world > 1:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        if os.environ.get("HCTR_BENCH_BACKEND") == "gloo":
            # functional check of the N > 1 orchestration with all ranks on ONE GPU (collectives
            # staged through the host); never used for measurements
            dist.init_process_group("gloo")
        else:
            dist.init_process_group("nccl", device_id=dev)

    if a.config != "c3":
        out = small_config_leg(a.config, a.steps, a.warmup, dev)
        if rank == 0:
            print(json.dumps(out))
        return

    rng = np.random.default_rng(1234)  # every rank draws the same full-batch CSR (reader semantics)
    sizes = [max(1, int(v * a.table_scale)) for v in CRITEO_1TB]
    shared = {"keys": [torch.from_numpy(make_keys(rng, a.batch * world, sizes, a.alpha)).to(dev)
                       for _ in range(a.nbatches)]}
    out = dlrm_leg(a, a.precision, a.steps, a.warmup, world, rank, dev, shared)
    if rank == 0 and world == 1 and a.extra != "none":
        extra = {}
        for prec in ("fp32", "fp16", "bf16"):
            if prec == a.precision or a.extra in ("ebc", "model"):
                continue
            try:
                leg = dlrm_leg(a, prec, a.extra_steps, max(a.nbatches, 4), world, rank, dev, shared)
                extra[prec] = {k: leg[k] for k in ("value", "unit", "ms_per_step", "steps", "warmup",
                                                   "dtype", "roofline", "stage_us_per_step")}
                extra[prec]["final_loss"] = leg["config"]["final_loss"]
            except Exception as e:  # an extra leg never takes the main line down
                extra[prec] = {"error": repr(e)}
        for cfg in ("c1", "c2"):
            if a.extra in ("ebc", "model"):
                continue
            try:
                extra[cfg] = small_config_leg(cfg, 50, 20, dev)
            except Exception as e:
                extra[cfg] = {"error": repr(e)}
        import gc
        if a.extra != "ebc":
            gc.collect()
            torch.cuda.empty_cache()
            try:
                extra["dlrm_ebc_model"] = dlrm_ebc_model_leg(a.extra_steps, 4, dev, alpha=a.alpha)
            except Exception as e:
                extra["dlrm_ebc_model"] = {"error": repr(e)}
        for kind in ("one_hot", "multi_hot"):
            if a.extra == "model":
                continue
            gc.collect()
            torch.cuda.empty_cache()  # (100 GB tables: the previous leg's must be gone first)
            try:
                extra["ebc_" + kind] = ebc_leg(kind, a.extra_steps, 3, dev, a.alpha)
            except Exception as e:
                extra["ebc_" + kind] = {"error": repr(e)}
        out["extra"] = extra
    if rank == 0:
        if not a.no_cpu_baseline and world == 1:
            try:
                out["cpu_baseline"] = cpu_baseline(CRITEO_1TB, a.alpha, a.dim, 4321)
            except Exception as e:  # the oracle is a reported baseline, never the product path
                out["cpu_baseline"] = {"error": repr(e)}
        print(json.dumps(out))
    if world > 1:
        dist.barrier()
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
