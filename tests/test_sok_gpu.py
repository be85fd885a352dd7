"""GPU parity of the SOK-shaped surface (hugectr_amd/sok.py): lookup_sparse forward / backward /
optimizer step against plain PyTorch (embedding_bag, the role tf.nn.embedding_lookup_sparse plays
in R/sparse_operation_kit/sparse_operation_kit/test/function_test), static and dynamic variables,
and a 2-process run (gloo, both ranks on this one GPU) of the sharded route."""
import os

import numpy as np
import pytest

from util import assert_close

pytestmark = pytest.mark.gpu


def _ragged(torch, rng, batch, max_hot, vocab, with_w=False):
    from hugectr_amd import sok
    lens = rng.integers(0, max_hot + 1, size=batch)
    lens[rng.random(batch) < 0.15] = 0
    vals = rng.integers(0, vocab, size=int(lens.sum()))
    ids = sok.Ragged(torch.from_numpy(vals).cuda(), torch.from_numpy(lens).cuda())
    w = None
    if with_w:
        w = sok.Ragged(torch.from_numpy(rng.random(vals.size).astype(np.float32) + 0.1).cuda(),
                       torch.from_numpy(lens).cuda())
    return ids, w


def _bag(torch, table, ids, w, mode):
    off = torch.cumsum(ids.row_lengths, 0) - ids.row_lengths
    if w is None:
        return torch.nn.functional.embedding_bag(ids.values, table, off, mode=mode)
    out = torch.nn.functional.embedding_bag(ids.values, table, off, mode="sum",
                                            per_sample_weights=w.values)
    if mode == "mean":
        seg = torch.repeat_interleave(torch.arange(ids.batch, device="cuda"), ids.row_lengths)
        den = torch.zeros(ids.batch, device="cuda").index_add_(0, seg, w.values)
        out = out / den.clamp_min(1e-30).unsqueeze(1) * (den > 0).unsqueeze(1)
    return out


@pytest.mark.parametrize("with_w", [False, True])
def test_sok_static_lookup_backward_and_sgd(with_w):
    import torch
    from hugectr_amd import sok
    sok.init()
    rng = np.random.default_rng(3 + with_w)
    tabs = [rng.standard_normal((50, 8)).astype(np.float32),
            rng.standard_normal((33, 20)).astype(np.float32)]
    vs = [sok.Variable(t) for t in tabs]
    refs = [torch.from_numpy(t).cuda().requires_grad_() for t in tabs]
    combs = ["sum", "mean"]
    opt = sok.OptimizerWrapper("sgd", lr=0.1)
    for step in range(3):
        ids_w = [_ragged(torch, rng, 40, 5, t.shape[0], with_w) for t in tabs]
        ids, ws = [a for a, _ in ids_w], [b for _, b in ids_w]
        outs = sok.lookup_sparse(vs, ids, ws if with_w else None, combs)
        want = [_bag(torch, r, i, w, c) for r, i, w, c in zip(refs, ids, ws, combs)]
        for o, x in zip(outs, want):
            assert_close(o.detach().cpu().numpy(), x.detach().cpu().numpy(), 1e-5, 1e-6, "fwd")
        gs = [torch.from_numpy(rng.standard_normal(tuple(o.shape)).astype(np.float32)).cuda()
              for o in outs]
        sum((o * g).sum() for o, g in zip(outs, gs)).backward()
        sum((x * g).sum() for x, g in zip(want, gs)).backward()
        opt.step(vs)
        with torch.no_grad():
            for r in refs:
                r -= 0.1 * r.grad
                r.grad = None
        for v, r in zip(vs, refs):
            assert_close(v.numpy(), r.detach().cpu().numpy(), 1e-5, 1e-6, f"table step {step}")


def test_sok_dynamic_variable_lookup_and_adagrad():
    import torch
    from hugectr_amd import sok
    sok.init()
    rng = np.random.default_rng(9)
    D, lr, eps = 12, 0.05, 1e-6
    v = sok.DynamicVariable(D, initializer="0.5", init_capacity=64)
    ref, acc = {}, {}
    opt = sok.OptimizerWrapper("adagrad", lr=lr, epsilon=eps)
    for step in range(4):
        ids, _ = _ragged(torch, rng, 64, 6, 300)   # many repeats across steps
        out = sok.lookup_sparse(v, ids, combiners="sum")
        keys, lens = ids.values.cpu().numpy(), ids.row_lengths.cpu().numpy()
        want = np.zeros((64, D), dtype=np.float32)
        pos = 0
        for b, n in enumerate(lens):
            for k in keys[pos:pos + n]:
                want[b] += ref.setdefault(int(k), np.full(D, 0.5, dtype=np.float32))
            pos += n
        assert_close(out.detach().cpu().numpy(), want, 1e-5, 1e-6, f"dyn fwd {step}")
        g = rng.standard_normal((64, D)).astype(np.float32)
        (out * torch.from_numpy(g).cuda()).sum().backward()
        opt.step([v])
        sums = {}
        pos = 0
        for b, n in enumerate(lens):
            for k in keys[pos:pos + n]:
                sums[int(k)] = sums.get(int(k), 0) + g[b].astype(np.float64)
            pos += n
        for k, s in sums.items():
            s = s.astype(np.float32)
            a = acc.get(k, np.zeros(D, dtype=np.float32)) + s * s
            acc[k] = a
            ref[k] = ref[k] - lr * s / (np.sqrt(a) + eps)
        assert v.size == len(ref)
    ks, vals = sok.export(v)
    for k, x in zip(ks.cpu().numpy().tolist(), vals.cpu().numpy()):
        assert_close(x, ref[k], 1e-4, 1e-5, "dynamic table after adagrad")
    # eval lookup of unknown keys: zeros, nothing inserted
    unk = sok.Ragged(torch.tensor([10**12, 10**12 + 1], device="cuda"),
                     torch.tensor([2], device="cuda"))
    z = sok.lookup_sparse(v, unk, combiners="sum", training=False)
    assert float(z.detach().abs().max()) == 0.0 and v.size == len(ref)
    sok.assign(v, torch.tensor([7, 10**12], device="cuda"), torch.ones((2, D), device="cuda"))
    assert float(v.sparse_read(torch.tensor([10**12], device="cuda")).min()) == 1.0


def _worker(rank, world, port, ret):
    import torch
    import torch.distributed as dist
    os.environ["MASTER_ADDR"], os.environ["MASTER_PORT"] = "127.0.0.1", str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        torch.cuda.set_device(0)
        from hugectr_amd import sok
        sok.init()
        rng = np.random.default_rng(21)              # same stream on both ranks
        tab = rng.standard_normal((37, 8)).astype(np.float32)
        var = sok.Variable(tab)                       # rows r % 2 == rank live here
        loc = sok.Variable(tab, mode="localized:1")
        dyn = sok.DynamicVariable(8, initializer="ones")
        full = torch.from_numpy(tab).cuda().requires_grad_()
        B = 12                                        # per rank
        seen = set()
        for step in range(2):
            lens = rng.integers(0, 4, size=B * world)
            vals = rng.integers(0, 37, size=int(lens.sum()))
            g_all = rng.standard_normal((B * world, 8)).astype(np.float32)
            off = np.concatenate([[0], np.cumsum(lens)])
            sl = slice(off[rank * B], off[(rank + 1) * B])
            ids = sok.Ragged(torch.from_numpy(vals[sl]).cuda(),
                             torch.from_numpy(lens[rank * B:(rank + 1) * B]).cuda())
            for v, comb in ((var, "mean"), (loc, "sum")):
                out = sok.lookup_sparse(v, ids, combiners=comb)
                gi = torch.from_numpy(vals).cuda()
                offs = torch.from_numpy(off[:-1]).cuda()
                want = torch.nn.functional.embedding_bag(gi, full, offs, mode=comb)
                mine = want[rank * B:(rank + 1) * B]
                assert torch.allclose(out, mine, rtol=1e-5, atol=1e-6), (comb, step)
                gl = torch.from_numpy(g_all[rank * B:(rank + 1) * B]).cuda()
                (out * gl).sum().backward()
                (want * torch.from_numpy(g_all).cuda()).sum().backward()
                sok.OptimizerWrapper("sgd", lr=0.1).step([v])
                new_full = (full - 0.1 * full.grad).detach()
                full.grad = None
                if v is var:
                    assert torch.allclose(v.weight, new_full[rank::world], rtol=1e-5, atol=1e-6)
                elif rank == 1:
                    assert torch.allclose(v.weight, new_full, rtol=1e-5, atol=1e-6)
                with torch.no_grad():  # keep the three copies of the table in step
                    full.copy_(new_full)
                    var.weight.copy_(new_full[rank::world])
                    if rank == 1:
                        loc.weight.copy_(new_full)
            out = sok.lookup_sparse(dyn, ids, combiners="sum")   # all-ones rows: sum = hotness
            assert torch.allclose(out, ids.row_lengths.float().unsqueeze(1).expand(-1, 8))
            seen |= set(vals[vals % world == rank].tolist())
            assert dyn.size == len(seen)              # only the keys this rank owns live here
        # dump from both ranks (GPU 0 writes), load into fresh variables on both ranks
        import tempfile
        from hugectr_amd import sok_format as fmt
        d = os.path.join(tempfile.gettempdir(), f"sok_dump_{port}")
        var.name, loc.name, dyn.name = "dist_var", "loc_var", "dyn_var"
        sok.dump(d, [var, loc, dyn])
        keys = fmt.read_array_file(os.path.join(d, "dist_var-key"))
        # one round: rank 0's rows (keys 0, 2, 4 ...) then rank 1's (1, 3, 5 ...)
        assert keys.tolist() == list(range(0, 37, 2)) + list(range(1, 37, 2))
        assert fmt.read_array_file(os.path.join(d, "loc_var-key")).dtype == np.uint64
        var2 = sok.Variable(np.zeros((37, 8), np.float32), name="dist_var")
        loc2 = sok.Variable(np.zeros((37, 8), np.float32), mode="localized:1", name="loc_var")
        dyn2 = sok.DynamicVariable(8, initializer="zeros", name="dyn_var")
        sok.load(d, [var2, loc2, dyn2])
        assert torch.equal(var2.weight, var.weight) and torch.equal(loc2.weight, loc.weight)
        assert dyn2.size == dyn.size
        ret[rank] = "ok"
    except Exception as e:  # surface the failure in the parent
        import traceback
        ret[rank] = "".join(traceback.format_exception(type(e), e, e.__traceback__))
    finally:
        dist.destroy_process_group()


def test_sok_two_ranks_on_one_gpu_gloo():
    import torch.multiprocessing as mp
    ctx = mp.get_context("spawn")
    ret = ctx.Manager().dict()
    port = 29500 + os.getpid() % 2000
    procs = [ctx.Process(target=_worker, args=(r, 2, port, ret)) for r in range(2)]
    for p in procs:
        p.start()
    for p in procs:
        p.join(300)
    for r in range(2):
        if ret.get(r) != "ok":
            print(f"--- rank {r} ---\n{ret.get(r)}")
    assert ret.get(0) == "ok" and ret.get(1) == "ok"


def test_dump_load_roundtrip_in_the_reference_layout(tmp_path):
    """sok.dump / sok.load: files in the reference's layout (byte format pinned on the CPU in
    tests/test_sok_format_cpu.py); weights, keys and optimizer state survive the round trip"""
    import torch
    from hugectr_amd import sok, sok_format as fmt
    sok.init()
    rng = np.random.default_rng(3)
    D = 8
    tab = rng.standard_normal((50, D)).astype(np.float32)
    var = sok.Variable(tab, name="emb/user:0")
    dyn = sok.DynamicVariable(D, initializer="0.5", name="dyn_table")
    opt = sok.OptimizerWrapper("adam", lr=0.05)
    for step in range(2):
        lens = rng.integers(1, 4, size=16)
        vals = rng.integers(0, 50, size=int(lens.sum()))
        ids = sok.Ragged(torch.from_numpy(vals).cuda(), torch.from_numpy(lens).cuda())
        # (static and dynamic variables go through separate lookups, as in the reference)
        outs = [sok.lookup_sparse(var, ids, combiners="sum"),
                sok.lookup_sparse(dyn, ids, combiners="mean")]
        sum((o * o).sum() for o in outs).backward()
        opt.step([var, dyn])
    sok.dump(str(tmp_path), [var, dyn], opt)
    names = sorted(os.listdir(tmp_path))
    assert names == sorted(["meta_info", "emb_user_0-key", "emb_user_0-weight", "emb_user_0-Adam-m",
                            "emb_user_0-Adam-v", "dyn_table-key", "dyn_table-weight",
                            "dyn_table-Adam-m", "dyn_table-Adam-v"])
    meta = fmt.load_meta_file(str(tmp_path))
    assert meta["emb/user:0"].emb_num == 50 and meta["emb/user:0"].opt_name == "Adam"
    assert meta["dyn_table"].emb_num == dyn.size and meta["dyn_table"].emb_length == D
    assert fmt.read_file_head(str(tmp_path / "dyn_table-Adam-v")) == ("dyn_table", 2, "v", 5)
    keys = fmt.read_array_file(str(tmp_path / "dyn_table-key"))
    assert (np.diff(keys) > 0).all()                       # sorted by key, as the reference writes
    var2 = sok.Variable(np.zeros_like(tab), name="emb/user:0")
    dyn2 = sok.DynamicVariable(D, initializer="zeros", name="dyn_table")
    opt2 = sok.OptimizerWrapper("adam", lr=0.05)
    sok.load(str(tmp_path), [var2, dyn2], opt2)
    assert torch.equal(var2.weight, var.weight)
    assert all(torch.equal(a, b) for a, b in zip(var2._states, var._states)) and len(var2._states) == 2
    k1, v1 = sok.export(dyn)
    k2, v2 = sok.export(dyn2)
    o1, o2 = torch.argsort(k1), torch.argsort(k2)
    assert torch.equal(k1[o1], k2[o2]) and torch.equal(v1[o1], v2[o2])
    s1k, s1v = dyn._opt.states.export(0)
    s2k, s2v = dyn2._opt.states.export(0)
    assert torch.equal(s1v[torch.argsort(s1k)], s2v[torch.argsort(s2k)])
    # training continues identically from the restored state
    lens = rng.integers(1, 4, size=16)
    vals = rng.integers(0, 50, size=int(lens.sum()))
    ids = sok.Ragged(torch.from_numpy(vals).cuda(), torch.from_numpy(lens).cuda())
    opt2.times = opt.times
    for vs, o in (((var, dyn), opt), ((var2, dyn2), opt2)):
        outs = [sok.lookup_sparse(vs[0], ids, combiners="sum"),
                sok.lookup_sparse(vs[1], ids, combiners="mean")]
        sum((x * x).sum() for x in outs).backward()
        o.step(list(vs))
    assert torch.allclose(var2.weight, var.weight, rtol=1e-6, atol=1e-7)
