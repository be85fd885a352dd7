"""CPU, world_size 2 over gloo: the multi-GPU exchange bookkeeping (split sizes, wire layout,
reorder) of hugectr_amd.parallel, with the per-rank pooled vectors produced by the CPU oracle.
This is the N > 1 path of bench.py minus the HIP kernels."""
import os
import sys

import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))


def _worker(rank, world, port, emb_type, ret):
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        from oracle import pyoracle as orc
        from hugectr_amd.parallel import DistributedExchange, LocalizedExchange, slots_on_rank
        from util import make_csr
        rng = np.random.default_rng(123)  # same data on every rank (full-batch CSR)
        B, S, D, hot, vps = 16, 7, 8, 3, 20
        V = S * vps
        ro, keys = make_csr(rng, B, S, hot, vps)
        dense = rng.standard_normal((V, D)).astype(np.float32)
        full = orc.forward(ro, keys.astype(np.uint64), dense, D, 0).reshape(B, S, D)
        bpg = B // world
        if emb_type == "localized":
            fro, fkeys = orc.localized_filter(ro, keys, B, S, rank, world)
            s_r = slots_on_rank(S, rank, world)
            assert s_r == orc.slots_on_gpu(S, rank, world)
            pooled = orc.forward(fro, fkeys.astype(np.uint64), dense, D, 0).reshape(B, s_r, D)
            ex = LocalizedExchange(B, S, D)
            recv = ex.forward(torch.from_numpy(pooled))
            got = orc.forward_reorder(recv.numpy(), bpg, S, D, world)
            assert np.array_equal(got, full[rank * bpg:(rank + 1) * bpg]), "forward exchange"
            # backward: top gradient of my sample slice -> owners of the slots
            g_full = rng.standard_normal((B, S, D)).astype(np.float32)  # same on all ranks
            my_grad = g_full[rank * bpg:(rank + 1) * bpg]
            gsend = orc.backward_reorder(my_grad, bpg, S, D, world)
            top = ex.backward(torch.from_numpy(gsend)).numpy()
            want = g_full[:, rank::world, :]
            assert np.array_equal(top, want), "backward exchange"
        else:
            fro, fkeys = orc.distributed_filter(ro, keys, B, S, rank, world)
            partial = orc.forward(fro, fkeys.astype(np.uint64), dense, D, 0).reshape(B, S, D)
            ex = DistributedExchange(B, S, D)
            out = ex.forward(torch.from_numpy(partial)).numpy()
            assert np.allclose(out, full[rank * bpg:(rank + 1) * bpg], rtol=1e-5, atol=1e-5)
            g = torch.full((bpg, S, D), float(rank + 1))
            allg = ex.backward(g).numpy()
            for r in range(world):
                assert (allg[r * bpg:(r + 1) * bpg] == r + 1).all()
        ret[rank] = "ok"
    except Exception as e:  # surface the failure in the parent
        import traceback
        ret[rank] = traceback.format_exc()
    finally:
        dist.destroy_process_group()


@pytest.mark.parametrize("emb_type", ["localized", "distributed"])
def test_exchange_world2_gloo(emb_type):
    world = 2
    port = 29500 + (os.getpid() % 2000) + (0 if emb_type == "localized" else 1)
    mgr = mp.Manager()
    ret = mgr.dict()
    mp.spawn(_worker, args=(world, port, emb_type, ret), nprocs=world, join=True)
    assert all(ret.get(r) == "ok" for r in range(world)), dict(ret)


def test_split_sizes_match_reference_counts():
    """send (B/N)*S_r*D to every peer, receive (B/N)*S_j*D from peer j
    (R/HugeCTR/src/embeddings/all2all_forward_functor.cu:157-264)"""
    from hugectr_amd.parallel import localized_split_sizes
    send, recv = localized_split_sizes(65536, 26, 128, 0, 8)
    assert send == [8192 * 4 * 128] * 8
    assert recv == [8192 * s * 128 for s in (4, 4, 3, 3, 3, 3, 3, 3)]
    send, recv = localized_split_sizes(65536, 26, 128, 5, 8)
    assert send == [8192 * 3 * 128] * 8


def _worker_chunked(rank, world, port, ret):
    """bench.py's N > 1 step: the global batch is cut into sub-batches; each sub-batch is a
    reference-layout global batch of its own, exchanged with the non-blocking all-to-all while the
    previous one is consumed.  Checks ownership labelling + async exchange in both directions."""
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        from oracle import pyoracle as orc
        from hugectr_amd.parallel import LocalizedExchange, slots_on_rank
        from util import make_csr
        rng = np.random.default_rng(5)
        C, Bc, S, D, vps = 3, 4, 5, 8, 11
        Bl = C * Bc
        B, Bsub = Bl * world, Bc * world
        ro, keys = make_csr(rng, B, S, 2, vps)
        dense = rng.standard_normal((S * vps, D)).astype(np.float32)
        full = orc.forward(ro, keys.astype(np.uint64), dense, D, 0).reshape(B, S, D)
        fro, fkeys = orc.localized_filter(ro, keys, B, S, rank, world)
        s_r = slots_on_rank(S, rank, world)
        pooled = torch.from_numpy(orc.forward(fro, fkeys.astype(np.uint64), dense, D, 0).reshape(B, s_r, D))
        ex = LocalizedExchange(Bsub, S, D)
        g_full = rng.standard_normal((B, S, D)).astype(np.float32)
        top_grad = torch.empty((B, s_r, D))
        recvs, works = [None] * C, [None] * C
        recvs[0], works[0] = ex.forward_async(pooled[0:Bsub])
        back = []
        for k in range(C):
            if k + 1 < C:
                recvs[k + 1], works[k + 1] = ex.forward_async(pooled[(k + 1) * Bsub:(k + 2) * Bsub])
            works[k].wait()
            E = orc.forward_reorder(recvs[k].numpy(), Bc, S, D, world)
            mine = slice(k * Bsub + rank * Bc, k * Bsub + (rank + 1) * Bc)  # samples I own in sub-batch k
            assert np.array_equal(E, full[mine]), f"sub-batch {k}"
            gsend = torch.from_numpy(orc.backward_reorder(g_full[mine], Bc, S, D, world))
            w = ex.backward_async(gsend, top_grad[k * Bsub:(k + 1) * Bsub].view(-1))
            back.append((w, gsend))
        for w, _ in back:
            w.wait()
        assert np.array_equal(top_grad.numpy(), g_full[:, rank::world, :]), "assembled top gradients"
        ret[rank] = "ok"
    except Exception:
        import traceback
        ret[rank] = traceback.format_exc()
    finally:
        dist.destroy_process_group()


def test_chunked_async_exchange_world2_gloo():
    world = 2
    port = 29500 + (os.getpid() % 2000) + 7
    mgr = mp.Manager()
    ret = mgr.dict()
    mp.spawn(_worker_chunked, args=(world, port, ret), nprocs=world, join=True)
    assert all(ret.get(r) == "ok" for r in range(world)), dict(ret)


def test_reorder_row_map_is_the_index_form_of_forward_and_backward_reorder():
    """hugectr.Model reads the all-to-all receive buffer through parallel.reorder_row_map (the
    interaction's indexed kernels) instead of running forward_reorder / backward_reorder: the map
    must gather exactly what the oracle's forward_reorder produces and scatter exactly what its
    backward_reorder produces, uneven slot counts included."""
    import numpy as np
    import torch
    from oracle import pyoracle as orc
    from hugectr_amd.parallel import reorder_row_map
    orc.build()
    rng = np.random.default_rng(0)
    for bpg, S, D, world in ((5, 26, 4, 8), (7, 26, 8, 4), (3, 7, 2, 2), (4, 3, 2, 3), (6, 5, 4, 1)):
        recv = rng.standard_normal(bpg * S * D).astype(np.float32)
        row = reorder_row_map(bpg, S, world).numpy()
        assert sorted(row.reshape(-1).tolist()) == list(range(bpg * S))  # a bijection
        want = orc.forward_reorder(recv, bpg, S, D, world).reshape(bpg, S, D)
        got = recv.reshape(-1, D)[row]
        assert (got == want).all()
        g = rng.standard_normal((bpg, S, D)).astype(np.float32)
        back = np.zeros((bpg * S, D), np.float32)
        back[row.reshape(-1)] = g.reshape(-1, D)
        assert (back.reshape(-1) == orc.backward_reorder(g.reshape(-1), bpg, S, D, world)).all()
