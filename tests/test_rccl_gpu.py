"""RCCL readiness on a 1-GPU box: a process group of ONE rank on the `nccl` backend (= RCCL on
ROCm), every exchange of the multi-GPU path issued through the communicator as self-sends --
the all-to-all of LocalizedExchange (rows payload), reduce-scatter / all-gather of
DistributedExchange, the variable-size all-to-alls + all-gather of UniqueExchange, the dense
gradient all-reduce, and a hugectr.Model that trains under an initialised RCCL group.  What a real
8-GPU run adds is peers and xGMI, not code paths."""
import os

import numpy as np
import pytest

pytestmark = pytest.mark.gpu


def _worker(port, ret):
    import torch
    import torch.distributed as dist
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK="0", WORLD_SIZE="1",
                      LOCAL_RANK="0")
    torch.cuda.set_device(0)
    dist.init_process_group("nccl", rank=0, world_size=1, device_id=torch.device("cuda", 0))
    try:
        assert dist.get_backend() == "nccl"
        import hugectr_amd as ha
        from hugectr_amd import _lib
        from hugectr_amd.parallel import DistributedExchange, LocalizedExchange, all_reduce
        from hugectr_amd.unique_exchange import UniqueExchange
        B, S, D = 256, 5, 64
        # ---- rows payload: all-to-all forward / backward, blocking and asynchronous --------------
        lx = LocalizedExchange(B, S, D, always_collective=True)
        assert not lx.single
        x = torch.randn(B, S, D, device="cuda").to(torch.float16)
        y = lx.forward(x)
        assert y.data_ptr() != x.data_ptr() and torch.equal(y.view_as(x), x)
        y2, work = lx.forward_async(x)
        work.wait()
        assert torch.equal(y2.view_as(x), x)
        g = lx.backward(y)
        assert torch.equal(g, x)
        out = torch.empty(B * S * D, dtype=x.dtype, device="cuda")
        lx.backward_async(y, out).wait()
        assert torch.equal(out.view_as(x), x)
        # ---- distributed embedding: reduce-scatter / all-gather ---------------------------------
        dx = DistributedExchange(B, S, D, always_collective=True)
        p = torch.randn(B, S, D, device="cuda")
        r = dx.forward(p)
        assert r.data_ptr() != p.data_ptr() and torch.equal(r, p)
        assert torch.equal(dx.backward(r), p)
        # ---- dense gradients: all-reduce of a flat buffer ---------------------------------------
        flat = torch.randn(1 << 20, device="cuda")
        ref = flat.clone()
        all_reduce(flat)
        assert torch.equal(flat, ref)
        # ---- unique-row exchange over RCCL vs the local path, 3 power-law batches ---------------
        sizes = [7, 300, 3, 5000, 41]
        offs = np.concatenate([[0], np.cumsum(sizes)[:-1]])
        mk = lambda: ha.SparseEmbeddingHash(
            _lib.EMB_LOCALIZED, B, 0, sum(sizes), D, S, S, 0,
            ha.OptParams(optimizer=_lib.OPT_SGD, lr=0.05, atomic_update=False),
            slot_size_array=sizes, out_dtype=torch.float16, rank=0, world=1, seed=5)
        eu, ed = mk(), mk()
        eu.init_params()
        ed.init_params()
        ux = UniqueExchange(eu, B, S, D)
        ro = torch.arange(0, B * S + 1, dtype=torch.int64, device="cuda")
        rng = np.random.default_rng(3)
        for step in range(3):
            kt = torch.from_numpy(np.stack([(rng.zipf(1.3, size=B) - 1) % v + o
                                            for v, o in zip(sizes, offs)], 1).reshape(-1)).cuda()
            ux.forward_begin(ro, kt)
            E = ux.forward_finish()
            E_ref = ed.forward(True, ro, kt)
            if step == 0:
                assert torch.equal(E, E_ref)
            gr = torch.randn(B, S, D, device="cuda").to(torch.float16)
            ux.backward_and_update(gr)
            ed.backward(gr)
            ed.update_params()
            torch.cuda.synchronize()
            err = (eu.table() - ed.table()).abs().max().item()
            assert err <= 2e-5 * ed.table().abs().max().item() + 1e-6, (step, err)
        assert sum(ux.u_send) < ux.P
        ret["exchanges"] = "ok"
    except Exception as e:
        import traceback
        ret["exchanges"] = "".join(traceback.format_exception(type(e), e, e.__traceback__))
    finally:
        dist.destroy_process_group()


def test_every_exchange_runs_through_rccl_in_a_group_of_one():
    import torch.multiprocessing as mp
    ctx = mp.get_context("spawn")
    ret = ctx.Manager().dict()
    p = ctx.Process(target=_worker, args=(24000 + os.getpid() % 4000, ret))
    p.start()
    p.join(600)
    assert ret.get("exchanges") == "ok", ret.get("exchanges")
