"""TEST INFRASTRUCTURE ONLY: the DLRM graph of tests/test_model_gpu.py trained on W ranks (gloo,
each rank the kernels' source under the host interpreter) against the SAME model, initial weights
and global batches on ONE rank: losses, every embedding vector (key -> vector over all ranks) and
the dense weights must agree to rounding -- for every exchange payload and with the overlapped
schedule.  26 slots on 4 / 8 ranks have uneven slot counts (7, 7, 6, 6 / 4, 4, 3, ...): the
driver's 8-GPU configuration, checked for its numbers rather than only for running.

    python tests/emu/parity_ranks.py --world 8 [--mixed] [--exchange rows,unique,unique16,auto]"""
import argparse
import os
import sys
import tempfile
from pathlib import Path

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--world", type=int, default=8)
    ap.add_argument("--mixed", action="store_true")
    ap.add_argument("--exchange", default="rows,unique,unique16,auto")
    ap.add_argument("--steps", type=int, default=20)
    a = ap.parse_args()
    os.environ["HCTR_EMU"] = "1"
    os.environ["PYTHONPATH"] = os.path.join(HERE, "site") + os.pathsep + ROOT + os.pathsep + \
        os.path.join(ROOT, "tests") + os.pathsep + os.environ.get("PYTHONPATH", "")
    for p in (ROOT, os.path.join(ROOT, "tests"), HERE):
        sys.path.insert(0, p)
    import fakecuda
    fakecuda.install()
    import torch
    import torch.multiprocessing as mp
    from numpy.testing import assert_allclose
    import hugectr_amd.hugectr as hugectr
    import test_model_gpu as T
    tmp = Path(tempfile.mkdtemp(prefix="parity_ranks_"))
    T._gen(tmp, hugectr, n_train=8192, n_eval=512)
    rng = np.random.default_rng(3)
    V = sum(T.SIZES)
    d = tmp / "init_sparse"
    d.mkdir()
    np.arange(V, dtype="<i8").tofile(d / "key")
    np.repeat(np.arange(26), T.SIZES).astype("<u8").tofile(d / "slot_id")
    (rng.standard_normal((V, 32)) * 0.1).astype("<f4").tofile(d / "emb_vector")
    # one rank: the reference result
    os.environ.pop("HCTR_EXCHANGE", None)
    torch.manual_seed(5)
    one = T._parity_run(T._overlap_model(hugectr, str(tmp), 1, False, a.mixed), a.steps)
    o1 = np.argsort(one[1])
    print("1 rank: loss", one[0][0], "->", one[0][-1], flush=True)
    bad = 0
    tol = dict(rtol=3e-2, atol=3e-3) if a.mixed else dict(rtol=5e-4, atol=5e-6)
    for salt, ex in enumerate(a.exchange.split(",")):
        ctx = mp.get_context("spawn")
        ret = ctx.Manager().dict()
        port = 25000 + os.getpid() % 3000 + salt * 7
        procs = [ctx.Process(target=T._overlap_worker,
                             args=(r, a.world, port, str(tmp), True, ex, a.mixed, a.steps, ret))
                 for r in range(a.world)]
        for p in procs:
            p.start()
        for p in procs:
            p.join(3000)
        try:
            for r in range(a.world):
                assert ret.get(r) is not None and ret[r][0] == "ok", (r, ret.get(r))
            loss = np.mean([ret[r][1] for r in range(a.world)], axis=0)
            assert_allclose(loss, np.array(one[0]), rtol=1e-2 if a.mixed else 5e-5)
            for r in range(a.world):
                assert_allclose(ret[r][4], one[3], **tol)          # dense weights
            k = np.concatenate([ret[r][2] for r in range(a.world)])
            v = np.concatenate([ret[r][3] for r in range(a.world)])
            assert len(np.unique(k)) == k.size == one[1].size
            o2 = np.argsort(k)
            assert (one[1][o1] == k[o2]).all()
            assert_allclose(v[o2], one[2][o1], **tol)
            pay = {ret[r][5]["payload"] for r in range(a.world)}
            assert len(pay) == 1, pay
            print(f"world {a.world} exchange {ex}: payload {pay.pop()}, loss {loss[0]:.6f} -> "
                  f"{loss[-1]:.6f}, tables and dense weights equal to one rank's "
                  f"({'mixed' if a.mixed else 'fp32'})", flush=True)
        except AssertionError as e:
            bad += 1
            print(f"FAIL world {a.world} exchange {ex}: {str(e)[:1500]}", flush=True)
    sys.exit(1 if bad else 0)


if __name__ == "__main__":
    main()
