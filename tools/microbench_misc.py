"""Throughput of the paths bench.py does not exercise: MultiCross v1 (DCN shape), SOK lookup_sparse
(static variable, multi-hot), distributed-slot embedding forward."""
import os
import sys

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import hugectr_amd as ha  # noqa: E402
from hugectr_amd import _lib, sok  # noqa: E402


def timed(fn, it=10):
    for _ in range(2):
        fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(it):
        fn()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / it * 1e3


def main():
    # --- MultiCross v1, DCN sample shape: width 13 + 26*16 = 429, 6 layers
    B, w, L = 16384, 429, 6
    layer = ha.MultiCrossLayer(w, L).cuda()
    x = torch.randn(B, w, device="cuda", requires_grad=True)
    out = layer(x)
    g = torch.randn_like(out)
    fwd = timed(lambda: layer(x))

    def fb():
        x.grad = None
        layer(x).backward(g)
    both = timed(fb)
    print({"op": "cross_v1", "B": B, "w": w, "layers": L, "fwd_us": round(fwd, 1),
           "fwd+bwd_us": round(both, 1),
           "fwd_GBps(B*w*4*(1+L))": round(B * w * 4 * (1 + L) / fwd / 1e3, 1)})
    # --- SOK static variable, multi-hot mean with / without weights
    sok.init()
    V, D, Bs, hot = 2_000_000, 128, 65536, 10
    var = sok.Variable(torch.randn(V, D))
    rng = np.random.default_rng(0)
    lens = rng.integers(0, 2 * hot + 1, size=Bs)
    vals = torch.from_numpy(rng.integers(0, V, size=int(lens.sum()))).cuda()
    ids = sok.Ragged(vals, torch.from_numpy(lens).cuda())
    wts = sok.Ragged(torch.rand(vals.numel(), device="cuda") + 0.1, ids.row_lengths)
    opt = sok.OptimizerWrapper("sgd", lr=0.01)
    for name, w_ in (("mean", None), ("weighted mean", wts)):
        f = timed(lambda: sok.lookup_sparse(var, ids, w_, "mean"))
        go = torch.randn(Bs, D, device="cuda")

        def step():
            o = sok.lookup_sparse(var, ids, w_, "mean")
            o.backward(go)
            opt.step([var])
        s = timed(step)
        print({"op": f"sok.lookup_sparse {name}", "nnz": vals.numel(), "D": D,
               "fwd_us": round(f, 1), "fwd+bwd+sgd_us": round(s, 1),
               "fwd_GBps": round(vals.numel() * (D * 4 + 8) / f / 1e3, 1)})
    # --- distributed-slot embedding forward (key % N), one rank of 1
    S, Dd, Bd = 26, 16, 16384
    emb = ha.SparseEmbeddingHash(_lib.EMB_DISTRIBUTED, Bd, 0, 2_000_000, Dd, S, S, 0,
                                 ha.OptParams(optimizer=_lib.OPT_ADAM, lr=0.001))
    emb.init_params()
    keys = torch.from_numpy(rng.integers(0, 1_200_000, size=Bd * S)).cuda()
    ro = torch.arange(0, Bd * S + 1, dtype=torch.int64, device="cuda")
    emb.forward(True, ro, keys)
    gg = torch.randn(Bd, S, Dd, device="cuda")

    def dstep():
        emb.forward(True, ro, keys)
        emb.backward(gg)
        emb.update_params()
    print({"op": "distributed hash D=16 (DeepFM shape) fwd", "us": round(timed(lambda: emb.forward(True, ro, keys)), 1),
           "fwd+adam_us": round(timed(dstep), 1)})


if __name__ == "__main__":
    main()
