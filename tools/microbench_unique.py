"""Device-side cost of the unique-row exchange for ONE rank of an N-GPU job, measured on one GPU
(no communication): index stage, plan, distinct-row gather, expand, presorted reduce, row update.
The receiver side is fed with this rank's own plan output replicated N times (every owner looks
alike at this level).  Usage: python tools/microbench_unique.py [--world 8]"""
import argparse
import ctypes
import os
import sys

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import hugectr_amd as ha  # noqa: E402
from hugectr_amd import _lib  # noqa: E402
from hugectr_amd._lib import check, lib, ptr, stream_ptr  # noqa: E402
from microbench_embedding import CRITEO_1TB, powerlaw  # noqa: E402


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
    ap = argparse.ArgumentParser()
    ap.add_argument("--world", type=int, default=8)
    ap.add_argument("--rank", type=int, default=0)
    ap.add_argument("--batch", type=int, default=65536)
    a = ap.parse_args()
    W, r, Bl, D = a.world, a.rank, a.batch, 128
    S = len(CRITEO_1TB)
    B = Bl * W
    mine = [i for i in range(S) if i % W == r]
    s_r = len(mine)
    rows_r = sum(CRITEO_1TB[i] for i in mine)
    emb = ha.SparseEmbeddingHash(_lib.EMB_LOCALIZED, B, 0, rows_r, D, S, S, 0,
                                 ha.OptParams(optimizer=_lib.OPT_SGD, lr=0.01, atomic_update=False),
                                 slot_size_array=CRITEO_1TB, out_dtype=torch.bfloat16, rank=r, world=W)
    emb.init_params()
    rng = np.random.default_rng(3)
    offs = np.concatenate([[0], np.cumsum(CRITEO_1TB)[:-1]]).astype(np.int64)
    keys = np.empty((B, S), dtype=np.int64)
    for s, v in enumerate(CRITEO_1TB):
        keys[:, s] = powerlaw(rng, B, v, 1.1) + offs[s]
    kt = torch.from_numpy(keys.reshape(-1)).cuda()
    ro = torch.arange(0, B * S + 1, dtype=torch.int64, device="cuda")
    P, ppp, Q = W * Bl * s_r, Bl * s_r, Bl * S
    res = {"world": W, "slots_on_rank": s_r, "positions": P}
    res["index_us"] = timed(lambda: emb.index(True, ro, kt))
    pooled = torch.empty((B, s_r, D), dtype=torch.bfloat16, device="cuda")
    res["dense_forward_us(index+gather)"] = timed(lambda: emb.forward(True, ro, kt, out=pooled))
    h = ctypes.c_void_p()
    check(lib.hctr_uniq_create(P, ctypes.byref(h)))
    meta = torch.empty((P, 2), dtype=torch.int32, device="cuda")
    urow = torch.empty(P, dtype=torch.int64, device="cuda")
    poff = torch.zeros(W + 1, dtype=torch.int64, device="cuda")
    vi = emb.value_index(P)

    def plan():
        check(lib.hctr_uniq_plan(h, P, ppp, Bl, s_r, S, r, W, ptr(vi), rows_r, ptr(meta), ptr(urow),
                                 ptr(poff), stream_ptr()))
    res["plan_us"] = timed(plan)
    cnt = poff.cpu().numpy()
    U = int(cnt[-1])
    res["distinct_rows_total"] = U
    res["distinct_rows_per_peer"] = [int(x) for x in np.diff(cnt)]
    res["payload_MB_rows_vs_unique"] = [P * D * 2 / 1e6, (U * D * 2 + P * 8) / 1e6]
    send = torch.empty((U, D), dtype=torch.bfloat16, device="cuda")
    res["gather_rows_us"] = timed(lambda: check(lib.hctr_uniq_gather_rows(
        U, D, ptr(urow), lib.hctr_emb_table_ptr(emb._h), ptr(send), _lib.BF16, stream_ptr())))
    # receiver: pretend W owners each sent what this rank sends to peer 0 (slots_on_rank each)
    u0 = int(cnt[1])
    q_off = torch.arange(0, W + 1, dtype=torch.int64, device="cuda") * ppp
    Qe = W * ppp  # positions received in this emulation (S_r * W slots)
    S_e = s_r * W
    r_off = torch.arange(0, W + 1, dtype=torch.int64, device="cuda") * u0
    meta_r = meta[:ppp].repeat(W, 1).contiguous()
    # buckets of owner j: b_local * S_e + (s_local * W + j)
    bl = (meta[:ppp, 1] // S).to(torch.int64)
    sl = ((meta[:ppp, 1] % S) // W).to(torch.int64)
    for j in range(W):
        meta_r[j * ppp:(j + 1) * ppp, 1] = (bl * S_e + sl * W + j).to(torch.int32)
    rows_r_buf = send[:u0].repeat(W, 1).contiguous()
    E = torch.empty((Bl, S_e, D), dtype=torch.bfloat16, device="cuda")
    srow = torch.empty(Qe, dtype=torch.int32, device="cuda")
    sbkt = torch.empty(Qe, dtype=torch.int32, device="cuda")
    res["expand_us"] = timed(lambda: check(lib.hctr_uniq_expand(
        Qe, W, ptr(q_off), ptr(r_off), ptr(meta_r), ptr(rows_r_buf), D, _lib.BF16, ptr(E), ptr(srow),
        ptr(sbkt), None, stream_ptr())))
    upd = ctypes.c_void_p()
    check(lib.hctr_updater_create(Qe, Qe, D, ctypes.byref(upd)))
    ar = torch.arange(max(Qe, U) + 1, dtype=torch.int64, device="cuda")
    dE = torch.randn((Bl, S_e, D), device="cuda").bfloat16()
    sums = torch.empty((W * u0, D), dtype=torch.float32, device="cuda")
    res["reduce_presorted_us"] = timed(lambda: check(lib.hctr_updater_reduce_presorted(
        upd, Qe, Qe, ptr(ar), ptr(srow), ptr(sbkt), ptr(dE), _lib.BF16, W * u0, ptr(sums),
        stream_ptr())))
    back = torch.randn((U, D), device="cuda")
    res["update_rows_us"] = timed(lambda: emb.update_rows(urow[:U], back, ar[:U + 1]))
    res["dense_update_us"] = None
    emb.forward(True, ro, kt, out=pooled)
    g = torch.randn((B, s_r, D), device="cuda").bfloat16()

    def dense_upd():
        emb.forward(True, ro, kt, out=pooled)
        emb.backward(g)
        emb.update_params()
    res["dense_forward+update_us"] = timed(dense_upd)
    import json
    print(json.dumps(res))


if __name__ == "__main__":
    main()
