"""Isolated timing of the few-features first-layer kernels (hctr_skinny_fc_fwd / _bwd) at the DLRM
shape (65536 x 13 -> 512, bf16), against the library GEMM forward.  usage: python tools/microbench_skinny.py"""
import json
import os
import sys

import torch

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
from hugectr_amd._lib import check, lib, ptr, stream_ptr  # noqa: E402


def timed(fn, n=50, warm=5):
    for _ in range(warm):
        fn()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(n):
        fn()
    b.record()
    torch.cuda.synchronize()
    return a.elapsed_time(b) / n * 1e3  # us


def main():
    B, K, N = 65536, 13, 512
    dev = "cuda"
    x = torch.rand((B, K), device=dev)
    w = (torch.randn((N, K), device=dev) / K ** 0.5).bfloat16()
    b = torch.zeros(N, device=dev).bfloat16()
    y = torch.empty((B, N), dtype=torch.bfloat16, device=dev)
    dy = (torch.randn((B, N), device=dev) / B).bfloat16()
    dw = torch.empty((N, K), dtype=torch.float32, device=dev)
    db = torch.empty(N, dtype=torch.float32, device=dev)
    ws = torch.empty(lib.hctr_skinny_fc_bwd_workspace_bytes(N) // 4, dtype=torch.float32, device=dev)
    x16 = x.bfloat16()
    res = {"shape": [B, K, N]}
    res["fwd_hip_us"] = timed(lambda: check(lib.hctr_skinny_fc_fwd(
        B, K, N, ptr(x), ptr(w), ptr(b), ptr(y), 2, stream_ptr())))
    res["fwd_gemm_us"] = timed(lambda: torch._addmm_activation(b, x16, w.t(), use_gelu=False))
    bwd = lambda: check(lib.hctr_skinny_fc_bwd(  # noqa: E731
        B, K, N, ptr(x), ptr(dy), ptr(y), ptr(dw), ptr(db), ptr(ws), 2, stream_ptr()))
    os.environ["HCTR_SKINNY_BWD"] = "valu"
    res["bwd_hip_us (incl. finish)"] = timed(bwd)
    dw_v, db_v = dw.clone(), db.clone()
    os.environ["HCTR_SKINNY_BWD"] = "mfma"
    dw.zero_()
    db.zero_()
    res["bwd_mfma_us (incl. finish)"] = timed(bwd)
    res["mfma_vs_valu_rel_err"] = [float((dw - dw_v).norm() / dw_v.norm()),
                                   float((db - db_v).norm() / db_v.norm())]
    dz = dy.double() * (y > 0)
    res["mfma_vs_fp64_rel_err"] = [float((dw.double() - dz.t() @ x16.double()).norm() /
                                         (dz.t() @ x16.double()).norm()),
                                   float((db.double() - dz.sum(0)).norm() / dz.sum(0).norm())]
    res["fwd_store_GBps"] = B * N * 2 / res["fwd_hip_us"] / 1e3
    res["bwd_read_GBps"] = B * N * 4 / res["bwd_hip_us (incl. finish)"] / 1e3
    res["fma_per_launch"] = B * N * 16
    print(json.dumps(res))


if __name__ == "__main__":
    main()
