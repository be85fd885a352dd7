"""interaction over the expanded tensor vs over (distinct rows, row index) at the DLRM shape"""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import hugectr_amd as ha  # noqa: E402


def timed(fn, it=20):
    for _ in range(3):
        fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(it):
        fn()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / it * 1e3


B, n, W, R = 65536, 26, 128, 190000
g = torch.Generator(device="cuda").manual_seed(0)
rows = torch.randn(R, W, device="cuda", generator=g).bfloat16()
# power-law-ish reuse: most positions hit a small set of hot rows
u = torch.rand(B, n, device="cuda", generator=g)
row_of = (u.pow(6) * R).to(torch.int32).clamp_(0, R - 1)
mlp = torch.randn(B, W, device="cuda", generator=g).bfloat16()
emb = rows[row_of.long()].contiguous()
top = torch.randn(B, W + 27 * 26 // 2 + 1, device="cuda", generator=g).bfloat16()
res = {}
res["dense_fwd_us"] = timed(lambda: ha.interaction(mlp, emb))
res["indexed_fwd_us"] = timed(lambda: ha.interaction_indexed(mlp, rows, row_of))


def fb_dense():
    m, e = mlp.clone().requires_grad_(), emb.clone().requires_grad_()
    ha.interaction(m, e).backward(top)


def fb_idx():
    m = mlp.clone().requires_grad_()
    ha.interaction_indexed(m, rows, row_of, on_emb_grad=lambda d: None).backward(top)


res["dense_fwd+bwd_us(+2 clones)"] = timed(fb_dense)
res["indexed_fwd+bwd_us(+1 clone)"] = timed(fb_idx)
print({k: round(v, 1) for k, v in res.items()})
