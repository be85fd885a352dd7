"""16-bit interaction forward / backward at the bench shape (B = 65536, 26 + 1 inputs, W = 128)."""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from hugectr_amd.layers import InteractionLayer  # noqa: E402


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


def main():
    B, n, W = 65536, 26, 128
    dt = torch.float16
    mlp = torch.randn(B, W, device="cuda").to(dt).requires_grad_(True)
    emb = torch.randn(B, n, W, device="cuda").to(dt).requires_grad_(True)
    layer = InteractionLayer()
    out = layer(mlp, emb)
    g = torch.randn_like(out)
    fwd = timed(lambda: layer(mlp, emb))

    def both():
        o = layer(mlp, emb)
        o.backward(g)
        mlp.grad = emb.grad = None
    fb = timed(both)
    nb = B * ((n + 1) * W + W + n * (n + 1) // 2 + 1) * 2
    print({"waves": os.environ.get("HCTR_INTER_WAVES", "8,8"), "fwd_us": round(fwd, 1),
           "fwd_TBps": round(nb / fwd / 1e6, 2), "fwd+bwd_us": round(fb, 1), "bwd_us": round(fb - fwd, 1)})


if __name__ == "__main__":
    main()
