"""K = 480 vs 512 for the first top-MLP layer (interaction output width): forward, dgrad, wgrad."""
import torch

def t(fn, it=20):
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

B, N = 65536, 1024
for K in (480, 512):
    x = torch.randn(B, K, device="cuda", dtype=torch.bfloat16)
    w = torch.randn(N, K, device="cuda", dtype=torch.bfloat16)
    b = torch.randn(N, device="cuda", dtype=torch.bfloat16)
    dy = torch.randn(B, N, device="cuda", dtype=torch.bfloat16)
    fwd = t(lambda: torch._addmm_activation(b, x, w.t(), use_gelu=False))
    dg = t(lambda: dy @ w)
    g = 16
    wg = t(lambda: torch.bmm(dy.view(g, B // g, N).transpose(1, 2), x.view(g, B // g, K)))
    print(K, "fwd %.1f dgrad %.1f wgrad(bmm16) %.1f us" % (fwd, dg, wg))
