"""Probe: weight-gradient GEMM dW = dY^T X with K = batch = 65536 -- library call vs split-K via bmm,
and _addmm_activation availability."""
import torch
torch.manual_seed(0)
dev = "cuda"
B = 65536
def t(fn, n=20):
    for _ in range(3): fn()
    torch.cuda.synchronize(); s = torch.cuda.Event(True); e = torch.cuda.Event(True)
    s.record()
    for _ in range(n): fn()
    e.record(); torch.cuda.synchronize()
    return s.elapsed_time(e) / n * 1e3
for (o, i) in [(1024, 480), (1024, 1024), (512, 1024), (256, 512), (512, 13), (256, 512), (128, 256)]:
    dY = torch.randn(B, o, device=dev, dtype=torch.bfloat16)
    X = torch.randn(B, i, device=dev, dtype=torch.bfloat16)
    ref = dY.t() @ X
    base = t(lambda: dY.t() @ X)
    line = f"dW[{o},{i}] K={B}: lib {base:7.1f}us"
    for G in (2, 4, 8, 16):
        def f():
            p = torch.bmm(dY.view(G, B // G, o).transpose(1, 2), X.view(G, B // G, i))
            return p.sum(0)
        def f32():
            p = torch.bmm(dY.view(G, B // G, o).transpose(1, 2), X.view(G, B // G, i)).float()
            return p.sum(0)
        us = t(f)
        err = (f().float() - ref.float()).abs().max().item() / ref.float().abs().max().item()
        line += f" | G={G}: {us:7.1f}us (rel {err:.1e})"
    print(line, f" flops={2*B*o*i/1e9:.1f}G")
x = torch.randn(B, 1024, device=dev, dtype=torch.bfloat16); W = torch.randn(1024, 1024, device=dev, dtype=torch.bfloat16); b = torch.randn(1024, device=dev, dtype=torch.bfloat16)
try:
    y = torch._addmm_activation(b, x, W.t(), use_gelu=False)
    ref = torch.relu(torch.addmm(b, x, W.t()))
    print("_addmm_activation ok, max diff", (y.float() - ref.float()).abs().max().item(),
          "fused", t(lambda: torch._addmm_activation(b, x, W.t(), use_gelu=False)),
          "unfused", t(lambda: torch.relu_(torch.addmm(b, x, W.t()))))
except Exception as ex:
    print("_addmm_activation failed:", ex)
