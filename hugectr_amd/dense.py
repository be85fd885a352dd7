"""Dense-tower building blocks (PyTorch-ROCm / hipBLASLt GEMMs) arranged for MI355X.

The MLP GEMMs themselves stay in the vendor library (dense tower out of scope as
kernels).  What this module fixes is how they are *issued* for the DLRM shapes
(`FusedFullyConnectedLayer` / `MLPLayer` in the reference, R/HugeCTR/src/layers/mlp_layer.cu):

* weight gradients dW[out,in] = dY^T X reduce over K = batch = 65536 into a small out x in tile
  set: a single library GEMM launches 64-128 workgroups on 256 CUs.  `split_k_wgrad` reshapes the
  reduction into G batched GEMMs (torch.bmm) + a sum, i.e. split-K through the library: 2-5x faster
  on these shapes (tools/gemm_probe.py).
* bias + ReLU ride in the GEMM epilogue (`torch._addmm_activation` -> hipBLASLt RELU_BIAS).
* bf16 shadow copies of the fp32 master weights are refreshed once per optimizer step with one
  multi-tensor copy instead of one cast kernel per layer per pass.
"""
from __future__ import annotations

from typing import List, Sequence

import torch

from ._lib import check, lib, ptr, stream_ptr

_DT16 = {torch.float16: 1, torch.bfloat16: 2}  # hctr_emb_dtype_t
_DT = {torch.float32: 0, torch.float16: 1, torch.bfloat16: 2}


def split_k_wgrad(dy: torch.Tensor, x: torch.Tensor, groups: int = 16) -> torch.Tensor:
    """dW = dy^T @ x with the batch (K) dimension split into `groups` batched GEMMs."""
    B, o = dy.shape
    i = x.shape[1]
    g = groups
    while g > 1 and B % g != 0:
        g //= 2
    if g <= 1 or min(o, i) < 8:
        # degenerate outputs (the logit layer, out = 1): the batched-GEMM path of the library
        # spends ~11 ms per call on the HOST for M = 1 (tools/gemm_probe2.py); one GEMV-like call
        # is 50-100 us
        return dy.t() @ x
    p = torch.bmm(dy.view(g, B // g, o).transpose(1, 2), x.view(g, B // g, i))
    if p.is_cuda and p.dtype in _DT16 and (o * i) % 8 == 0:
        out = torch.empty((o, i), dtype=torch.float32, device=p.device)
        check(lib.hctr_sum_groups(g, o * i, ptr(p), _DT16[p.dtype], ptr(out), stream_ptr()))
        return out
    return p.float().sum(0)


def bce_with_logits(logit: torch.Tensor, label: torch.Tensor, grad_scale: float):
    """BinaryCrossEntropyLoss forward + logit gradient in one pass (R/HugeCTR/src/loss.cu:231-262).

    Returns (mean loss [1] fp32, dlogit like `logit`) with dlogit = (sigmoid(x) - y) * grad_scale;
    feed it to `logit.backward(dlogit)`."""
    x = logit.detach().contiguous()
    y = label.contiguous()
    assert x.is_cuda and y.dtype == torch.float32 and x.numel() == y.numel()
    dlogit = torch.empty_like(x)
    loss = torch.empty(1, dtype=torch.float32, device=x.device)
    ws = torch.empty(lib.hctr_bce_loss_workspace_bytes() // 4, dtype=torch.float32, device=x.device)
    check(lib.hctr_bce_loss(x.numel(), ptr(x), ptr(y), float(grad_scale), ptr(dlogit), ptr(loss),
                            ptr(ws), _DT[x.dtype], stream_ptr()))
    return loss, dlogit


class _LinearFn(torch.autograd.Function):
    """y = act(x @ W^T + b) on 16-bit shadow weights; gradients returned for the fp32 masters."""

    @staticmethod
    def forward(ctx, x, w_master, b_master, w16, b16, relu: bool, groups: int):
        x = x.contiguous()
        if relu:
            y = torch._addmm_activation(b16, x, w16.t(), use_gelu=False)
        else:
            y = torch.addmm(b16, x, w16.t())
        ctx.relu, ctx.groups = relu, groups
        ctx.save_for_backward(x, w16, y if relu else None)
        return y

    @staticmethod
    def backward(ctx, dy):
        x, w16, y = ctx.saved_tensors
        dy = dy.contiguous()
        n = dy.shape[1]
        if ctx.relu and n % 8 == 0 and dy.dtype in _DT16 and dy.is_cuda:
            # fused ReLU backward + bias gradient (HIP): one pass instead of two
            dz = torch.empty_like(dy)
            db = torch.empty(n, dtype=torch.float32, device=dy.device)
            ws = torch.empty(lib.hctr_relu_bwd_bias_workspace_bytes(dy.shape[0], n) // 4,
                             dtype=torch.float32, device=dy.device)
            check(lib.hctr_relu_bwd_bias(dy.shape[0], n, ptr(dy), ptr(y), ptr(dz), ptr(db), ptr(ws),
                                         _DT16[dy.dtype], stream_ptr()))
            dy = dz
        else:
            if ctx.relu:
                dy = torch.ops.aten.threshold_backward(dy, y, 0)
            db = dy.sum(0, dtype=torch.float32)
        dx = dy @ w16 if ctx.needs_input_grad[0] else None
        dw = split_k_wgrad(dy, x, ctx.groups)
        return dx, dw.float(), db, None, None, None, None


class FusedMLP(torch.nn.Module):
    """Stack of Linear(+ReLU) layers with fp32 master weights and 16-bit compute copies.

    dims = [in, h1, ..., out]; `last_relu` tells whether the final layer is activated (the DLRM
    bottom MLP is, the top MLP's logit layer is not)."""

    def __init__(self, dims: Sequence[int], last_relu: bool, dtype=torch.bfloat16,
                 wgrad_groups: int = 16):
        super().__init__()
        self.dims = list(dims)
        self.dtype = dtype
        self.wgrad_groups = wgrad_groups
        self.relu = [True] * (len(dims) - 2) + [last_relu]
        self.weights = torch.nn.ParameterList()
        self.biases = torch.nn.ParameterList()
        for i in range(len(dims) - 1):
            lin = torch.nn.Linear(dims[i], dims[i + 1])  # reference default init equivalent
            self.weights.append(torch.nn.Parameter(lin.weight.detach().clone()))
            self.biases.append(torch.nn.Parameter(lin.bias.detach().clone()))
        self._w16: List[torch.Tensor] = []
        self._b16: List[torch.Tensor] = []

    def refresh_shadow(self):
        """call after every optimizer step (and once after .to(device))"""
        if not self._w16:
            self._w16 = [w.detach().to(self.dtype) for w in self.weights]
            self._b16 = [b.detach().to(self.dtype) for b in self.biases]
        else:
            torch._foreach_copy_(self._w16 + self._b16,
                                 [w.detach() for w in self.weights] +
                                 [b.detach() for b in self.biases])

    def forward(self, x):
        if not self._w16:
            self.refresh_shadow()
        x = x.to(self.dtype)
        for i in range(len(self.weights)):
            x = _LinearFn.apply(x, self.weights[i], self.biases[i], self._w16[i], self._b16[i],
                                self.relu[i], self.wgrad_groups)
        return x
