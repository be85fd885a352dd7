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

import os

from typing import List, Sequence

import torch

from ._lib import check, lib, ptr, stream_ptr

_DT16 = {torch.float16: 1, torch.bfloat16: 2}  # hctr_emb_dtype_t
_DT = {torch.float32: 0, torch.float16: 1, torch.bfloat16: 2}


def split_k_wgrad(dy: torch.Tensor, x: torch.Tensor, groups: int = 16,
                  out: torch.Tensor = None) -> torch.Tensor:
    """dW = dy^T @ x with the batch (K) dimension split into `groups` batched GEMMs (fp32 result,
    written to `out` when given)."""
    B, o = dy.shape
    i = x.shape[1]
    g = groups
    while g > 1 and B % g != 0:
        g //= 2
    if g <= 1 or min(o, i) < 8:
        # degenerate outputs (the logit layer, out = 1): the batched-GEMM path of the library
        # spends ~11 ms per call on the HOST for M = 1 (tools/gemm_probe2.py); one GEMV-like call
        # is 50-100 us
        r = dy.t() @ x
        if out is not None:
            out.copy_(r)
            return out
        return r
    p = torch.bmm(dy.view(g, B // g, o).transpose(1, 2), x.view(g, B // g, i))
    if p.is_cuda and p.dtype in _DT16 and (o * i) % 8 == 0:
        if out is None:
            out = torch.empty((o, i), dtype=torch.float32, device=p.device)
        check(lib.hctr_sum_groups(g, o * i, ptr(p), _DT16[p.dtype], ptr(out), stream_ptr()))
        return out
    r = p.float().sum(0)
    if out is not None:
        out.copy_(r)
        return out
    return r


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
    def forward(ctx, x, w_master, b_master, w16, b16, relu: bool, groups: int, gw=None, gb=None):
        """gw / gb: views of the module's flat gradient buffer; when given, backward writes the
        weight / bias gradients there and returns no gradient for the masters"""
        x = x.contiguous()
        ctx.gw, ctx.gb = gw, gb
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
            db = ctx.gb if ctx.gb is not None else torch.empty(n, dtype=torch.float32,
                                                             device=dy.device)
            ws = torch.empty(lib.hctr_relu_bwd_bias_workspace_bytes(dy.shape[0], n) // 4,
                             dtype=torch.float32, device=dy.device)
            check(lib.hctr_relu_bwd_bias(dy.shape[0], n, ptr(dy), ptr(y), ptr(dz), ptr(db), ptr(ws),
                                         _DT16[dy.dtype], stream_ptr()))
            dy = dz
        else:
            if ctx.relu:
                dy = torch.ops.aten.threshold_backward(dy, y, 0)
            db = dy.sum(0, dtype=torch.float32)
            if ctx.gb is not None:
                ctx.gb.copy_(db)
        dx = dy @ w16 if ctx.needs_input_grad[0] else None
        dw = split_k_wgrad(dy, x, ctx.groups, out=ctx.gw)
        if ctx.gw is not None:  # gradients live in the flat buffer; nothing for autograd to keep
            return dx, None, None, None, None, None, None, None, None
        return dx, dw.float(), db, None, None, None, None, None, None


class _SkinnyFirstFn(torch.autograd.Function):
    """y = relu(x @ W^T + b) for an input with a handful of features (x fp32 [B, K <= 16], no data
    gradient).  The backward (hctr_skinny_fc_bwd) reads dy and y once and leaves dw / db, no dz
    tensor in between; the forward is the library GEMM unless HCTR_SKINNY_FC_FWD=hip selects
    hctr_skinny_fc_fwd."""

    @staticmethod
    def forward(ctx, x, w_master, b_master, w16, b16, gw=None, gb=None):
        x = x.contiguous()
        B, K = x.shape
        N = w16.shape[0]
        if os.environ.get("HCTR_SKINNY_FC_FWD", "gemm") == "hip":
            y = torch.empty((B, N), dtype=w16.dtype, device=x.device)
            check(lib.hctr_skinny_fc_fwd(B, K, N, ptr(x), ptr(w16), ptr(b16), ptr(y),
                                         _DT16[w16.dtype], stream_ptr()))
        else:  # the library GEMM is the faster forward at 13 -> 512 (30 us against 43 us)
            y = torch._addmm_activation(b16, x.to(w16.dtype), w16.t(), use_gelu=False)
        ctx.gw, ctx.gb = gw, gb
        ctx.save_for_backward(x, y)
        return y

    @staticmethod
    def backward(ctx, dy):
        x, y = ctx.saved_tensors
        dy = dy.contiguous()
        B, K = x.shape
        N = y.shape[1]
        dev = x.device
        dw = ctx.gw if ctx.gw is not None else torch.empty((N, K), dtype=torch.float32, device=dev)
        db = ctx.gb if ctx.gb is not None else torch.empty(N, dtype=torch.float32, device=dev)
        ws = torch.empty(lib.hctr_skinny_fc_bwd_workspace_bytes(N) // 4, dtype=torch.float32,
                         device=dev)
        check(lib.hctr_skinny_fc_bwd(B, K, N, ptr(x), ptr(dy), ptr(y), ptr(dw), ptr(db), ptr(ws),
                                     _DT16[y.dtype], stream_ptr()))
        if ctx.gw is not None:
            return None, None, None, None, None, None, None
        return None, dw, db, None, None, None, None


class _LogitHeadFn(torch.autograd.Function):
    """loss = mean BCE(x @ w^T + b, label) with the whole backward of the head computed in the same
    pass (hctr_logit_head): dx, dw, db exist when forward returns.  backward() hands dx on; it
    assumes the unit upstream gradient of `loss.backward()` -- scale through grad_scale."""

    @staticmethod
    def forward(ctx, x, w_master, b_master, w16, b16, label, grad_scale: float, gw=None, gb=None):
        x = x.contiguous()
        B, K = x.shape
        dev = x.device
        dx = torch.empty_like(x) if ctx.needs_input_grad[0] else None
        dw = gw if gw is not None else torch.empty((1, K), dtype=torch.float32, device=dev)
        db = gb if gb is not None else torch.empty(1, dtype=torch.float32, device=dev)
        loss = torch.empty(1, dtype=torch.float32, device=dev)
        ws = torch.empty(lib.hctr_logit_head_workspace_bytes(K) // 4, dtype=torch.float32, device=dev)
        label = label.contiguous()
        check(lib.hctr_logit_head(B, K, ptr(x), ptr(w16), ptr(b16), ptr(label),
                                  float(grad_scale), ptr(dx), ptr(dw), ptr(db), ptr(loss), ptr(ws),
                                  _DT16[x.dtype], stream_ptr()))
        ctx.flat = gw is not None
        ctx.dx = dx
        ctx.dw, ctx.db = (None, None) if ctx.flat else (dw, db)
        return loss

    @staticmethod
    def backward(ctx, _grad_loss):
        dx, dw, db = ctx.dx, ctx.dw, ctx.db
        ctx.dx = ctx.dw = ctx.db = None
        return dx, dw, db, None, None, None, None, None, None


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
        self.flat_w = self.flat_g = self.flat_w16 = None
        self._gw: List[torch.Tensor] = []
        self._gb: List[torch.Tensor] = []

    def flatten(self):
        """Move masters, gradients and 16-bit copies into three flat buffers (parameters become
        views).  Afterwards backward writes gradients straight into `flat_g`, `sgd_step` is one
        kernel (update + shadow refresh), and a data-parallel all-reduce runs on `flat_g` as is."""
        dev = self.weights[0].device
        sizes = [p.numel() for p in list(self.weights) + list(self.biases)]
        offs, tot = [], 0
        for n in sizes:
            offs.append(tot)
            tot += (n + 3) // 4 * 4  # every view starts 16-byte aligned
        self.flat_w = torch.zeros(tot, dtype=torch.float32, device=dev)
        self.flat_g = torch.zeros(tot, dtype=torch.float32, device=dev)
        self.flat_w16 = torch.zeros(tot, dtype=self.dtype, device=dev)
        params = list(self.weights) + list(self.biases)
        views16 = []
        self._gw, self._gb = [], []
        for p, o, n in zip(params, offs, sizes):
            v = self.flat_w[o:o + n].view_as(p)
            v.copy_(p.data)
            p.data = v
            views16.append(self.flat_w16[o:o + n].view_as(p))
        nl = len(self.weights)
        self._w16, self._b16 = views16[:nl], views16[nl:]
        gviews = [self.flat_g[o:o + n].view_as(p) for p, o, n in zip(params, offs, sizes)]
        self._gw, self._gb = gviews[:nl], gviews[nl:]
        self.flat_w16.copy_(self.flat_w)
        return self

    def sgd_step(self, lr: float, grad_scale: float = 1.0):
        """w -= lr * grad_scale * g and refresh of the 16-bit copy, one launch (after flatten())"""
        check(lib.hctr_sgd_shadow(self.flat_w.numel(), float(lr), float(grad_scale),
                                  ptr(self.flat_w), ptr(self.flat_g), ptr(self.flat_w16),
                                  _DT16[self.dtype], stream_ptr()))

    def refresh_shadow(self):
        """call after every optimizer step (and once after .to(device))"""
        if self.flat_w is not None:
            self.flat_w16.copy_(self.flat_w)
            return
        if not self._w16:
            self._w16 = [w.detach().to(self.dtype) for w in self.weights]
            self._b16 = [b.detach().to(self.dtype) for b in self.biases]
        else:
            torch._foreach_copy_(self._w16 + self._b16,
                                 [w.detach() for w in self.weights] +
                                 [b.detach() for b in self.biases])

    def _skinny_first(self, x) -> bool:
        """first layer through the few-input-features kernels?  (fp32 input without a gradient)"""
        return (self.relu[0] and self.dims[0] <= 16 and self.dims[1] % 4 == 0 and
                self.dims[1] <= 512 and self.dtype in _DT16 and x.is_cuda and
                x.dtype == torch.float32 and not x.requires_grad and
                os.environ.get("HCTR_SKINNY_FC", "1") != "0")

    def _layer(self, i, x):
        gw = self._gw[i] if self._gw else None
        gb = self._gb[i] if self._gb else None
        if i == 0 and self._skinny_first(x):
            return _SkinnyFirstFn.apply(x, self.weights[0], self.biases[0], self._w16[0],
                                        self._b16[0], gw, gb)
        if x.dtype != self.dtype:
            x = x.to(self.dtype)
        return _LinearFn.apply(x, self.weights[i], self.biases[i], self._w16[i], self._b16[i],
                               self.relu[i], self.wgrad_groups, gw, gb)

    def can_fuse_bce_head(self) -> bool:
        k = self.dims[-2]
        return (self.dims[-1] == 1 and not self.relu[-1] and k % 4 == 0 and k <= 2048 and
                self.dtype in _DT16)

    def forward_bce(self, x, label, grad_scale: float):
        """the stack up to the last hidden layer, then logit layer + BinaryCrossEntropyLoss + the
        head's backward in one kernel: returns the mean loss [1]; `loss.backward()` continues
        through the hidden layers (gradients of the head are already in place)"""
        assert self.can_fuse_bce_head()
        if not self._w16:
            self.refresh_shadow()
        n = len(self.weights)
        for i in range(n - 1):
            x = self._layer(i, x)
        if x.dtype != self.dtype:
            x = x.to(self.dtype)
        return _LogitHeadFn.apply(x, self.weights[-1], self.biases[-1], self._w16[-1], self._b16[-1],
                                  label, grad_scale, self._gw[-1] if self._gw else None,
                                  self._gb[-1] if self._gb else None)

    def forward(self, x):
        if not self._w16:
            self.refresh_shadow()
        for i in range(len(self.weights)):
            x = self._layer(i, x)
        return x
