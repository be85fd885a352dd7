"""torch.autograd wrappers of the dense ops on the path (HIP kernels through the C ABI).

InteractionLayer: R/HugeCTR/include/layers/interaction_layer.hpp:26-67
MultiCrossLayer:  R/HugeCTR/include/layers/multi_cross_layer.hpp:104-177
"""
from __future__ import annotations

import os

import torch

from . import _lib
from ._lib import check, lib, ptr, stream_ptr
from .dense import _DT16, split_k_wgrad

_DT = {torch.float32: _lib.F32, torch.float16: _lib.F16, torch.bfloat16: _lib.BF16}


class _InteractionFn(torch.autograd.Function):
    @staticmethod
    def forward(ctx, mlp, emb):
        mlp, emb = mlp.contiguous(), emb.contiguous()
        B, W = mlp.shape
        n_emb = emb.shape[1]
        n_ins = n_emb + 1
        out = torch.empty((B, W + n_ins * (n_ins - 1) // 2 + 1), dtype=mlp.dtype, device=mlp.device)
        check(lib.hctr_interaction_fwd(B, n_emb, W, ptr(mlp), ptr(emb), ptr(out), _DT[mlp.dtype],
                                       stream_ptr()))
        ctx.save_for_backward(mlp, emb)
        return out

    @staticmethod
    def backward(ctx, grad):
        mlp, emb = ctx.saved_tensors
        grad = grad.contiguous()
        B, W = mlp.shape
        n_emb = emb.shape[1]
        mlp_grad = torch.empty_like(mlp)
        emb_grad = torch.empty_like(emb)
        check(lib.hctr_interaction_bwd(B, n_emb, W, ptr(mlp), ptr(emb), ptr(grad), ptr(mlp_grad),
                                       ptr(emb_grad), _DT[mlp.dtype], stream_ptr()))
        return mlp_grad, emb_grad


class _InteractionIndexedFn(torch.autograd.Function):
    """interaction over a table of distinct rows (unique-row exchange): embedding s of sample b is
    rows[row_of[b, s]].  `on_emb_grad(dE)` receives the dense [B, n_emb, W] embedding gradient as
    soon as it exists (the exchange reduces it per row); rows itself gets no gradient here."""

    @staticmethod
    def forward(ctx, mlp, rows, row_of, on_emb_grad, scatter_grad=False):
        mlp = mlp.contiguous()
        ctx.scatter_grad = scatter_grad
        B, W = mlp.shape
        n_emb = row_of.shape[1]
        n_ins = n_emb + 1
        out = torch.empty((B, W + n_ins * (n_ins - 1) // 2 + 1), dtype=mlp.dtype, device=mlp.device)
        check(lib.hctr_interaction_fwd_indexed(B, n_emb, W, ptr(mlp), ptr(rows), ptr(row_of),
                                               ptr(out), _DT[mlp.dtype], stream_ptr()))
        ctx.save_for_backward(mlp, rows, row_of)
        ctx.on_emb_grad = on_emb_grad
        return out

    @staticmethod
    def backward(ctx, grad):
        mlp, rows, row_of = ctx.saved_tensors
        grad = grad.contiguous()
        B, W = mlp.shape
        n_emb = row_of.shape[1]
        mlp_grad = torch.empty_like(mlp)
        emb_grad = torch.empty((B, n_emb, W), dtype=mlp.dtype, device=mlp.device)
        fn = lib.hctr_interaction_bwd_indexed_scatter if ctx.scatter_grad else \
            lib.hctr_interaction_bwd_indexed
        check(fn(B, n_emb, W, ptr(mlp), ptr(rows), ptr(row_of), ptr(grad), ptr(mlp_grad),
                 ptr(emb_grad), _DT[mlp.dtype], stream_ptr()))
        if ctx.on_emb_grad is not None:
            ctx.on_emb_grad(emb_grad)
        return mlp_grad, None, None, None, None


class _InteractionGatherFn(torch.autograd.Function):
    """gather fused into the interaction (one GPU, one key per bucket): the embedding's table rows
    are read through its value_index straight into the interaction's LDS tile, the pooled vectors
    are written once for the backward.  `on_emb_grad(dE)` receives the [B, n_emb, W] gradient of
    the pooled vectors as soon as it exists (the sparse update consumes it)."""

    @staticmethod
    def forward(ctx, mlp, emb, is_train, on_emb_grad):
        mlp = mlp.contiguous()
        B, W = mlp.shape
        n_emb = emb.slot_num
        n_ins = n_emb + 1
        out = torch.empty((B, W + n_ins * (n_ins - 1) // 2 + 1), dtype=mlp.dtype, device=mlp.device)
        pooled = torch.empty((B, n_emb, W), dtype=mlp.dtype, device=mlp.device)
        check(lib.hctr_emb_forward_interaction(emb._h, 1 if is_train else 0, ptr(mlp), ptr(pooled),
                                               ptr(out), stream_ptr()))
        ctx.save_for_backward(mlp, pooled)
        ctx.on_emb_grad = on_emb_grad
        return out

    @staticmethod
    def backward(ctx, grad):
        mlp, pooled = ctx.saved_tensors
        grad = grad.contiguous()
        B, W = mlp.shape
        n_emb = pooled.shape[1]
        mlp_grad = torch.empty_like(mlp)
        emb_grad = torch.empty_like(pooled)
        check(lib.hctr_interaction_bwd(B, n_emb, W, ptr(mlp), ptr(pooled), ptr(grad), ptr(mlp_grad),
                                       ptr(emb_grad), _DT[mlp.dtype], stream_ptr()))
        if ctx.on_emb_grad is not None:
            ctx.on_emb_grad(emb_grad)
        return mlp_grad, None, None, None


def interaction_gather(mlp: torch.Tensor, emb, is_train: bool = True, on_emb_grad=None):
    """mlp [B, W] 16-bit; emb: a SparseEmbeddingHash (world 1, one key per bucket, vector size W,
    output type = mlp's) whose index stage has run (emb.index) -> interaction output; the pooled
    vectors never make a second trip through HBM."""
    assert mlp.dtype == emb.out_dtype and mlp.shape[1] == emb.embedding_vec_size
    return _InteractionGatherFn.apply(mlp, emb, is_train, on_emb_grad)


def interaction_indexed(mlp: torch.Tensor, rows: torch.Tensor, row_of: torch.Tensor,
                        on_emb_grad=None, scatter_grad: bool = False) -> torch.Tensor:
    """mlp [B,W], rows [R,W] (same 16-bit dtype), row_of int32 [B,n_emb] -> interaction output.
    scatter_grad (row_of must be a bijection onto the B * n_emb rows, e.g. the reorder map of an
    all-to-all receive buffer): on_emb_grad receives the embedding gradient in the ROWS' layout
    (row row_of[b, s] = gradient of embedding s of sample b) instead of [B, n_emb, W]"""
    assert rows.dtype == mlp.dtype and row_of.dtype == torch.int32 and rows.is_contiguous()
    assert not scatter_grad or rows.shape[0] == row_of.numel()
    return _InteractionIndexedFn.apply(mlp, rows, row_of.contiguous(), on_emb_grad, scatter_grad)


def interaction(mlp: torch.Tensor, emb: torch.Tensor) -> torch.Tensor:
    """mlp [B,W], emb [B,n_emb,W] -> [B, W + n_ins(n_ins-1)/2 + 1] (last column zero)."""
    return _InteractionFn.apply(mlp, emb)


class InteractionLayer(torch.nn.Module):
    def forward(self, mlp, emb):
        return interaction(mlp, emb)


class _CrossV1Fn(torch.autograd.Function):
    @staticmethod
    def forward(ctx, x0, kernels, biases):
        x0 = x0.contiguous().float()
        kernels, biases = kernels.contiguous(), biases.contiguous()
        B, w = x0.shape
        L = kernels.shape[0]
        outputs = torch.empty((L, B, w), dtype=torch.float32, device=x0.device)
        hiddens = torch.empty((L, B), dtype=torch.float32, device=x0.device)
        check(lib.hctr_cross_v1_fwd(B, w, L, ptr(x0), ptr(kernels), ptr(biases), ptr(outputs),
                                    ptr(hiddens), stream_ptr()))
        ctx.save_for_backward(x0, kernels, outputs, hiddens)
        return outputs[L - 1]

    @staticmethod
    def backward(ctx, grad):
        x0, kernels, outputs, hiddens = ctx.saved_tensors
        grad = grad.contiguous()
        B, w = x0.shape
        L = kernels.shape[0]
        in_grad = torch.empty_like(x0)
        kg = torch.empty_like(kernels)
        bg = torch.empty_like(kernels)
        ws = torch.empty(lib.hctr_cross_v1_bwd_workspace_bytes(B, w, L) // 4, dtype=torch.float32,
                         device=x0.device)
        check(lib.hctr_cross_v1_bwd(B, w, L, ptr(x0), ptr(kernels), ptr(outputs), ptr(hiddens),
                                    ptr(grad), ptr(in_grad), ptr(kg), ptr(bg), ptr(ws),
                                    stream_ptr()))
        return in_grad, kg, bg


def _mm_f32(a, b):
    """a @ b of 16-bit operands with an fp32 result (weight gradients: a sum over the whole batch
    does not fit fp16's range at loss scale 1024)"""
    if a.dtype in (torch.float16, torch.bfloat16):
        try:
            return torch.mm(a, b, out_dtype=torch.float32)
        except (TypeError, NotImplementedError, RuntimeError):  # (no out_dtype on this backend)
            return torch.mm(a.float(), b.float())
    return torch.mm(a, b)


def _own_gemm_ok(x, n, k):
    """the shapes hctr_gemm_nt16 takes (cross_gemm.hip): 16-bit operands on the device, N a
    multiple of 128, K of 64; HCTR_CROSS_GEMM=0 keeps the library GEMMs (measurements)"""
    return (x.is_cuda and x.dtype in _DT16 and n % 128 == 0 and k % 64 == 0 and
            os.environ.get("HCTR_CROSS_GEMM", "1") != "0")


def gemm_nt16(a, bt, epilogue=0, bias=None, x0=None, xl=None):
    """a [M, K] @ bt [N, K]^T on the library's own matrix-core kernel (hctr_gemm_nt16); epilogue 1
    returns (x0 * (a @ bt^T + bias) + xl, a @ bt^T + bias), epilogue 2 returns a @ bt^T + xl"""
    M, K = a.shape
    N = bt.shape[0]
    # rows may be strided (a leading dimension), elements of a row may not; the epilogue operands
    # are read as [M, N] with leading dimension N
    if a.stride(1) != 1:
        a = a.contiguous()
    if bt.stride(1) != 1:
        bt = bt.contiguous()
    assert bt.shape[1] == K and a.dtype == bt.dtype
    bias, x0, xl = (t if t is None or t.is_contiguous() else t.contiguous() for t in (bias, x0, xl))
    assert x0 is None or x0.shape == (M, N)
    assert xl is None or xl.shape == (M, N)
    c = torch.empty((M, N), dtype=a.dtype, device=a.device)
    h = torch.empty_like(c) if epilogue == 1 else None
    check(lib.hctr_gemm_nt16(M, N, K, ptr(a), a.stride(0), ptr(bt), bt.stride(0), ptr(c), N, epilogue,
                             ptr(bias) if bias is not None else None,
                             ptr(x0) if x0 is not None else None,
                             ptr(xl) if xl is not None else None,
                             ptr(h) if h is not None else None, _DT16[a.dtype], stream_ptr()))
    return (c, h) if epilogue == 1 else c


class _CrossV2Fn(torch.autograd.Function):
    """DCN-v2 cross layers, x_{l+1} = x0 * (x_l U_l V_l + b_l) + x_l, in the activations' own type
    T (fp16 / bf16 under use_mixed_precision: the GEMMs run on the matrix cores; fp32 otherwise)
    with fp32 master weights -- MultiCrossForwardFunctorv2 / MultiCrossBackwardFunctorv2
    (R/HugeCTR/src/layers/multi_cross_layer.cu:582-700,732-812): per layer two GEMMs forward (the
    bias rides in the second one's epilogue) and four backward (S1 = S0 V^T, dV = (XU)^T S0,
    dU = X^T S1, dY_{l-1} = S1 U^T + dY_l with the residual in the GEMM's epilogue); S0 = dY .* x0
    and the x0 gradient dY .* H accumulate in T as the reference's fused_mul_fma3 does.  Weight
    and bias gradients leave as fp32."""

    @staticmethod
    def forward(ctx, x0, U, V, b):
        T = x0.dtype
        L = U.shape[0]
        xs, ps, hs = [x0], [], []
        xl = x0
        w, p = U.shape[1], U.shape[2]
        # (hctr_convert_transpose16 reads U / V as fp32: 16-bit parameters -- module.half() -- take
        #  the library GEMMs; the epilogue operands are read with leading dimension N)
        own = (_own_gemm_ok(x0, p, w) and _own_gemm_ok(x0, w, p) and
               U.dtype == torch.float32 and V.dtype == torch.float32 and x0.is_contiguous())
        bt = b.to(T)
        if own:
            # the 16-bit copies of the master weights as they lie (backward) and transposed (the
            # forward's operands are read K-contiguous), one pass per tensor
            Ut, UtT = torch.empty_like(U, dtype=T), torch.empty((L, p, w), dtype=T, device=U.device)
            Vt, VtT = torch.empty_like(V, dtype=T), torch.empty((L, w, p), dtype=T, device=V.device)
            Uc, Vc = U.contiguous(), V.contiguous()
            check(lib.hctr_convert_transpose16(L, w, p, ptr(Uc), ptr(Ut), ptr(UtT), _DT16[T],
                                               stream_ptr()))
            check(lib.hctr_convert_transpose16(L, p, w, ptr(Vc), ptr(Vt), ptr(VtT), _DT16[T],
                                               stream_ptr()))
        else:
            Ut, Vt = U.to(T), V.to(T)
        for l in range(L):
            if own:
                # two launches per layer: P = X_l U; then bias + X_0 .* H + X_l in the second
                # GEMM's epilogue (the reference's fused epilogues, multi_cross_layer.cu:640-700)
                pl = gemm_nt16(xl, UtT[l])
                xl, h = gemm_nt16(pl, VtT[l], 1, bt[l], x0, xl)
            else:
                pl = xl @ Ut[l]
                h = torch.addmm(bt[l], pl, Vt[l])
                xl = torch.addcmul(xl, x0, h)
            ps.append(pl)
            hs.append(h)
            xs.append(xl)
        ctx.save_for_backward(Ut, Vt, *xs[:-1], *ps, *hs)
        ctx.L = L
        return xl

    @staticmethod
    def backward(ctx, grad):
        L = ctx.L
        sv = ctx.saved_tensors
        Ut, Vt = sv[0], sv[1]
        xs, ps, hs = sv[2:2 + L], sv[2 + L:2 + 2 * L], sv[2 + 2 * L:2 + 3 * L]
        x0 = xs[0]
        dy = grad.contiguous()
        acc = torch.empty_like(x0)
        wdt = torch.float32 if x0.dtype in (torch.float16, torch.bfloat16) else x0.dtype
        dU = torch.empty(Ut.shape, dtype=wdt, device=x0.device)
        dV = torch.empty(Vt.shape, dtype=wdt, device=x0.device)
        db = torch.empty((L, x0.shape[1]), dtype=wdt, device=x0.device)
        B, w = x0.shape
        fused = x0.is_cuda and x0.dtype in _DT16 and w % 8 == 0
        if fused:
            # S0 = dY .* X0, dX += dY .* H and db = colsum(S0) in one pass over dY (HIP), as the
            # reference's fused_mul_fma3 + the dV GEMM's bias-gradient epilogue
            s0 = torch.empty_like(x0)
            ws = torch.empty(lib.hctr_cross_v2_bwd_step_workspace_bytes(B, w) // 4,
                             dtype=torch.float32, device=x0.device)
        else:
            acc.zero_()
        for l in range(L - 1, -1, -1):
            if fused:
                check(lib.hctr_cross_v2_bwd_step(B, w, ptr(dy), ptr(x0), ptr(hs[l]), ptr(acc), ptr(s0),
                                                 ptr(db[l]), ptr(ws), 1 if l == L - 1 else 0,
                                                 _DT16[x0.dtype], stream_ptr()))
            else:
                s0 = dy * x0
                acc.addcmul_(dy, hs[l])
            own1 = _own_gemm_ok(s0, Vt.shape[1], Vt.shape[2])
            s1 = gemm_nt16(s0, Vt[l]) if own1 else s0 @ Vt[l].t()
            # weight gradients reduce over K = batch into a small tile set (512 x 3456: 27 tiles of
            # 256 x 256 on 256 CUs): split-K through batched GEMMs, as the MLP's are (dense.py)
            if x0.is_cuda and x0.dtype in (torch.float16, torch.bfloat16):
                split_k_wgrad(ps[l], s0, out=dV[l])
                split_k_wgrad(xs[l], s1, out=dU[l])
            else:
                dV[l] = _mm_f32(ps[l].t(), s0)
                dU[l] = _mm_f32(xs[l].t(), s1)
            if not fused:
                db[l] = s0.sum(0, dtype=wdt)
            if _own_gemm_ok(s1, Ut.shape[1], Ut.shape[2]):
                dy = gemm_nt16(s1, Ut[l], 2, xl=dy)  # (the residual in the GEMM's epilogue)
            else:
                dy = torch.addmm(dy, s1, Ut[l].t())
        return acc + dy, dU, dV, db


class MultiCrossLayer(torch.nn.Module):
    """DCN cross layers.  projection_dim == 0: v1 (x_{l+1} = x0 * (x_l . w_l) + b_l + x_l), all
    layers fused in one HIP launch.  projection_dim > 0: v2 (x_{l+1} = x0 * (x_l U_l V_l + b_l)
    + x_l): the GEMMs go to hipBLASLt through torch in the activations' type (16-bit inputs: on the
    matrix cores), bias and residual ride in GEMM epilogues (_CrossV2Fn)."""

    def __init__(self, width: int, num_layers: int, projection_dim: int = 0):
        super().__init__()
        self.width, self.num_layers, self.projection_dim = width, num_layers, projection_dim
        if projection_dim == 0:
            # XavierUniform-like default init (weights [1,w] per layer, bias zeros)
            bound = (6.0 / (width + 1)) ** 0.5
            self.kernels = torch.nn.Parameter(torch.empty(num_layers, width).uniform_(-bound, bound))
            self.biases = torch.nn.Parameter(torch.zeros(num_layers, width))
        else:
            bu = (6.0 / (width + projection_dim)) ** 0.5
            self.U = torch.nn.Parameter(torch.empty(num_layers, width, projection_dim).uniform_(-bu, bu))
            self.V = torch.nn.Parameter(torch.empty(num_layers, projection_dim, width).uniform_(-bu, bu))
            self.biases = torch.nn.Parameter(torch.zeros(num_layers, width))

    def forward(self, x0):
        if self.projection_dim == 0:
            return _CrossV1Fn.apply(x0, self.kernels, self.biases)
        return _CrossV2Fn.apply(x0.contiguous(), self.U, self.V, self.biases)
