"""GPU parity: InteractionLayer / MultiCrossLayer kernels vs the CPU restatement of the
reference's own layer-test references.  Tolerance: rel 1e-4 (the reference's eps is 1e-3)."""
import numpy as np
import pytest

from util import assert_close

pytestmark = pytest.mark.gpu


@pytest.mark.parametrize("B,n_emb,W", [
    (512, 26, 128),   # interaction_layer_test.cpp:290-340 shapes / DLRM
    (130, 26, 128),   # ragged tail (B not a multiple of waves)
    (64, 26, 64),
    (33, 13, 32),
    (7, 5, 16),
    (9, 3, 10),       # generic path (W not MFMA friendly)
    (5, 40, 8),       # n_ins > 32 -> generic path
])
def test_interaction_fwd_bwd(oracle, B, n_emb, W):
    import torch
    import hugectr_amd as ha
    rng = np.random.default_rng(B)
    mlp = rng.standard_normal((B, W)).astype(np.float32)
    emb = rng.standard_normal((B, n_emb, W)).astype(np.float32)
    mt = torch.from_numpy(mlp).cuda().requires_grad_(True)
    et = torch.from_numpy(emb).cuda().requires_grad_(True)
    out = ha.interaction(mt, et)
    want = oracle.interaction_fwd(mlp, emb)
    assert out.shape == want.shape
    assert_close(out.detach().cpu().numpy(), want, 2e-4, 1e-3, "interaction fwd")
    assert float(out.detach()[:, -1].abs().max()) == 0.0  # zero pad column
    g = rng.standard_normal(want.shape).astype(np.float32)
    out.backward(torch.from_numpy(g).cuda())
    mg, eg = oracle.interaction_bwd(mlp, emb, g)
    assert_close(mt.grad.cpu().numpy(), mg, 2e-4, 1e-3, "interaction mlp grad")
    assert_close(et.grad.cpu().numpy(), eg, 2e-4, 1e-3, "interaction emb grad")


def test_interaction_asymmetric_catches_transpose(oracle):
    """distinct magnitude per row so a row<->col swap or a wrong pair order cannot pass"""
    import torch
    import hugectr_amd as ha
    B, n_emb, W = 4, 26, 128
    mlp = np.full((B, W), 1.0, np.float32)
    emb = np.zeros((B, n_emb, W), np.float32)
    for i in range(n_emb):
        emb[:, i, i] = 10.0 ** (i % 5) * (i + 2)
        emb[:, i, 127 - i] = 1.0
    out = ha.interaction(torch.from_numpy(mlp).cuda(), torch.from_numpy(emb).cuda())
    assert_close(out.cpu().numpy(), oracle.interaction_fwd(mlp, emb), 1e-5, 1e-5, "asym")


@pytest.mark.parametrize("B,w,L", [(1024, 429, 6), (37, 64, 1), (200, 1000, 3), (16, 13, 2)])
def test_cross_v1_fwd_bwd(oracle, B, w, L):
    import torch
    import hugectr_amd as ha
    rng = np.random.default_rng(w)
    x0 = np.clip(rng.standard_normal((B, w)), -0.09 * 10, 0.09 * 10).astype(np.float32) * 0.1
    layer = ha.MultiCrossLayer(w, L, 0).cuda()
    with torch.no_grad():
        layer.kernels.copy_(torch.from_numpy(np.clip(rng.standard_normal((L, w)), -1, 1).astype(np.float32)))
        layer.biases.copy_(torch.from_numpy((rng.standard_normal((L, w)) * 0.01).astype(np.float32)))
    k = layer.kernels.detach().cpu().numpy()
    b = layer.biases.detach().cpu().numpy()
    xt = torch.from_numpy(x0).cuda().requires_grad_(True)
    out = layer(xt)
    outs, hid = oracle.cross_v1_fwd(x0, k, b)
    assert_close(out.detach().cpu().numpy(), outs[-1], 1e-4, 1e-4, "cross fwd")
    og = np.full((B, w), 0.1, np.float32) + rng.standard_normal((B, w)).astype(np.float32) * 0.01
    out.backward(torch.from_numpy(og).cuda())
    ig, kg, bg = oracle.cross_v1_bwd(x0, k, outs, hid, og)
    assert_close(xt.grad.cpu().numpy(), ig, 1e-4, 1e-4, "cross dx")
    assert_close(layer.kernels.grad.cpu().numpy(), kg, 2e-4, 1e-4, "cross dw")
    assert_close(layer.biases.grad.cpu().numpy(), bg, 2e-4, 1e-4, "cross db")


@pytest.mark.parametrize("dtype_name", ["float32", "float16", "bfloat16"])
def test_cross_v2_matches_oracle(oracle, dtype_name):
    """MultiCross v2 forward AND backward in the activations' type (fp32; fp16 / bf16 = the
    reference's mixed-precision MultiCrossLayer<__half>: GEMMs in the 16-bit type, fp32 master
    weights and weight gradients) against the fp32 oracle (CPU restatement of
    multi_cross_layer_test.cpp's reference).

    The bound is magnitude-aware: every [B, w] intermediate (P = X_l U, H = P V + b, X_0 .* H,
    X_{l+1}) is rounded to the activation type, so an element of the result carries a few roundings
    of numbers as large as the LARGEST intermediate that fed it -- not of its own (possibly tiny)
    value.  err <= k * eps_T * (|want| + scale), scale = the largest magnitude among the oracle's
    own intermediates of that tensor's chain.  (The reference's own test accepts rel 0.1 - 0.4 for
    half, R/test/utest/core23_layer_test/multi_cross_layer_test.cpp:152-432.)"""
    import torch
    import hugectr_amd as ha
    dt = getattr(torch, dtype_name)
    eps = {"float32": 2.0 ** -24, "float16": 2.0 ** -11, "bfloat16": 2.0 ** -8}[dtype_name]
    rng = np.random.default_rng(11)
    B, w, p, L = 64, 96, 16, 3
    x0 = (rng.standard_normal((B, w)) * 0.5).astype(np.float32)
    layer = ha.MultiCrossLayer(w, L, p).cuda()
    with torch.no_grad():
        layer.biases.normal_(0, 0.1)
    U = layer.U.detach().cpu().numpy(); V = layer.V.detach().cpu().numpy()
    b = layer.biases.detach().cpu().numpy()
    xt = torch.from_numpy(x0).cuda().to(dt).requires_grad_(True)
    x0r = xt.detach().float().cpu().numpy()  # (the oracle sees the rounded input)
    out = layer(xt)
    assert out.dtype == dt
    outs, hid, xus = oracle.cross_v2_fwd(x0r, U, V, b)

    def amax(*xs):
        return max(float(np.abs(np.asarray(x)).max()) for x in xs)

    def near(got, want, scale, k, what):
        err = np.abs(np.asarray(got, np.float64) - want)
        tol = k * eps * (np.abs(want) + scale)
        bad = err > tol
        assert not bad.any(), (f"{what}: {bad.sum()} / {bad.size} beyond {k} eps x (|want| + {scale:.3g}); "
                               f"max err {err.max():.3e}")

    fscale = amax(*outs, *hid, *xus, *[x0r * h for h in hid])
    # k: measured worst case over 40 parameter draws is 1.1 (fwd) / 1.9 (bwd) eps for the 16-bit types
    # and 3.1 / 5.4 eps for fp32 (summation order of the w = 96 dot products); 6x headroom
    kf, kb = (32, 64) if dtype_name == "float32" else (8, 16)
    near(out.detach().float().cpu().numpy(), outs[-1], fscale, kf, "cross v2 fwd")
    og = (rng.standard_normal((B, w)) * 0.5).astype(np.float32)
    ogt = torch.from_numpy(og).cuda().to(dt)
    out.backward(ogt)
    ogr = ogt.float().cpu().numpy()
    ig, dU, dV, db = oracle.cross_v2_bwd(x0r, U, V, outs, hid, xus, ogr)
    assert layer.U.grad.dtype == torch.float32 and layer.biases.grad.dtype == torch.float32
    # backward: the gradient chain's intermediates scale with |dy| x the forward's magnitudes, and
    # dU / dV / db are sums over the B rows of rounded [B, w] / [B, p] products
    near(xt.grad.float().cpu().numpy(), ig, amax(ig, ogr) * max(1.0, fscale), kb, "cross v2 dx")
    near(layer.U.grad.cpu().numpy(), dU, amax(dU), kb, "cross v2 dU")
    near(layer.V.grad.cpu().numpy(), dV, amax(dV), kb, "cross v2 dV")
    near(layer.biases.grad.cpu().numpy(), db, amax(db), kb, "cross v2 db")


@pytest.mark.parametrize("dtype_name", ["float16", "bfloat16"])
@pytest.mark.parametrize("M,N,K,bm", [(200, 128, 64, 0), (1000, 256, 192, 128), (130, 384, 128, 64),
                                      (64, 128, 512, 0), (700, 512, 192, 256), (256, 256, 64, 256)])
def test_own_gemm_nt16_and_its_epilogues(monkeypatch, dtype_name, M, N, K, bm):
    """hctr_gemm_nt16 (cross_gemm.hip: MFMA 32x32x16, LDS-DMA staging, swizzled tile image) against
    fp64 products of the same 16-bit operands: plain, the forward's fused epilogue
    (H = acc + b; C = X_l + X_0 * H, each rounded once) and the backward's residual; the tile
    heights 64 / 128 and the 256 x 256 tiles of the plain product (8 wavefronts, two epilogue passes); M that is no multiple of the tile (rows past the end are read clamped, never stored);
    the operand B asymmetric (a swapped fragment layout cannot pass)."""
    import torch
    from hugectr_amd.layers import gemm_nt16
    dt = getattr(torch, dtype_name)
    if bm:
        monkeypatch.setenv("HCTR_GEMM_BM", str(bm))
    g = torch.Generator(device="cuda")
    g.manual_seed(M + N + K)
    a = (torch.randn((M, K), device="cuda", generator=g) * 0.5).to(dt)
    bt = (torch.randn((N, K), device="cuda", generator=g) * 0.5).to(dt)
    bt[:, 0] += torch.arange(N, device="cuda").to(dt) * 0.01  # (asymmetric in n)
    bias = torch.randn((N,), device="cuda", generator=g).to(dt)
    x0 = torch.randn((M, N), device="cuda", generator=g).to(dt)
    xl = torch.randn((M, N), device="cuda", generator=g).to(dt)
    ref = a.double() @ bt.double().t()
    eps = 2.0 ** -8 if dt == torch.bfloat16 else 2.0 ** -11
    scale = float(ref.abs().max())

    def close(got, want, what, ulps=1.0):
        err = float((got.double() - want).abs().max())
        bound = ulps * eps * max(float(want.abs().max()), 1e-3) + 1e-5 * scale
        assert err <= bound, (what, err, bound)

    c = gemm_nt16(a, bt)
    torch.cuda.synchronize()
    close(c, ref, "plain")
    out, h = gemm_nt16(a, bt, 1, bias, x0, xl)
    torch.cuda.synchronize()
    hw = ref + bias.double()
    close(h, hw, "cross: h")
    # C is computed from the ROUNDED h, as the unfused passes do
    close(out, xl.double() + x0.double() * h.double(), "cross: out")
    r = gemm_nt16(a, bt, 2, xl=xl)
    torch.cuda.synchronize()
    close(r, ref + xl.double(), "residual")
    # untouched neighbours: a C with a wider leading dimension keeps its other columns
    if N == 128:
        from hugectr_amd import _lib
        from hugectr_amd._lib import check, lib, ptr
        wide = torch.full((M, 2 * N), 7.0, device="cuda").to(dt)
        check(lib.hctr_gemm_nt16(M, N, K, ptr(a), K, ptr(bt), K, ptr(wide), 2 * N, 0, None, None, None,
                                 None, _lib.F16 if dt == torch.float16 else _lib.BF16, None))
        torch.cuda.synchronize()
        assert torch.equal(wide[:, :N], c) and bool((wide[:, N:] == 7.0).all())


@pytest.mark.parametrize("dtype_name", ["float16", "bfloat16"])
@pytest.mark.parametrize("L,R,C", [(3, 3456, 512), (2, 70, 33), (1, 1, 5)])
def test_convert_transpose16(dtype_name, L, R, C):
    """hctr_convert_transpose16: the 16-bit copy of fp32 weights and its transpose, the same bits as
    torch's conversion"""
    import torch
    from hugectr_amd import _lib
    from hugectr_amd._lib import check, lib, ptr
    dt = getattr(torch, dtype_name)
    src = torch.randn((L, R, C), device="cuda")
    a = torch.zeros((L, R, C), device="cuda", dtype=dt)
    b = torch.zeros((L, C, R), device="cuda", dtype=dt)
    code = _lib.F16 if dt == torch.float16 else _lib.BF16
    check(lib.hctr_convert_transpose16(L, R, C, ptr(src), ptr(a), ptr(b), code, None))
    torch.cuda.synchronize()
    assert torch.equal(a, src.to(dt)) and torch.equal(b, src.to(dt).transpose(1, 2))
    b.zero_()
    check(lib.hctr_convert_transpose16(L, R, C, ptr(src), None, ptr(b), code, None))
    torch.cuda.synchronize()
    assert torch.equal(b, src.to(dt).transpose(1, 2))


@pytest.mark.parametrize("dtype_name,rt,at", [("float16", 2e-2, 2e-2), ("bfloat16", 6e-2, 6e-2)])
@pytest.mark.parametrize("B,w,p,L", [(200, 256, 128, 2), (1024, 3456, 512, 3)])
def test_cross_v2_on_the_own_gemm_matches_oracle(oracle, monkeypatch, dtype_name, rt, at, B, w, p, L):
    """MultiCross v2 with its four activation GEMMs per layer on hctr_gemm_nt16 (forward: P = X U,
    then bias + X_0 .* H + X_l in the second GEMM's epilogue; backward: S1 = S0 V^T and the residual
    GEMM) against the fp32 oracle at a small shape and at `extra.cross`'s X1 shape (B = 8192,
    w = 3456, p = 512, 3 layers), and against the library-GEMM form of the same layer
    (HCTR_CROSS_GEMM=0)."""
    import os
    import torch
    import hugectr_amd as ha
    if B > 1000 and os.environ.get("HCTR_EMU") == "1":
        pytest.skip("22 GFLOP: not under the host interpreter")
    dt = getattr(torch, dtype_name)
    rng = np.random.default_rng(5)
    x0 = (rng.standard_normal((B, w)) * 0.5).astype(np.float32)
    layer = ha.MultiCrossLayer(w, L, p).cuda()
    with torch.no_grad():
        layer.biases.normal_(0, 0.1)
    U = layer.U.detach().cpu().numpy(); V = layer.V.detach().cpu().numpy()
    b = layer.biases.detach().cpu().numpy()
    og = (rng.standard_normal((B, w)) * 0.5).astype(np.float32)
    res = {}
    for own in ("1", "0"):
        monkeypatch.setenv("HCTR_CROSS_GEMM", own)
        layer.zero_grad()
        xt = torch.from_numpy(x0).cuda().to(dt).requires_grad_(True)
        out = layer(xt)
        out.backward(torch.from_numpy(og).cuda().to(dt))
        torch.cuda.synchronize()
        res[own] = (out.detach().float().cpu().numpy(), xt.grad.float().cpu().numpy(),
                    layer.U.grad.cpu().numpy().copy(), layer.V.grad.cpu().numpy().copy(),
                    layer.biases.grad.cpu().numpy().copy())
    x0r = torch.from_numpy(x0).to(dt).float().numpy()
    ogr = torch.from_numpy(og).to(dt).float().numpy()
    outs, hid, xus = oracle.cross_v2_fwd(x0r, U, V, b)
    ig, dU, dV, db = oracle.cross_v2_bwd(x0r, U, V, outs, hid, xus, ogr)
    # tolerances: relative to the tensor's own scale (sums over w = 3456 terms of 16-bit products)
    def near(got, want, what, k=1.0):
        err = np.abs(got.astype(np.float64) - want).max()
        assert err <= k * (rt * np.abs(want).max() + at * 1e-2), (what, err, np.abs(want).max())
    for own in ("1", "0"):
        o, dx, gU, gV, gb = res[own]
        near(o, outs[-1], f"fwd own={own}")
        near(dx, ig, f"dx own={own}", 2)
        near(gU, dU, f"dU own={own}", 2)
        near(gV, dV, f"dV own={own}", 2)
        near(gb, db, f"db own={own}", 2)
    near(res["1"][0], res["0"][0].astype(np.float64), "own vs library fwd")


def test_cross_v2_own_gemm_at_the_x1_shape_equals_the_library_form(monkeypatch):
    """`extra.cross`'s X1 shape (B = 8192, w = 3456, p = 512, 3 layers, fp16) forward + backward on
    hctr_gemm_nt16 against the same layer on the library GEMMs (both sides on the device: the
    oracle at this size takes minutes of host time): outputs and gradients agree to the 16-bit
    rounding of the [B, w] intermediates."""
    import os
    import torch
    import hugectr_amd as ha
    if os.environ.get("HCTR_EMU") == "1":
        pytest.skip("174 GFLOP: not under the host interpreter")
    B, w, p, L = 8192, 3456, 512, 3
    torch.manual_seed(3)
    layer = ha.MultiCrossLayer(w, L, p).cuda()
    with torch.no_grad():
        layer.biases.normal_(0, 0.1)
    x = (torch.randn(B, w, device="cuda") * 0.5).half()
    g = (torch.randn(B, w, device="cuda") * 0.5).half()
    res = {}
    for own in ("1", "0"):
        monkeypatch.setenv("HCTR_CROSS_GEMM", own)
        layer.zero_grad()
        xt = x.clone().requires_grad_(True)
        out = layer(xt)
        out.backward(g)
        torch.cuda.synchronize()
        res[own] = [t.detach().double() for t in (out, xt.grad, layer.U.grad, layer.V.grad,
                                                  layer.biases.grad)]
    for a, b, what in zip(res["1"], res["0"], ("out", "dx", "dU", "dV", "db")):
        err = float((a - b).abs().max())
        assert err <= 4e-3 * float(b.abs().max()) + 1e-4, (what, err, float(b.abs().max()))


@pytest.mark.parametrize("dtype_name", ["bfloat16", "float16"])
@pytest.mark.parametrize("B,w", [(300, 3456), (129, 24), (7, 2056)])
def test_cross_v2_backward_step_equals_the_three_torch_passes(dtype_name, B, w):
    """hctr_cross_v2_bwd_step (S0 = dY .* X0, dX (+)= dY .* H, db = column sums of S0, one pass over
    dY) against the three passes it replaces: the same bits for S0 and dX (products of 16-bit values
    are exact in fp32, each element is rounded once), db within fp32 summation error"""
    import torch
    from hugectr_amd import _lib
    from hugectr_amd._lib import check, lib, ptr
    dt = getattr(torch, dtype_name)
    g = torch.Generator(device="cuda")
    g.manual_seed(B + w)
    dy, x0, h, acc0 = (torch.randn((B, w), device="cuda", generator=g).to(dt) for _ in range(4))
    ws = torch.empty(lib.hctr_cross_v2_bwd_step_workspace_bytes(B, w) // 4, dtype=torch.float32, device="cuda")
    code = _lib.F16 if dt == torch.float16 else _lib.BF16
    for first in (1, 0):
        acc = torch.full_like(acc0, float("nan")) if first else acc0.clone()
        s0 = torch.full_like(dy, float("nan"))
        db = torch.full((w,), float("nan"), device="cuda")
        check(lib.hctr_cross_v2_bwd_step(B, w, ptr(dy), ptr(x0), ptr(h), ptr(acc), ptr(s0), ptr(db), ptr(ws),
                                         first, code, None))
        torch.cuda.synchronize()
        want_s0 = dy * x0
        want_acc = dy * h if first else torch.addcmul(acc0, dy, h)
        assert torch.equal(s0, want_s0), (first, "s0")
        assert torch.equal(acc, want_acc), (first, "acc")
        ref = want_s0.double().sum(0)
        assert float((db.double() - ref).abs().max()) <= 1e-5 * float(want_s0.double().abs().sum(0).max()) + 1e-6


@pytest.mark.parametrize("dtype_name", ["bfloat16", "float16"])
@pytest.mark.parametrize("B,n_emb,W", [(512, 26, 128), (70, 26, 64), (33, 13, 32)])
def test_interaction_16bit(oracle, dtype_name, B, n_emb, W):
    """mixed-precision mode (reference: InteractionLayer<__half>, eps 1.0 in its own test,
    interaction_layer_test.cpp:44-47); oracle = fp32 math on the 16-bit-rounded inputs"""
    import torch
    import hugectr_amd as ha
    dt = getattr(torch, dtype_name)
    rng = np.random.default_rng(B + W)
    mt = torch.from_numpy(rng.standard_normal((B, W)).astype(np.float32)).cuda().to(dt)
    et = torch.from_numpy(rng.standard_normal((B, n_emb, W)).astype(np.float32)).cuda().to(dt)
    mlp, emb = mt.float().cpu().numpy(), et.float().cpu().numpy()
    mt.requires_grad_(True)
    et.requires_grad_(True)
    out = ha.interaction(mt, et)
    assert out.dtype == dt
    want = oracle.interaction_fwd(mlp, emb)
    eps = 2.0 ** -7 if dtype_name == "bfloat16" else 2.0 ** -10
    assert_close(out.detach().float().cpu().numpy(), want, 2 * eps, 2 * eps * 12, "fwd16")
    gt = torch.from_numpy(rng.standard_normal(want.shape).astype(np.float32)).cuda().to(dt)
    out.backward(gt)
    mg, eg = oracle.interaction_bwd(mlp, emb, gt.float().cpu().numpy())
    assert_close(mt.grad.float().cpu().numpy(), mg, 4 * eps, 4 * eps * 8, "mlp grad16")
    assert_close(et.grad.float().cpu().numpy(), eg, 4 * eps, 4 * eps * 8, "emb grad16")


@pytest.mark.parametrize("dtype_name", ["bfloat16", "float16"])
@pytest.mark.parametrize("rows,n", [(1, 8), (127, 64), (4096, 512), (1000, 1024), (300, 4096), (513, 40)])
def test_relu_bwd_bias_fused(dtype_name, rows, n):
    """fused dz = dy*(y>0), db = colsum(dz) vs torch threshold_backward + sum (dz bit-exact)"""
    import torch
    from hugectr_amd._lib import check, lib, ptr, stream_ptr
    dt = getattr(torch, dtype_name)
    g = torch.Generator(device="cuda").manual_seed(rows * 31 + n)
    dy = torch.randn(rows, n, device="cuda", generator=g).to(dt)
    y = torch.relu(torch.randn(rows, n, device="cuda", generator=g)).to(dt)
    dz = torch.empty_like(dy)
    db = torch.empty(n, dtype=torch.float32, device="cuda")
    ws = torch.empty(lib.hctr_relu_bwd_bias_workspace_bytes(rows, n) // 4, dtype=torch.float32,
                     device="cuda")
    check(lib.hctr_relu_bwd_bias(rows, n, ptr(dy), ptr(y), ptr(dz), ptr(db), ptr(ws),
                                 1 if dt == torch.float16 else 2, stream_ptr()))
    ref = torch.ops.aten.threshold_backward(dy, y, 0)
    assert torch.equal(dz, ref)
    ref_db = ref.double().sum(0)
    assert torch.allclose(db.double(), ref_db, rtol=1e-5, atol=1e-4)


def test_fused_mlp_matches_torch_fp32():
    """FusedMLP (bf16 compute, fused relu-bwd/bias-grad, split-K wgrad) vs fp32 torch MLP"""
    import torch
    from hugectr_amd.dense import FusedMLP
    torch.manual_seed(5)
    mlp = FusedMLP([64, 128, 64, 1], last_relu=False).cuda()
    mlp.refresh_shadow()
    x = torch.randn(2048, 64, device="cuda")
    out = mlp(x).float()
    out.sum().backward()
    h = x
    ws = [w.detach().clone().requires_grad_() for w in mlp.weights]
    bs = [b.detach().clone().requires_grad_() for b in mlp.biases]
    for i, (w, b) in enumerate(zip(ws, bs)):
        h = torch.nn.functional.linear(h, w, b)
        if mlp.relu[i]:
            h = torch.relu(h)
    h.sum().backward()
    assert torch.allclose(out, h, rtol=5e-2, atol=5e-2)
    for p, r in zip(list(mlp.weights) + list(mlp.biases), ws + bs):
        rel = (p.grad - r.grad).norm() / (r.grad.norm() + 1e-12)
        assert rel < 3e-2, rel


@pytest.mark.parametrize("dtype_name", ["float32", "bfloat16", "float16"])
@pytest.mark.parametrize("batch", [1, 255, 65536, 300001])
def test_bce_loss_fused(dtype_name, batch):
    """fused BCE loss + logit gradient vs torch BCEWithLogitsLoss (R/HugeCTR/src/loss.cu:231-262)"""
    import torch
    from hugectr_amd.dense import bce_with_logits
    dt = getattr(torch, dtype_name)
    g = torch.Generator(device="cuda").manual_seed(batch)
    x = (torch.randn(batch, 1, device="cuda", generator=g) * 4).to(dt)
    y = (torch.rand(batch, 1, device="cuda", generator=g) < 0.3).float()
    loss, dx = bce_with_logits(x, y, 0.5 / batch)
    xr = x.double().requires_grad_()
    ref = torch.nn.functional.binary_cross_entropy_with_logits(xr, y.double())
    (ref * 0.5).backward()
    refv = float(ref.detach())
    assert abs(float(loss) - refv) <= 1e-5 * max(1.0, abs(refv))
    tol = 1e-6 if dt == torch.float32 else 1e-2
    assert torch.allclose(dx.double(), xr.grad, rtol=tol, atol=tol / batch)
    # deterministic: same bits on a second run
    loss2, dx2 = bce_with_logits(x, y, 0.5 / batch)
    assert torch.equal(loss, loss2) and torch.equal(dx, dx2)


def test_sum_groups_fixed_order():
    import torch
    from hugectr_amd.dense import split_k_wgrad
    g = torch.Generator(device="cuda").manual_seed(3)
    dy = torch.randn(4096, 64, device="cuda", generator=g).bfloat16()
    x = torch.randn(4096, 40, device="cuda", generator=g).bfloat16()
    dw = split_k_wgrad(dy, x, 16)
    ref = dy.double().t() @ x.double()
    assert dw.dtype == torch.float32
    assert (dw.double() - ref).norm() / ref.norm() < 1e-2
    assert torch.equal(dw, split_k_wgrad(dy, x, 16))


@pytest.mark.parametrize("dtype_name", ["bfloat16", "float16"])
@pytest.mark.parametrize("B,n_emb,W", [(300, 26, 128), (65, 5, 64), (33, 10, 32), (70, 21, 128)])
def test_interaction_indexed_equals_dense(dtype_name, B, n_emb, W):
    """interaction over (rows, row_of) == interaction over the expanded tensor, bit for bit,
    forward and both gradients"""
    import torch
    import hugectr_amd as ha
    dt = getattr(torch, dtype_name)
    g = torch.Generator(device="cuda").manual_seed(B + n_emb)
    R = 97
    rows = torch.randn(R, W, device="cuda", generator=g).to(dt)
    row_of = torch.randint(0, R, (B, n_emb), device="cuda", generator=g, dtype=torch.int32)
    mlp = torch.randn(B, W, device="cuda", generator=g).to(dt)
    emb = rows[row_of.long()].contiguous()
    m1, m2 = mlp.clone().requires_grad_(), mlp.clone().requires_grad_()
    e1 = emb.clone().requires_grad_()
    got = {}
    out_i = ha.interaction_indexed(m2, rows, row_of, on_emb_grad=lambda d: got.setdefault("dE", d))
    out_d = ha.interaction(m1, e1)
    assert torch.equal(out_i, out_d)
    top = torch.randn(out_d.shape, device="cuda", generator=g).to(dt)
    out_d.backward(top)
    out_i.backward(top)
    assert torch.equal(m1.grad, m2.grad)
    assert torch.equal(e1.grad, got["dE"])


@pytest.mark.parametrize("dtype_name", ["bfloat16", "float16"])
@pytest.mark.parametrize("B,n_emb,W,world", [(96, 26, 128, 4), (64, 26, 32, 2), (40, 10, 64, 8)])
def test_interaction_through_the_reorder_map_needs_no_reorder_pass(dtype_name, B, n_emb, W, world):
    """N > 1 rows payload: the interaction reads the all-to-all receive buffer
    [peer][b][slot in peer][D] through the reorder map (forward_reorder_functor.cu:43-57) and writes
    the embedding gradients straight in backward_reorder's send layout
    (hctr_interaction_bwd_indexed_scatter): output and gradients bit-equal to forward_reorder ->
    interaction -> backward_reorder"""
    import torch
    import hugectr_amd as ha
    dt = getattr(torch, dtype_name)
    g = torch.Generator(device="cuda").manual_seed(B + n_emb + world)
    recv = torch.randn(B * n_emb * W, device="cuda", generator=g).to(dt)
    mlp = torch.randn(B, W, device="cuda", generator=g).to(dt)
    s_of = [n_emb // world + (1 if r < n_emb % world else 0) for r in range(world)]
    base = [B * sum(s_of[:r]) for r in range(world)]
    row = torch.empty((B, n_emb), dtype=torch.int32)
    for b in range(B):
        for s in range(n_emb):
            r, j = s % world, s // world
            row[b, s] = base[r] + b * s_of[r] + j
    row = row.cuda()
    E = ha.forward_reorder(recv, B, n_emb, W, world).requires_grad_()
    m1, m2 = mlp.clone().requires_grad_(), mlp.clone().requires_grad_()
    out_d = ha.interaction(m1, E)
    got = {}
    out_i = ha.interaction_indexed(m2, recv.view(-1, W), row, on_emb_grad=lambda d: got.update(g=d),
                                   scatter_grad=True)
    assert torch.equal(out_i, out_d)
    top = torch.randn(out_d.shape, device="cuda", generator=g).to(dt)
    out_d.backward(top)
    out_i.backward(top)
    assert torch.equal(m1.grad, m2.grad)
    want = ha.backward_reorder(E.grad.contiguous(), B, n_emb, W, world)
    assert torch.equal(got["g"].reshape(-1), want.reshape(-1))


@pytest.mark.parametrize("dtype_name", ["float16", "bfloat16"])
@pytest.mark.parametrize("B,S,D", [(300, 26, 128), (65, 5, 64), (129, 31, 32), (70, 7, 16)])
def test_gather_fused_into_interaction_equals_pool_then_interaction(dtype_name, B, S, D):
    """hctr_emb_forward_interaction (table rows read through value_index into the interaction's
    tile, pooled vectors written once) == hctr_emb_forward + hctr_interaction_fwd, bit for bit:
    pooled vectors, interaction output, both gradients; training batch (keys inserted) and an
    evaluation batch with unseen keys (-> zeros)"""
    import torch
    import hugectr_amd as ha
    from hugectr_amd import _lib
    dt = getattr(torch, dtype_name)
    vps = 50
    opt = ha.OptParams(optimizer=_lib.OPT_SGD, lr=0.1, atomic_update=False)
    emb = ha.SparseEmbeddingHash(_lib.EMB_LOCALIZED, B, B, S * vps, D, S, S, 0, opt,
                                 slot_size_array=[vps] * S, out_dtype=dt)
    emb.init_params()
    emb.table().normal_(0, 1)
    emb.table()[3, :4] = torch.tensor([-0.0, 0.0, 1e-9, -1e-9], device="cuda")  # signed zeros, tiny
    g = torch.Generator(device="cuda").manual_seed(B + S)
    ro = torch.arange(B * S + 1, dtype=torch.int64, device="cuda")
    off = torch.arange(S, device="cuda") * vps
    for is_train, hi in ((True, vps - 5), (False, vps)):  # the eval batch meets unseen keys
        keys = (torch.randint(0, hi, (B, S), device="cuda", generator=g) + off).reshape(-1)
        mlp = torch.randn(B, D, device="cuda", generator=g).to(dt)
        pooled = emb.forward(is_train, ro, keys).clone()
        m1 = mlp.clone().requires_grad_()
        e1 = pooled.clone().requires_grad_()
        out_d = ha.interaction(m1, e1)
        emb.index(is_train, ro, keys)
        m2 = mlp.clone().requires_grad_()
        got = {}
        out_f = ha.interaction_gather(m2, emb, is_train, on_emb_grad=lambda d: got.update(dE=d))
        assert torch.equal(out_f.view(torch.int16), out_d.view(torch.int16))
        saved = out_f.grad_fn.saved_tensors[1]  # the pooled vectors the fused kernel wrote
        assert torch.equal(saved.view(torch.int16), pooled.view(torch.int16))
        top = torch.randn(out_d.shape, device="cuda", generator=g).to(dt)
        out_d.backward(top)
        out_f.backward(top)
        assert torch.equal(m1.grad, m2.grad) and torch.equal(e1.grad, got["dE"])
        if not is_train:
            assert (pooled.float().abs().sum(-1) == 0).any(), "no evaluation miss in this batch"


def test_fused_mlp_flat_mode_equals_parameter_mode():
    """flatten(): gradients written straight into the flat buffer and the fused SGD + shadow-refresh
    kernel give the same weights as torch.optim.SGD on the unflattened module"""
    import copy
    import torch
    from hugectr_amd.dense import FusedMLP
    torch.manual_seed(11)
    a = FusedMLP([40, 128, 64, 1], last_relu=False).cuda()
    b = copy.deepcopy(a)
    a.refresh_shadow()
    b.flatten()
    opt = torch.optim.SGD(a.parameters(), lr=0.05)
    for step in range(3):
        x = torch.randn(4096, 40, device="cuda")
        ya, yb = a(x), b(x)
        # same math; the library may pick another GEMM kernel for the flat views, and after a step
        # the masters agree to an fp32 ulp, so single bf16 weights may round differently
        assert torch.allclose(ya.float(), yb.float(), rtol=2e-2, atol=2e-2)
        gy = torch.randn_like(ya)
        ya.backward(gy)
        yb.backward(gy)
        for p, gw in zip(list(a.weights) + list(a.biases), b._gw + b._gb):
            assert (p.grad - gw).norm() <= 2e-2 * p.grad.norm() + 1e-6, "flat gradient differs"
        assert all(p.grad is None for p in b.parameters())
        opt.step()
        opt.zero_grad(set_to_none=True)
        a.refresh_shadow()
        b.sgd_step(0.05)
        for pa, pb in zip(a.parameters(), b.parameters()):
            assert (pa - pb).norm() <= 1e-3 * pa.norm() + 1e-6
        for wa, wb in zip(a._w16 + a._b16, b._w16 + b._b16):
            # the masters agree to one fp32 ulp (fma vs mul+sub), so a bf16 rounding may flip
            assert (wa.float() - wb.float()).norm() <= 1e-2 * wa.float().norm() + 1e-6


@pytest.mark.parametrize("B,K", [(4096, 256), (1000, 64), (777, 520), (300, 2048), (5, 4)])
@pytest.mark.parametrize("dtype", ["bf16", "f16"])
def test_logit_head_matches_torch(B, K, dtype):
    """last FC layer (K -> 1) + BCE loss + dx / dw / db in one kernel vs the fp32 formulas"""
    import torch
    from hugectr_amd._lib import check, lib, ptr, stream_ptr
    tdt = torch.bfloat16 if dtype == "bf16" else torch.float16
    code = 2 if dtype == "bf16" else 1
    g = torch.Generator(device="cuda").manual_seed(B + K)
    x = torch.randn((B, K), device="cuda", generator=g).to(tdt)
    w = (torch.randn((1, K), device="cuda", generator=g) / K ** 0.5).to(tdt)
    b = torch.tensor([0.1], device="cuda").to(tdt)
    y = (torch.rand((B, 1), device="cuda", generator=g) < 0.4).float()
    scale = 1.0 / B
    dx = torch.empty_like(x)
    dw = torch.empty((1, K), dtype=torch.float32, device="cuda")
    db = torch.empty(1, dtype=torch.float32, device="cuda")
    loss = torch.empty(1, dtype=torch.float32, device="cuda")
    ws = torch.empty(lib.hctr_logit_head_workspace_bytes(K) // 4, dtype=torch.float32, device="cuda")
    check(lib.hctr_logit_head(B, K, ptr(x), ptr(w), ptr(b), ptr(y), scale, ptr(dx), ptr(dw), ptr(db),
                              ptr(loss), ptr(ws), code, stream_ptr()))
    xf, wf = x.double(), w.double()
    z = xf @ wf.t() + b.double()
    want_loss = torch.nn.functional.binary_cross_entropy_with_logits(z, y.double())
    dz = (torch.sigmoid(z) - y.double()) * scale
    assert abs(float(loss) - float(want_loss)) < 2e-6 * max(1.0, abs(float(want_loss)))
    assert torch.allclose(dw.double(), dz.t() @ xf, rtol=1e-4, atol=1e-7)
    assert abs(float(db) - float(dz.sum())) < 1e-6
    tol = 2 ** -7 if dtype == "bf16" else 2 ** -10          # one rounding of the 16-bit store
    atol = 1e-9 if dtype == "bf16" else 6e-8                # (fp16: gradients this small are subnormal)
    assert torch.allclose(dx.double(), dz * wf, rtol=tol, atol=atol)
    # deterministic: a second launch gives the same bits
    dw2, loss2 = torch.empty_like(dw), torch.empty_like(loss)
    check(lib.hctr_logit_head(B, K, ptr(x), ptr(w), ptr(b), ptr(y), scale, None, ptr(dw2), ptr(db),
                              ptr(loss2), ptr(ws), code, stream_ptr()))
    assert torch.equal(dw, dw2) and torch.equal(loss, loss2)


def test_fused_mlp_bce_head_equals_the_unfused_tower():
    """FusedMLP.forward_bce (head fused) vs forward + bce_with_logits + backward: same gradients up
    to the 16-bit rounding of the logits the unfused path goes through"""
    import torch
    from hugectr_amd.dense import FusedMLP, bce_with_logits
    torch.manual_seed(0)
    B = 2048
    a = FusedMLP([96, 64, 32, 1], last_relu=False).cuda()
    b = FusedMLP([96, 64, 32, 1], last_relu=False).cuda()
    b.load_state_dict(a.state_dict())
    a.flatten()
    b.flatten()
    x = torch.randn((B, 96), device="cuda").bfloat16()
    y = (torch.rand((B, 1), device="cuda") < 0.5).float()
    xa, xb = x.clone().requires_grad_(True), x.clone().requires_grad_(True)
    la = a.forward_bce(xa, y, 1.0 / B)
    la.backward()
    logit = b(xb)
    lb, dlogit = bce_with_logits(logit, y, 1.0 / B)
    logit.backward(dlogit)
    assert abs(float(la) - float(lb)) < 2e-3
    assert torch.allclose(a.flat_g, b.flat_g, rtol=5e-2, atol=2e-4)
    assert torch.allclose(xa.grad.float(), xb.grad.float(), rtol=5e-2, atol=2e-5)


@pytest.mark.parametrize("B,K,N", [(4096, 13, 512), (1000, 13, 256), (777, 16, 128), (37, 1, 4),
                                   (9000, 7, 508), (300, 7, 136), (5000, 15, 384), (3, 13, 512)])
@pytest.mark.parametrize("dtype", ["bf16", "f16"])
@pytest.mark.parametrize("bwd", ["valu", "mfma"])
def test_skinny_first_layer_matches_torch(B, K, N, dtype, bwd, monkeypatch):
    """first MLP layer with a handful of input features: forward and the fused
    ReLU-backward + dw + db (vector-ALU and matrix-core forms; the latter falls back to the former
    for K = 16 or N % 8 != 0) against fp64 formulas on the same 16-bit operands"""
    import torch
    from hugectr_amd._lib import check, lib, ptr, stream_ptr
    monkeypatch.setenv("HCTR_SKINNY_BWD", bwd)
    tdt = torch.bfloat16 if dtype == "bf16" else torch.float16
    code = 2 if dtype == "bf16" else 1
    g = torch.Generator(device="cuda").manual_seed(B + K + N)
    x = torch.randn((B, K), device="cuda", generator=g)
    w = (torch.randn((N, K), device="cuda", generator=g) / K ** 0.5).to(tdt)
    b = (torch.randn(N, device="cuda", generator=g) * 0.1).to(tdt)
    y = torch.empty((B, N), dtype=tdt, device="cuda")
    check(lib.hctr_skinny_fc_fwd(B, K, N, ptr(x), ptr(w), ptr(b), ptr(y), code, stream_ptr()))
    x16 = x.to(tdt).double()
    want = torch.relu(x16 @ w.double().t() + b.double())
    tol = 2 ** -7 if dtype == "bf16" else 2 ** -10
    assert torch.allclose(y.double(), want, rtol=tol, atol=1e-6)
    # the rounded result itself: equal to rounding the fp64 value except at rounding ties
    assert (y != want.to(tdt)).float().mean() < 1e-3

    dy = (torch.randn((B, N), device="cuda", generator=g) / B).to(tdt)
    dw = torch.empty((N, K), dtype=torch.float32, device="cuda")
    db = torch.empty(N, dtype=torch.float32, device="cuda")
    ws = torch.empty(lib.hctr_skinny_fc_bwd_workspace_bytes(N) // 4, dtype=torch.float32,
                     device="cuda")
    check(lib.hctr_skinny_fc_bwd(B, K, N, ptr(x), ptr(dy), ptr(y), ptr(dw), ptr(db), ptr(ws), code,
                                 stream_ptr()))
    dz = dy.double() * (y > 0)
    want_dw, want_db = dz.t() @ x16, dz.sum(0)
    bound_w = 2e-6 * (dz.abs().t() @ x16.abs()) + 1e-12
    assert bool(((dw.double() - want_dw).abs() <= bound_w).all())
    assert bool(((db.double() - want_db).abs() <= 2e-6 * dz.abs().sum(0) + 1e-12).all())
    dw2, db2 = torch.empty_like(dw), torch.empty_like(db)
    check(lib.hctr_skinny_fc_bwd(B, K, N, ptr(x), ptr(dy), ptr(y), ptr(dw2), ptr(db2), ptr(ws), code,
                                 stream_ptr()))
    assert torch.equal(dw, dw2) and torch.equal(db, db2)


@pytest.mark.parametrize("fwd", ["gemm", "hip"])
def test_fused_mlp_skinny_first_layer_equals_the_gemm_path(monkeypatch, fwd):
    """FusedMLP on fp32 dense features: the few-features kernels for layer 1 against the library
    GEMM path (HCTR_SKINNY_FC=0) -- same 16-bit activations up to accumulation order, same
    gradients in the flat buffer"""
    import torch
    from hugectr_amd.dense import FusedMLP
    torch.manual_seed(1)
    B = 4096
    a = FusedMLP([13, 512, 256, 128], last_relu=True).cuda()
    b = FusedMLP([13, 512, 256, 128], last_relu=True).cuda()
    b.load_state_dict(a.state_dict())
    a.flatten()
    b.flatten()
    x = torch.rand((B, 13), device="cuda")
    gy = (torch.randn((B, 128), device="cuda") / B).bfloat16()
    assert a._skinny_first(x)
    monkeypatch.setenv("HCTR_SKINNY_FC_FWD", fwd)
    ya = a(x)
    ya.backward(gy)
    monkeypatch.setenv("HCTR_SKINNY_FC", "0")
    assert not b._skinny_first(x)
    yb = b(x)
    yb.backward(gy)
    assert (ya.float() - yb.float()).norm() <= 2e-3 * yb.float().norm()
    ga, gb = a.flat_g, b.flat_g
    assert (ga - gb).norm() <= 5e-3 * gb.norm()
    # layer-1 gradients specifically (the part that changed hands)
    for va, vb in ((a._gw[0], b._gw[0]), (a._gb[0], b._gb[0])):
        assert (va - vb).norm() <= 5e-3 * vb.norm() + 1e-9
