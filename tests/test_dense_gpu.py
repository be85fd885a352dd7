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


def test_cross_v2_matches_oracle(oracle):
    import torch
    import hugectr_amd as ha
    rng = np.random.default_rng(11)
    B, w, p, L = 64, 96, 16, 3
    x0 = (rng.standard_normal((B, w)) * 0.1).astype(np.float32)
    layer = ha.MultiCrossLayer(w, L, p).cuda()
    U = layer.U.detach().cpu().numpy(); V = layer.V.detach().cpu().numpy()
    b = layer.biases.detach().cpu().numpy()
    out = layer(torch.from_numpy(x0).cuda())
    outs, _, _ = oracle.cross_v2_fwd(x0, U, V, b)
    assert_close(out.detach().cpu().numpy(), outs[-1], 1e-3, 1e-4, "cross v2 fwd")


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
