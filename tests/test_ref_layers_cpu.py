"""oracle/hctr_oracle.c's Interaction and MultiCross references (what the GPU tests of
hctr_interaction_* / hctr_cross_* compare against) against the REFERENCE'S OWN CPU references of
those layers -- the host-side statement blocks of its layer tests
(R/test/utest/core23_layer_test/interaction_layer_test.cpp:95-282,
multi_cross_layer_test.cpp:152-432), cut out of those files by the build recipe and compiled into
oracle/_ref/libref_layers.so (oracle/Makefile `ref`, oracle/ref_layers_shim.cpp)."""
import ctypes
import os

import numpy as np
import pytest

from oracle import pyoracle as po

LIB = os.path.join(os.path.dirname(__file__), "..", "oracle", "_ref", "libref_layers.so")
pytestmark = pytest.mark.skipif(not os.path.exists(LIB),
                                reason="oracle/_ref not built (needs the reference checkout)")


def _ref():
    L = ctypes.CDLL(LIB)
    P, Z, I = ctypes.c_void_p, ctypes.c_size_t, ctypes.c_int
    L.ref_interaction.argtypes = [Z, Z, Z, P, P, P, P]
    L.ref_cross.argtypes = [Z, Z, I, Z, P, P, P, P, P, P, P, P]
    return L


def _p(a):
    return a.ctypes.data_as(ctypes.c_void_p) if a is not None else None


@pytest.mark.parametrize("B,n_emb,W", [(7, 3, 8), (5, 26, 16), (3, 26, 128), (4, 1, 4)])
def test_interaction_oracle_matches_the_reference_test_code(B, n_emb, W):
    L = _ref()
    rng = np.random.default_rng(B * 100 + n_emb)
    mlp = rng.standard_normal((B, W)).astype(np.float32)
    emb = rng.standard_normal((B, n_emb, W)).astype(np.float32)
    n_ins = n_emb + 1
    out_w = W + n_ins * (n_ins - 1) // 2 + 1
    top_grad = rng.standard_normal((B, out_w)).astype(np.float32)
    m, e = mlp.copy(), emb.copy()
    top = np.empty((B, out_w), np.float32)
    assert L.ref_interaction(B, n_emb, W, _p(m), _p(e), _p(top), _p(top_grad)) == 0
    want = po.interaction_fwd(mlp, emb)
    # q13: [mlp | strict lower triangle, row major | one zero column]; same fp32 accumulation order
    assert np.array_equal(want, top)
    assert (top[:, -1] == 0).all() and np.array_equal(top[:, :W], mlp)
    mg, eg = po.interaction_bwd(mlp, emb, top_grad)
    # the reference accumulates ((dM + dM^T) x) term by term in float, the oracle likewise; the
    # mlp gradient ADDS the pass-through slice of the top gradient
    np.testing.assert_allclose(eg, e, rtol=1e-5, atol=1e-5)
    np.testing.assert_allclose(mg, m, rtol=1e-5, atol=1e-5)


@pytest.mark.parametrize("B,w,L_", [(6, 10, 1), (9, 429, 6), (4, 33, 3)])
def test_cross_v1_oracle_matches_the_reference_test_code(B, w, L_):
    L = _ref()
    rng = np.random.default_rng(w)
    x = np.clip(rng.standard_normal((B, w)), -0.9, 0.9).astype(np.float32)
    k = np.clip(rng.standard_normal((L_, w)), -1, 1).astype(np.float32)
    b = (rng.standard_normal((L_, w)) * 0.01).astype(np.float32)
    g = rng.standard_normal((B, w)).astype(np.float32)
    out, ig = np.empty((B, w), np.float32), np.empty((B, w), np.float32)
    kg, bg = np.empty((L_, w), np.float32), np.empty((L_, w), np.float32)
    assert L.ref_cross(B, w, L_, 0, _p(x), _p(k), _p(b), _p(g), _p(out), _p(ig), _p(kg), _p(bg)) == 0
    outputs, hiddens = po.cross_v1_fwd(x, k, b)
    np.testing.assert_allclose(outputs[-1], out, rtol=1e-6, atol=1e-6)
    ig2, kg2, bg2 = po.cross_v1_bwd(x, k, outputs, hiddens, g)
    np.testing.assert_allclose(ig2, ig, rtol=1e-5, atol=1e-5)
    np.testing.assert_allclose(kg2, kg, rtol=1e-5, atol=1e-5)
    np.testing.assert_allclose(bg2, bg, rtol=1e-5, atol=1e-5)


@pytest.mark.parametrize("B,w,p,L_", [(6, 10, 4, 1), (8, 48, 16, 3)])
def test_cross_v2_oracle_matches_the_reference_test_code(B, w, p, L_):
    L = _ref()
    rng = np.random.default_rng(w + p)
    x = np.clip(rng.standard_normal((B, w)), -0.9, 0.9).astype(np.float32)
    U = (rng.standard_normal((L_, w, p)) * 0.3).astype(np.float32)
    V = (rng.standard_normal((L_, p, w)) * 0.3).astype(np.float32)
    b = (rng.standard_normal((L_, w)) * 0.01).astype(np.float32)
    g = rng.standard_normal((B, w)).astype(np.float32)
    # reference weight order: per layer U_l [w][p] then V_l [p][w]
    kern = np.concatenate([np.concatenate([U[l].reshape(-1), V[l].reshape(-1)]) for l in range(L_)])
    out, ig = np.empty((B, w), np.float32), np.empty((B, w), np.float32)
    kg, bg = np.empty_like(kern), np.empty((L_, w), np.float32)
    assert L.ref_cross(B, w, L_, p, _p(x), _p(kern), _p(b), _p(g), _p(out), _p(ig), _p(kg),
                       _p(bg)) == 0
    outputs, hiddens, XUs = po.cross_v2_fwd(x, U, V, b)
    np.testing.assert_allclose(outputs[-1], out, rtol=1e-5, atol=1e-6)
    ig2, dU, dV, db = po.cross_v2_bwd(x, U, V, outputs, hiddens, XUs, g)
    kg = kg.reshape(L_, 2, w * p)
    np.testing.assert_allclose(ig2, ig, rtol=1e-4, atol=1e-5)
    np.testing.assert_allclose(dU.reshape(L_, -1), kg[:, 0], rtol=1e-4, atol=1e-5)
    np.testing.assert_allclose(dV.reshape(L_, -1), kg[:, 1], rtol=1e-4, atol=1e-5)
    np.testing.assert_allclose(db, bg, rtol=1e-4, atol=1e-5)
