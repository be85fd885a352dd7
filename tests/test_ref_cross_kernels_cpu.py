"""The REFERENCE's device code of MultiCrossLayer (R/HugeCTR/src/layers/multi_cross_layer.cu:54-563:
its kernels with the host functions that launch them, and the v1 layer's MultiCrossForwardFunctor /
MultiCrossBackwardFunctor, :582-600 / :698-732, that compose them), cut out of the checkout into
oracle/_ref/libref_cross_kernels.so and executed by the host interpreter of tests/emu, next to

  * oracle/pyoracle.py cross_v1_fwd / cross_v1_bwd (so far pinned to the CPU code inside the
    reference's gtest file only, tests/test_ref_layers_cpu.py), and
  * this repo's kernel source: hctr_cross_v1_fwd / hctr_cross_v1_bwd of
    hugectr_amd/csrc/dense_ops.hip, stepped through by the same interpreter.

v1: the dot product x_l . w_l is a cuBLAS GEMV in the reference (order unspecified) and a wavefront
tree here, so outputs agree within fp32 summation error; everything after the dot product is the
same sequence of roundings (row_scaling, matrix_add, matrix_vec_add as three kernels there, three
statements here).  v2: the layer's own kernels are the elementwise ones around cublasLt GEMMs;
their binary16 forms round ONCE per multiply-add when the array length is a multiple of 8
(__hfma2) and TWICE otherwise (operator* then operator+ on __half) -- the 16-bit cross layer of
hugectr_amd/layers.py (_CrossV2Fn: addcmul in the activation type) has the single rounding, the
form every MLPerf shape takes."""
import ctypes
import os
import sys

import numpy as np
import pytest

from oracle import pyoracle as po

HERE = os.path.dirname(os.path.abspath(__file__))
LIB = os.path.join(HERE, "..", "oracle", "_ref", "libref_cross_kernels.so")
pytestmark = pytest.mark.skipif(not os.path.exists(LIB),
                                reason="oracle/_ref not built (needs the reference checkout)")
sys.path.insert(0, os.path.join(HERE, "emu"))
import emu  # noqa: E402

f32, f16 = np.float32, np.float16


def _p(a):
    return a.ctypes.data_as(ctypes.c_void_p)


@pytest.fixture(scope="module")
def ref():
    L = ctypes.CDLL(LIB)
    I, P = ctypes.c_int, ctypes.c_void_p
    L.refcross_v1_fwd.argtypes = [I, I, I, P, P, P, P, P]
    L.refcross_v1_bwd.argtypes = [I, I, I, P, P, P, P, P, P, P, P]
    L.refcross_v2_dot_add.argtypes = [I, I, I, P, P, P, P]
    L.refcross_v2_mul_fma3.argtypes = [I, I, I, P, P, P, P, P]
    return L


@pytest.fixture(scope="module")
def elib():
    return emu.load_under_test()


def _ulps32(a, b):
    ia = a.view(np.int32).astype(np.int64)
    ib = b.view(np.int32).astype(np.int64)
    ia = np.where(ia < 0, -(ia & 0x7FFFFFFF), ia)
    ib = np.where(ib < 0, -(ib & 0x7FFFFFFF), ib)
    return np.abs(ia - ib)


# (B, w, L): the DCN sample's width 429 = 13 + 26 * 16 with 6 layers, widths below /
# above one wavefront, a width that is no multiple of anything
@pytest.mark.parametrize("B,w,L_", [(5, 429, 6), (9, 64, 2), (3, 1000, 2), (4, 33, 4), (1, 1, 1)])
def test_v1_forward_backward_next_to_the_reference_device_code(ref, elib, B, w, L_):
    rng = np.random.default_rng(B * 7919 + w * 31 + L_)
    x0 = rng.standard_normal((B, w)).astype(f32)
    k = (rng.standard_normal((L_, w)) / np.sqrt(w)).astype(f32)
    b = (rng.standard_normal((L_, w)) * 0.1).astype(f32)
    g = rng.standard_normal((B, w)).astype(f32)
    # ---- forward -------------------------------------------------------------------------------
    r_out = np.full((L_, B, w), np.nan, f32)
    r_hid = np.full((L_, B), np.nan, f32)
    ref.refcross_v1_fwd(B, w, L_, _p(x0), _p(k), _p(b), _p(r_out), _p(r_hid))
    o_out, o_hid = po.cross_v1_fwd(x0, k, b)
    h_out = np.full((L_, B, w), np.nan, f32)
    h_hid = np.full((L_, B), np.nan, f32)
    emu.check(elib, elib.hctr_cross_v1_fwd(B, w, L_, _p(x0), _p(k), _p(b), _p(h_out), _p(h_hid), None))
    scale = np.abs(r_out).max()
    for name, out, hid in (("oracle", o_out, o_hid), ("HIP source", h_out, h_hid)):
        np.testing.assert_allclose(hid, r_hid, rtol=2e-5, atol=2e-6, err_msg=f"{name}: x_l . w_l")
        np.testing.assert_allclose(out, r_out, rtol=0, atol=3e-6 * scale, err_msg=f"{name}: outputs")
    # the steps after the dot product are the reference's roundings: fed the REFERENCE's hidden
    # value, x0 * h + x_l + b is the reference's output bit for bit
    xl = x0
    for l in range(L_):
        nxt = ((x0 * r_hid[l][:, None]).astype(f32) + xl).astype(f32) + b[l]
        np.testing.assert_array_equal(nxt.astype(f32), r_out[l])
        xl = r_out[l]
    # ---- backward (all three sides from the REFERENCE's forward arrays) ---------------------------
    r_ig = np.full((B, w), np.nan, f32)
    r_kg = np.zeros((L_, w), f32)
    r_bg = np.zeros((L_, w), f32)
    ref.refcross_v1_bwd(B, w, L_, _p(x0), _p(k), _p(r_out), _p(r_hid), _p(g), _p(r_ig), _p(r_kg), _p(r_bg))
    o_ig, o_kg, o_bg = po.cross_v1_bwd(x0, k, r_out, r_hid, g)
    h_ig = np.full((B, w), np.nan, f32)
    h_kg = np.full((L_, w), np.nan, f32)
    h_bg = np.full((L_, w), np.nan, f32)
    ws = np.zeros(max(elib.hctr_cross_v1_bwd_workspace_bytes(B, w, L_) // 4, 1), f32)
    emu.check(elib, elib.hctr_cross_v1_bwd(B, w, L_, _p(x0), _p(k), _p(r_out), _p(r_hid), _p(g), _p(h_ig),
                                           _p(h_kg), _p(h_bg), _p(ws), None))
    for name, ig, kg, bg in (("oracle", o_ig, o_kg, o_bg), ("HIP source", h_ig, h_kg, h_bg)):
        np.testing.assert_allclose(ig, r_ig, rtol=0, atol=2e-5 * max(np.abs(r_ig).max(), 1.0),
                                   err_msg=f"{name}: input gradient")
        np.testing.assert_allclose(kg, r_kg, rtol=0, atol=2e-5 * max(np.abs(r_kg).max(), 1.0),
                                   err_msg=f"{name}: kernel gradients")
        np.testing.assert_allclose(bg, r_bg, rtol=0, atol=2e-5 * max(np.abs(r_bg).max(), 1.0),
                                   err_msg=f"{name}: bias gradients")


@pytest.mark.parametrize("B,w", [(4, 3456), (3, 40), (5, 13), (1, 7)])
def test_v2_fp32_elementwise_kernels(ref, elib, B, w):
    rng = np.random.default_rng(B * 100 + w)
    x0, xl, h, dy = (rng.standard_normal((B, w)).astype(f32) for _ in range(4))
    # forward: x0 .* h + x_l (vector_fma4).  (The HIP side of this step is the epilogue of the
    # library's own GEMM, hctr_gemm_nt16 epilogue 1: tests/test_dense_gpu.py, against fp64.)
    r = np.full((B, w), np.nan, f32)
    ref.refcross_v2_dot_add(B, w, 0, _p(r), _p(h), _p(x0), _p(xl))
    np.testing.assert_array_equal(r, (h * x0).astype(f32) + xl)
    # backward: S0 = dY .* X0, dX += dY .* H (vector_mul_fma3_align)
    s0 = np.full((B, w), np.nan, f32)
    acc = rng.standard_normal((B, w)).astype(f32)
    acc0 = acc.copy()
    ref.refcross_v2_mul_fma3(B, w, 0, _p(s0), _p(acc), _p(dy), _p(x0), _p(h))
    np.testing.assert_array_equal(s0, dy * x0)
    np.testing.assert_array_equal(acc, acc0 + (dy * h).astype(f32))


# len % 8 == 0: the paired-half kernels (one rounding per multiply-add); else the generic ones
@pytest.mark.parametrize("B,w", [(4, 3456), (8, 24), (3, 40), (5, 13), (1, 7)])
def test_v2_binary16_elementwise_kernels_round_as_the_16_bit_cross_layer_does(ref, B, w):
    import torch
    rng = np.random.default_rng(B * 100 + w + 1)
    x0, xl, h, dy, acc = (rng.standard_normal((B, w)).astype(f16) for _ in range(5))
    aligned = (B * w) % 8 == 0
    f64 = np.float64
    r = np.zeros((B, w), f16)
    ref.refcross_v2_dot_add(B, w, 1, _p(r), _p(h), _p(x0), _p(xl))
    once = (h.astype(f64) * x0.astype(f64) + xl.astype(f64)).astype(f16)
    twice = ((h.astype(f32) * x0.astype(f32)).astype(f16).astype(f32) + xl.astype(f32)).astype(f16)
    np.testing.assert_array_equal(r, once if aligned else twice)
    # in place (out == x_l: vector_fma3_align8)
    r_in = xl.copy()
    ref.refcross_v2_dot_add(B, w, 1, _p(r_in), _p(h), _p(x0), _p(r_in))
    np.testing.assert_array_equal(r_in, once if aligned else twice)
    # _CrossV2Fn's forward statement (hugectr_amd/layers.py): xl = addcmul(xl, x0, h)
    t = torch.addcmul(torch.from_numpy(xl), torch.from_numpy(x0), torch.from_numpy(h)).numpy()
    d = np.abs(t.view(np.int16).astype(np.int32) - once.view(np.int16).astype(np.int32))
    assert d.max() <= 1 and np.mean(d == 0) > 0.999  # (fp32 intermediate: a double rounding is rare)
    # backward: S0 = dY .* X0 (one rounding either way), dX += dY .* H
    s0 = np.zeros((B, w), f16)
    a = acc.copy()
    ref.refcross_v2_mul_fma3(B, w, 1, _p(s0), _p(a), _p(dy), _p(x0), _p(h))
    np.testing.assert_array_equal(s0, (dy.astype(f32) * x0.astype(f32)).astype(f16))
    once = (dy.astype(f64) * h.astype(f64) + acc.astype(f64)).astype(f16)
    twice = (acc.astype(f32) + (dy.astype(f32) * h.astype(f32)).astype(f16).astype(f32)).astype(f16)
    np.testing.assert_array_equal(a, once if aligned else twice)


@pytest.mark.parametrize("first", [1, 0])
@pytest.mark.parametrize("B,w", [(130, 3456), (257, 24), (5, 40), (128, 2056)])
def test_v2_backward_step_of_the_hip_source_next_to_fused_mul_fma3(ref, elib, B, w, first):
    """hctr_cross_v2_bwd_step (dense_ops.hip: S0 = dY .* X0, dX += dY .* H, db = column sums of S0 in
    one pass) next to the reference's fused_mul_fma3 launch (its paired-half kernel: B * w % 8 == 0)"""
    from hugectr_amd import _lib
    rng = np.random.default_rng(B * 100 + w + first)
    x0, h, dy, acc = (rng.standard_normal((B, w)).astype(f16) for _ in range(4))
    r_s0 = np.zeros((B, w), f16)
    r_acc = np.zeros((B, w), f16) if first else acc.copy()
    ref.refcross_v2_mul_fma3(B, w, 1, _p(r_s0), _p(r_acc), _p(dy), _p(x0), _p(h))
    g_s0 = np.full((B, w), np.nan, f16)
    g_acc = np.full((B, w), np.nan, f16) if first else acc.copy()  # (first: never read)
    db = np.full(w, np.nan, f32)
    ws = np.zeros(elib.hctr_cross_v2_bwd_step_workspace_bytes(B, w) // 4, f32)
    emu.check(elib, elib.hctr_cross_v2_bwd_step(B, w, _p(dy), _p(x0), _p(h), _p(g_acc), _p(g_s0), _p(db),
                                                _p(ws), first, _lib.F16, None))
    np.testing.assert_array_equal(g_s0, r_s0)
    # the accumulator: exact product + addend rounded to fp32, then to binary16 -- the fused
    # instruction rounds once; the two differ in a rare tie case by one ulp
    d = np.abs(g_acc.view(np.int16).astype(np.int32) - r_acc.view(np.int16).astype(np.int32))
    assert d.max() <= 1 and np.mean(d == 0) > 0.999
    if first:
        np.testing.assert_array_equal(g_acc, r_acc)  # (no addend: one rounding either way)
    np.testing.assert_allclose(db, r_s0.astype(np.float64).sum(0), rtol=1e-5, atol=1e-4)
