"""The oracle's dot interaction AND the HIP interaction kernels' source against the REFERENCE'S
DEVICE CODE of InteractionLayer (R/HugeCTR/src/layers/interaction_layer.cu:31-955): the fused fp16
kernels dotBasedInteractFwdKernel / BwdKernel (aligned and NonAligned forms, nvcuda::wmma tiles)
through the reference's own launch wrappers dotBasedInteractFwd / Bwd, and the generic fp32 path's
concat_kernel + gather_concat_fprop_kernel around the X * X^T product -- cut out of the checkout and
executed by the host interpreter of tests/emu (oracle/_ref/libref_interaction.so, oracle/Makefile
`ref`; wmma by its documented contract).  The reference's tests hold only a CPU restatement of the
layer (tests/test_ref_layers_cpu.py pins the oracle to it); here the kernels themselves run: output
layout [mlp | lower triangle of X X^T row by row | one zero] element for element, fp16 values
within one binary16 ulp (the matrix cores' summation order is unspecified)."""
import ctypes
import os
import sys

import numpy as np
import pytest

from util import assert_close

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "emu"))
import emu  # noqa: E402

LIB = os.path.join(os.path.dirname(__file__), "..", "oracle", "_ref", "libref_interaction.so")
pytestmark = pytest.mark.skipif(not os.path.exists(LIB),
                                reason="oracle/_ref not built (needs the reference checkout)")


def _p(a):
    return a.ctypes.data_as(ctypes.c_void_p)


@pytest.fixture(scope="module")
def ref():
    L = ctypes.CDLL(LIB)
    P, U = ctypes.c_void_p, ctypes.c_uint
    L.refinter_fwd16.argtypes = [P, P, P, U, U, U]
    L.refinter_bwd16.argtypes = [P, P, P, U, U, U]
    L.refinter_fwd32.argtypes = [P, P, P, ctypes.c_int, ctypes.c_int, ctypes.c_int]
    return L


def _inputs(rng, B, n_emb, W, dt):
    mlp = (rng.standard_normal((B, W)) * 0.5).astype(dt)
    emb = (rng.standard_normal((B, n_emb, W)) * 0.5).astype(dt)
    return mlp, emb


# (B, n_emb, W): W % 8 == 0 and an output length % 8 == 0 take the aligned kernels (DLRM: 26 + 1
# vectors of 128 -> 128 + 351 + 1 = 480), everything else the NonAligned ones
SHAPES16 = [(5, 26, 128), (3, 26, 64), (9, 7, 16), (4, 3, 32), (2, 12, 24), (6, 1, 8), (3, 30, 128)]


@pytest.mark.parametrize("B,n_emb,W", SHAPES16)
def test_fp16_forward_equals_the_reference_fused_kernel(oracle, ref, B, n_emb, W):
    from hugectr_amd import _lib
    rng = np.random.default_rng(B * 1000 + n_emb * 10 + W)
    mlp, emb = _inputs(rng, B, n_emb, W, np.float16)
    n_ins = n_emb + 1
    out_len = W + n_ins * (n_ins - 1) // 2 + 1
    got = np.full((B, out_len), np.nan, np.float16)
    ref.refinter_fwd16(_p(mlp), _p(emb), _p(got), B, n_ins, W)
    want = oracle.interaction_fwd(mlp.astype(np.float32), emb.astype(np.float32))
    assert np.array_equal(got[:, :W], mlp), "bottom-MLP vector copied through"
    assert not got[:, -1].any(), "padding element"
    assert np.isfinite(got.astype(np.float32)).all()
    # pair order and values: fp32 dot products of the binary16 inputs, rounded once
    assert_close(got.astype(np.float32), want, 1e-3, 1e-4, "reference kernel vs oracle")
    if emu.available():
        lib = emu.load_under_test()
        hip = np.full((B, out_len), np.nan, np.float16)
        emu.check(lib, lib.hctr_interaction_fwd(B, n_emb, W, _p(mlp), _p(emb), _p(hip), _lib.F16, None))
        assert np.isfinite(hip.astype(np.float32)).all()
        assert_close(hip.astype(np.float32), got.astype(np.float32), 1e-3, 1e-4, "hip vs reference")
        exact = (hip.view(np.uint16) == got.view(np.uint16)).mean()
        assert exact > 0.97, f"only {exact:.3f} of the elements bit-equal"


@pytest.mark.parametrize("B,n_emb,W", SHAPES16)
def test_fp16_backward_equals_the_reference_fused_kernel(oracle, ref, B, n_emb, W):
    from hugectr_amd import _lib
    rng = np.random.default_rng(B * 1000 + n_emb * 10 + W + 1)
    mlp, emb = _inputs(rng, B, n_emb, W, np.float16)
    n_ins = n_emb + 1
    out_len = W + n_ins * (n_ins - 1) // 2 + 1
    top = (rng.standard_normal((B, out_len)) * 0.5).astype(np.float16)
    top[:, -1] = 0
    mg, eg = mlp.copy(), emb.copy()          # in place: inputs in, gradients out
    ug = top.copy()
    ref.refinter_bwd16(_p(ug), _p(mg), _p(eg), B, n_ins, W)
    wmg, weg = oracle.interaction_bwd(mlp.astype(np.float32), emb.astype(np.float32),
                                      top.astype(np.float32))
    assert np.isfinite(mg.astype(np.float32)).all() and np.isfinite(eg.astype(np.float32)).all()
    # (sums of up to n_ins products of two binary16 values, rounded to binary16 at the end)
    assert_close(mg.astype(np.float32), wmg, 4e-3, 2e-3, "reference kernel vs oracle: mlp grad")
    assert_close(eg.astype(np.float32), weg, 4e-3, 2e-3, "reference kernel vs oracle: emb grad")
    if emu.available():
        lib = emu.load_under_test()
        hmg, heg = np.full_like(mlp, np.nan), np.full_like(emb, np.nan)
        emu.check(lib, lib.hctr_interaction_bwd(B, n_emb, W, _p(mlp), _p(emb), _p(top), _p(hmg),
                                                _p(heg), _lib.F16, None))
        assert np.isfinite(hmg.astype(np.float32)).all() and np.isfinite(heg.astype(np.float32)).all()
        assert_close(hmg.astype(np.float32), mg.astype(np.float32), 4e-3, 2e-3, "hip vs reference: mlp")
        assert_close(heg.astype(np.float32), eg.astype(np.float32), 4e-3, 2e-3, "hip vs reference: emb")


@pytest.mark.parametrize("B,n_emb,W", [(4, 26, 128), (3, 5, 16), (2, 33, 8), (5, 2, 11)])
def test_fp32_generic_path_kernels_equal_the_oracle(oracle, ref, B, n_emb, W):
    """concat_kernel -> X X^T (cuBLAS in the reference; a plain fp32 product here) ->
    gather_concat_fprop_kernel: the layout the oracle restates, n_ins >= 32 included"""
    from hugectr_amd import _lib
    rng = np.random.default_rng(B + n_emb + W)
    mlp, emb = _inputs(rng, B, n_emb, W, np.float32)
    n_ins = n_emb + 1
    out_len = W + n_ins * (n_ins - 1) // 2 + 1
    got = np.full((B, out_len), np.nan, np.float32)
    ref.refinter_fwd32(_p(mlp), _p(emb), _p(got), B, n_ins, W)
    want = oracle.interaction_fwd(mlp, emb)
    assert np.isfinite(got).all()
    assert_close(got, want, 1e-6, 1e-6, "reference kernels vs oracle")
    assert np.array_equal(got[:, :W], mlp) and not got[:, -1].any()
    if emu.available():
        lib = emu.load_under_test()
        hip = np.full((B, out_len), np.nan, np.float32)
        emu.check(lib, lib.hctr_interaction_fwd(B, n_emb, W, _p(mlp), _p(emb), _p(hip), _lib.F32, None))
        # (the HIP fp32 kernel splits the inputs into three bf16 terms for the matrix cores:
        #  2^-16 relative per product, the reference layer test's own tolerance is 1e-3)
        assert np.isfinite(hip).all()
        assert_close(hip, got, 1e-3, 1e-4, "hip vs reference")
