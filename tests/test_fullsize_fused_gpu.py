"""The headline kernel at BASELINE.json's full size: hctr_emb_forward_interaction (gather fused into
the dot interaction, `interaction_fwd16_gather_kernel`) on the 187.8 M-row, 89.5 GiB Criteo-1TB table
at batch 65536 -- row offsets `r * W` with r up to 1.9e8, `b * n_emb` strides, the index prefetch
clamped at the last sample.  The oracle cannot run this size in seconds, so: pooled vectors ==
the table rows rounded to fp16 (bit-exact), output == the unfused interaction kernel on those pooled
vectors (bit-exact; that kernel is oracle-checked), a sample of the batch directly against the
oracle's interaction, an evaluation batch whose unseen keys pool to exact zeros."""
import numpy as np
import pytest

from test_fullsize_gpu import CRITEO_1TB, _batch

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def criteo16():
    import torch
    import hugectr_amd as ha
    from hugectr_amd import _lib
    torch.cuda.empty_cache()
    free, _ = torch.cuda.mem_get_info()
    if free < 130 * 2**30:
        pytest.skip("needs ~110 GiB of free HBM")
    B, S, D = 65536, len(CRITEO_1TB), 128
    emb = ha.SparseEmbeddingHash(_lib.EMB_LOCALIZED, B, B, sum(CRITEO_1TB), D, S, S, 0,
                                 ha.OptParams(optimizer=_lib.OPT_SGD, lr=0.5, atomic_update=False),
                                 slot_size_array=CRITEO_1TB, out_dtype=torch.float16)
    emb.init_params()
    yield emb, (B, S, D)
    del emb
    torch.cuda.empty_cache()


def test_fullsize_gather_fused_into_interaction(criteo16, oracle):
    import torch
    import hugectr_amd as ha
    emb, (B, S, D) = criteo16
    rng = np.random.default_rng(5)
    ro = torch.arange(0, B * S + 1, dtype=torch.int64, device="cuda")
    g = torch.Generator(device="cuda").manual_seed(3)
    for it in range(2):
        kb = torch.from_numpy(_batch(rng, B)).cuda()
        mlp = torch.randn(B, D, device="cuda", generator=g).half()
        emb.index(True, ro, kb)
        vi = emb.value_index(B * S).clone()
        assert int(vi.max()) < emb.get_vocabulary_size()
        m2 = mlp.clone().requires_grad_()
        got = {}
        out_f = ha.interaction_gather(m2, emb, True, on_emb_grad=lambda d: got.update(dE=d))
        pooled = out_f.grad_fn.saved_tensors[1]
        # one key per bucket: the pooled vector is the fp32 row rounded once to fp16
        want_pooled = emb.table()[vi].half().view(B, S, D)
        assert torch.equal(pooled.view(torch.int16), want_pooled.view(torch.int16))
        # the MFMA chain and the output stage: the unfused kernel on the same pooled vectors
        m1 = mlp.clone().requires_grad_()
        e1 = want_pooled.clone().requires_grad_()
        out_d = ha.interaction(m1, e1)
        assert torch.equal(out_f.view(torch.int16), out_d.view(torch.int16))
        # a sample of the batch (first, last and a stride in between) directly against the oracle
        sel = torch.cat([torch.arange(0, 64), torch.arange(B - 64, B),
                         torch.arange(64, B - 64, 997)]).cuda()
        zr = oracle.interaction_fwd(mlp[sel].float().cpu().numpy(),
                                    want_pooled[sel].float().cpu().numpy())
        assert np.allclose(out_f.detach()[sel].float().cpu().numpy(), zr, rtol=4e-3, atol=4e-2)
        top = torch.randn(out_d.shape, device="cuda", generator=g).half()
        out_d.backward(top)
        out_f.backward(top)
        assert torch.equal(m1.grad, m2.grad) and torch.equal(e1.grad, got["dE"])
    # evaluation: keys the table has never met pool to exact zeros, the others to their rows
    kb = torch.from_numpy(_batch(rng, B)).cuda()
    mlp = torch.randn(B, D, device="cuda", generator=g).half()
    rows_before = emb.get_vocabulary_size()
    emb.index(False, ro, kb)
    out_e = ha.interaction_gather(mlp, emb, False, on_emb_grad=None)
    assert emb.get_vocabulary_size() == rows_before, "an evaluation batch inserted keys"
    pooled_e = emb.forward(False, ro, kb)
    miss = (pooled_e.float().abs().sum(-1) == 0)
    assert bool(miss.any()) and not bool(miss.all())
    out_d = ha.interaction(mlp, pooled_e)
    assert torch.equal(out_e.view(torch.int16), out_d.view(torch.int16))
    emb.check_overflow()
