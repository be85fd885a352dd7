"""Size-independent properties at BASELINE.json's full DLRM Criteo-1TB size (26 tables,
187.8 M rows x 128 fp32 = 89.5 GiB, batch 65536, power-law keys): the oracle cannot run this in
seconds, so the checks are properties the domain offers -- one-hot sum == the gathered row itself
(bit-exact), a key <-> row bijection, idempotence, a zero-gradient update is the identity, and the
SGD update equals an fp64 index_add over the batch's unique rows."""
import numpy as np
import pytest

pytestmark = pytest.mark.gpu

# R/test/embedding_collection_test/dgx_a100_one_hot.py:24-51 (Criteo-1TB slot sizes)
CRITEO_1TB = [39884406, 39043, 17289, 7420, 20263, 3, 7120, 1543, 63, 38532951, 2953546, 403346,
              10, 2208, 11938, 155, 4, 976, 14, 39979771, 25641295, 39664984, 585935, 12972, 108,
              36]


def _powerlaw(rng, n, vocab, alpha):
    u = rng.random(n, dtype=np.float32).astype(np.float64)
    a = 1.0 - alpha
    y = ((float(vocab) ** a - 1.0) * u + 1.0) ** (1.0 / a)
    return np.clip(np.round(y) - 1, 0, vocab - 1).astype(np.int64)


def _batch(rng, B):
    offs = np.concatenate([[0], np.cumsum(CRITEO_1TB)[:-1]]).astype(np.int64)
    keys = np.empty((B, len(CRITEO_1TB)), dtype=np.int64)
    for s, v in enumerate(CRITEO_1TB):
        keys[:, s] = _powerlaw(rng, B, v, 1.1) + offs[s]
    return keys.reshape(-1)


@pytest.fixture(scope="module")
def criteo():
    import torch
    import hugectr_amd as ha
    from hugectr_amd import _lib
    free, _ = torch.cuda.mem_get_info()
    if free < 130 * 2**30:
        pytest.skip("needs ~110 GiB of free HBM")
    B, S, D = 65536, len(CRITEO_1TB), 128
    lr = 0.5
    emb = ha.SparseEmbeddingHash(_lib.EMB_LOCALIZED, B, 0, sum(CRITEO_1TB), D, S, S, 0,
                                 ha.OptParams(optimizer=_lib.OPT_SGD, lr=lr, atomic_update=False),
                                 slot_size_array=CRITEO_1TB)
    emb.init_params()
    rng = np.random.default_rng(77)
    keys = [torch.from_numpy(_batch(rng, B)).cuda() for _ in range(2)]
    ro = torch.arange(0, B * S + 1, dtype=torch.int64, device="cuda")
    yield emb, ro, keys, (B, S, D, lr)
    del emb
    torch.cuda.empty_cache()


def test_fullsize_forward_is_the_gathered_row_and_index_is_a_bijection(criteo):
    import torch
    emb, ro, keys, (B, S, D, _) = criteo
    seen = 0
    for kb in keys:
        out = emb.forward(True, ro, kb)
        vi = emb.value_index(B * S).clone()
        assert int(vi.max()) < emb.get_vocabulary_size() <= sum(CRITEO_1TB)
        # one key per bucket, sum combiner: the pooled vector IS the table row, bit for bit
        rows = emb.table()[vi]
        assert torch.equal(out.view(B * S, D), rows)
        # equal keys <-> equal rows (torch.unique as an independent grouping)
        uk, inv_k = torch.unique(kb, return_inverse=True)
        uv, inv_v = torch.unique(vi, return_inverse=True)
        assert uk.numel() == uv.numel()
        first_v = torch.zeros(uk.numel(), dtype=vi.dtype, device="cuda")
        first_v[inv_k] = vi
        assert torch.equal(first_v[inv_k], vi), "one key resolved to two different rows"
        seen = max(seen, emb.get_vocabulary_size())
        # idempotence: the same batch again inserts nothing and returns identical bits
        out2 = emb.forward(True, ro, kb)
        assert torch.equal(out2, out) and torch.equal(emb.value_index(B * S), vi)
        assert emb.get_vocabulary_size() == seen
    emb.check_overflow()


def test_fullsize_zero_gradient_update_is_identity_and_sgd_matches_fp64(criteo):
    import torch
    emb, ro, keys, (B, S, D, lr) = criteo
    kb = keys[0]
    emb.forward(True, ro, kb)
    vi = emb.value_index(B * S).clone()
    urows, inv = torch.unique(vi, return_inverse=True)
    before = emb.table()[urows].clone()
    emb.backward(torch.zeros((B, S, D), device="cuda"))
    emb.update_params()
    assert torch.equal(emb.table()[urows], before), "zero gradient changed the table"
    g = torch.randn((B, S, D), device="cuda")
    emb.forward(True, ro, kb)
    emb.backward(g)
    emb.update_params()
    want = torch.zeros((urows.numel(), D), dtype=torch.float64, device="cuda")
    want.index_add_(0, inv, g.view(B * S, D).double())
    want = before.double() - lr * want
    got = emb.table()[urows].double()
    err = (got - want).abs().max().item()
    scale = want.abs().max().item()
    assert err <= 1e-5 * scale, (err, scale)  # the hottest rows add ~20 k gradients in fp32


def test_fullsize_multihot_pooling_is_linear_in_the_rows(criteo):
    """ragged multi-hot CSR at full batch on the real table (flat key-walk kernel): every pooled
    vector equals the fp64 sum of its rows (sum) / their mean, checked through an independent
    torch segment reduction; empty buckets pool to exactly zero"""
    import torch
    import hugectr_amd as ha
    from hugectr_amd import _lib
    emb, _, keys, (B, S, D, _) = criteo
    table = emb.table()
    g = torch.Generator(device="cuda").manual_seed(11)
    nb = B * S
    lens = torch.randint(0, 6, (nb,), device="cuda", generator=g)
    lens[torch.rand(nb, device="cuda", generator=g) < 0.2] = 0
    ro = torch.zeros(nb + 1, dtype=torch.int64, device="cuda")
    torch.cumsum(lens, 0, out=ro[1:])
    nnz = int(ro[-1])
    rows = torch.randint(0, emb.get_vocabulary_size(), (nnz,), device="cuda", generator=g)
    seg = torch.repeat_interleave(torch.arange(nb, device="cuda"), lens)
    for comb in (0, 1):
        out = torch.empty((nb, D), device="cuda")
        _lib.check(_lib.lib.hctr_forward_pool_multihot(nb, D, comb, _lib.ptr(ro), _lib.KEY_I64,
                                                       _lib.ptr(rows), _lib.ptr(table),
                                                       _lib.ptr(out), _lib.F32, _lib.stream_ptr()))
        # reference on a sample of buckets (an fp64 index_add over all 4 M rows x 128 is 4 GB: fine)
        want = torch.zeros((nb, D), dtype=torch.float64, device="cuda")
        want.index_add_(0, seg, table[rows].double())
        if comb == 1:
            want = want / lens.clamp_min(1).unsqueeze(1)
        err = (out.double() - want).abs().max().item()
        assert err <= 1e-6 * max(1.0, want.abs().max().item()), (comb, err)
        assert float(out[lens == 0].abs().max()) == 0.0


def test_fullsize_interaction_matches_fp32_bmm():
    import torch
    import hugectr_amd as ha
    B, n, W = 65536, 26, 128
    g = torch.Generator(device="cuda").manual_seed(5)
    mlp = torch.randn((B, W), device="cuda", generator=g)
    emb = torch.randn((B, n, W), device="cuda", generator=g)
    out = ha.interaction(mlp, emb)
    x = torch.cat([mlp[:, None, :], emb], dim=1)
    m = torch.bmm(x, x.transpose(1, 2))
    li, lj = torch.tril_indices(n + 1, n + 1, offset=-1, device="cuda")
    assert torch.equal(out[:, :W], mlp)
    assert torch.allclose(out[:, W:-1], m[:, li, lj], rtol=2e-4, atol=1e-3)
    assert (out[:, -1] == 0).all()
