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


@pytest.mark.parametrize("opt_name", ["sgd", "adagrad"])
def test_fullsize_fp16_gradients_with_the_loss_scaler_match_fp64(opt_name):
    """The update the bench times -- fp16 top gradients, loss scaler 1024, the hot-row / cold-count
    split of a one-key-per-bucket batch -- at BASELINE configs[2]'s full size, against an fp64
    index_add_ over the batch's distinct rows (only fp32 SGD was checked at this
    size).  SGD: w - lr * sum(g) / scaler.  AdaGrad (a stateful optimizer through the same split;
    fp16 embeddings keep fp16 state): from zero state one step is the closed form
    w - lr * gi / (|gi| + eps), state = fp16(gi^2) (sparse_optimizer.cu:413-439).  Every distinct
    row of the batch is compared (227 k), the hottest of which add ~ 20 k gradients; untouched rows
    keep their bits."""
    import torch
    import hugectr_amd as ha
    from hugectr_amd import _lib
    free, _ = torch.cuda.mem_get_info()
    need = (100 if opt_name == "sgd" else 150) * 2**30
    if free < need:
        pytest.skip(f"needs ~{need >> 30} GiB of free HBM")
    B, S, D = 65536, len(CRITEO_1TB), 128
    lr, scaler, eps = 0.5, 1024.0, 1e-6
    opt = ha.OptParams(optimizer=_lib.OPT_SGD if opt_name == "sgd" else _lib.OPT_ADAGRAD, lr=lr,
                       atomic_update=False, scaler=scaler, epsilon=eps, initial_accu_value=0.0)
    emb = ha.SparseEmbeddingHash(_lib.EMB_LOCALIZED, B, 0, sum(CRITEO_1TB), D, S, S, 0, opt,
                                 slot_size_array=CRITEO_1TB, out_dtype=torch.float16)
    try:
        emb.init_params()
        rng = np.random.default_rng(78)
        ro = torch.arange(0, B * S + 1, dtype=torch.int64, device="cuda")
        kb = torch.from_numpy(_batch(rng, B)).cuda()
        emb.forward(True, ro, kb)
        vi = emb.value_index(B * S).clone()
        urows, inv = torch.unique(vi, return_inverse=True)
        before = emb.table()[urows].clone()
        gen = torch.Generator(device="cuda").manual_seed(79)
        # (scaled as a loss-scaled gradient of a 65536-sample mean is: small numbers times 1024)
        g = (torch.randn((B, S, D), device="cuda", generator=gen) * 0.02 * scaler / 64).half()
        # an untouched neighbour of every touched row (when it is not touched itself)
        other = (urows + 1).clamp_max(emb.get_vocabulary_size() - 1)
        other = other[~torch.isin(other, urows)]
        other_before = emb.table()[other].clone()
        emb.forward(True, ro, kb)
        emb.backward(g)
        emb.update_params()
        torch.cuda.synchronize()
        gi = torch.zeros((urows.numel(), D), dtype=torch.float64, device="cuda")
        gi.index_add_(0, inv, g.view(B * S, D).double())
        gi /= scaler
        got = emb.table()[urows].double()
        n_r = torch.bincount(inv, minlength=urows.numel()).double().unsqueeze(1)
        # fp32 sums of n fp16 addends, any fixed association: |err| <= n eps32 sum|g| / scaler per
        # element; the bound below uses the row's own n and its gradients' size
        gabs = torch.zeros_like(gi)
        gabs.index_add_(0, inv, g.view(B * S, D).double().abs())
        sum_err = n_r * 2.0 ** -24 * gabs / scaler
        if opt_name == "sgd":
            want = before.double() - lr * gi
            bound = lr * sum_err + 2 * torch.abs(want) * 2.0 ** -24 + 1e-12
        else:
            want = before.double() - lr * gi / (gi.abs() + eps)
            # d/dgi of gi / (|gi| + eps) = eps / (|gi| + eps)^2 <= 1 / (|gi| + eps)
            bound = lr * sum_err / (gi.abs() + eps) + 4 * torch.abs(want) * 2.0 ** -24 + 1e-7
            st = emb.opt_state(0)[urows]
            assert st.dtype == torch.float16
            want_st = (gi * gi)
            st_err = (st.double() - want_st).abs()
            assert bool((st_err <= want_st * 2.0 ** -10 + 2 * gi.abs() * sum_err + 1e-7).all()), \
                float(st_err.max())
        err = (got - want).abs()
        bad = err > bound
        assert not bool(bad.any()), (int(bad.sum()), float((err / bound).max()), float(err.max()))
        assert float((got - before.double()).abs().max()) > 1e-4, "the table did not move"
        assert torch.equal(emb.table()[other], other_before), "an untouched row changed"
    finally:
        del emb
        torch.cuda.empty_cache()


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
