"""The HIP path (through the C ABI) against the committed fixtures of tests/golden/ (provenance:
tests/golden/make_golden.py).  Index stage and forward bit-exact; optimizer results rel 1e-5."""
import os

import numpy as np
import pytest

from util import assert_close

pytestmark = pytest.mark.gpu
G = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")


def load(name):
    return np.load(os.path.join(G, name))


def _t(torch, a, dtype=None):
    t = torch.from_numpy(np.ascontiguousarray(a)).to("cuda")
    return t if dtype is None else t.to(dtype)


@pytest.mark.parametrize("kb", [8, 4])
def test_hash_index_golden(kb):
    import torch
    from hugectr_amd import _lib
    from test_hash_gpu import GpuHT
    z = load(f"hash_index_k{kb}.npz")
    ktype = _lib.KEY_I64 if kb == 8 else _lib.KEY_U32

    def dev(keys):
        if kb == 8:
            return _t(torch, keys)
        return _t(torch, keys.astype(np.uint32).view(np.int32))

    ht = GpuHT(int(z["capacity"]), ktype)
    assert ht.table_size() == int(z["slots"])
    k1 = dev(z["batch1"])
    h = torch.empty(k1.numel(), dtype=torch.int32, device="cuda")
    _lib.check(_lib.lib.hctr_hash_keys(_lib.ptr(k1), ktype, k1.numel(), _lib.ptr(h),
                                       _lib.stream_ptr()))
    assert (h.cpu().numpy().view(np.uint32) == z["hash1"]).all()
    assert (ht.get_insert(k1) == z["vi1"]).all()
    assert (ht.get_insert(dev(z["batch2"])) == z["vi2"]).all()
    assert (ht.get_mark(dev(z["eval"])) == z["vi_eval"]).all()
    assert ht.size() == int(z["size"])


@pytest.mark.parametrize("name", ["mean_multihot", "sum_onehot"])
@pytest.mark.parametrize("optimizer", ["sgd", "adam", "adagrad"])
def test_embedding_golden(name, optimizer):
    import torch
    import hugectr_amd as ha
    from hugectr_amd import _lib
    z = load(f"embedding_{name}.npz")
    B, S, D, comb, V = (int(z[k]) for k in ("B", "S", "D", "combiner", "V"))
    sc = float(z["scaler"])
    if optimizer == "sgd":
        opt = ha.OptParams(optimizer=_lib.OPT_SGD, lr=float(z["sgd_lr"]), scaler=sc,
                           atomic_update=False)
    elif optimizer == "adam":
        lr, b1, b2, eps, times = z["adam"]
        opt = ha.OptParams(optimizer=_lib.OPT_ADAM, lr=lr, beta1=b1, beta2=b2, epsilon=eps,
                           scaler=sc)
    else:
        lr, eps = z["adagrad"]
        opt = ha.OptParams(optimizer=_lib.OPT_ADAGRAD, lr=lr, epsilon=eps, scaler=sc)
    emb = ha.SparseEmbeddingHash(_lib.EMB_LOCALIZED, B, 0, V, D, S * 4, S, comb, opt)
    emb.table().copy_(_t(torch, z["table"]))
    if optimizer == "adam":
        emb.opt_state(0).copy_(_t(torch, z["m0"]))
        emb.opt_state(1).copy_(_t(torch, z["v0"]))
        assert int(z["adam"][4]) == 1  # update_params increments the step counter 0 -> 1
    elif optimizer == "adagrad":
        emb.opt_state(0).copy_(_t(torch, z["a0"]))
    out = emb.forward(True, _t(torch, z["row_offset"]), _t(torch, z["keys"]))
    got_vi = emb.value_index(z["keys"].size).cpu().numpy().view(np.uint64)
    assert (got_vi == z["value_index"]).all()
    got = out.cpu().numpy().reshape(-1, D)
    assert (got.view(np.uint32) == z["out"].view(np.uint32)).all(), "forward not bit-exact"
    emb.backward(_t(torch, z["top_grad"]).view(B, S, D))
    wg = emb.get_wgrad().cpu().numpy().reshape(-1, D)
    assert (wg.view(np.uint32) == z["wgrad"].view(np.uint32)).all(), "wgrad not bit-exact"
    emb.update_params()
    t = emb.table().cpu().numpy()
    if optimizer == "sgd":
        assert_close(t, z["table_sgd"], 1e-5, 1e-7, "sgd")
    elif optimizer == "adam":
        assert_close(t, z["table_adam"], 1e-5, 1e-7, "adam table")
        assert_close(emb.opt_state(0).cpu().numpy(), z["m1"], 1e-5, 1e-8, "adam m")
        assert_close(emb.opt_state(1).cpu().numpy(), z["v1"], 1e-5, 1e-9, "adam v")
    else:
        assert_close(t, z["table_adagrad"], 1e-5, 1e-7, "adagrad table")
        assert_close(emb.opt_state(0).cpu().numpy(), z["a1"], 1e-5, 1e-8, "adagrad accum")


@pytest.mark.parametrize("name", ["dlrm", "small"])
def test_interaction_golden(name):
    import torch
    import hugectr_amd as ha
    z = load(f"interaction_{name}.npz")
    mlp = _t(torch, z["mlp"]).requires_grad_()
    emb = _t(torch, z["emb"]).requires_grad_()
    out = ha.interaction(mlp, emb)
    assert_close(out.detach().cpu().numpy(), z["out"], 2e-4, 1e-3, "interaction fwd")
    out.backward(_t(torch, z["top_grad"]))
    assert_close(mlp.grad.cpu().numpy(), z["dmlp"], 2e-4, 2e-3, "interaction dmlp")
    assert_close(emb.grad.cpu().numpy(), z["demb"], 2e-4, 2e-3, "interaction demb")


def test_cross_v1_golden():
    import torch
    import hugectr_amd as ha
    z = load("cross_v1.npz")
    L, w = z["kernels"].shape
    layer = ha.MultiCrossLayer(w, L).cuda()
    with torch.no_grad():
        layer.kernels.copy_(_t(torch, z["kernels"]))
        layer.biases.copy_(_t(torch, z["biases"]))
    out = layer(_t(torch, z["x0"]))
    assert_close(out.detach().cpu().numpy(), z["out"], 1e-4, 1e-4, "cross v1")
