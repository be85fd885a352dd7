"""The C oracle (oracle/hctr_oracle.c) against the committed fixtures of tests/golden/ -- an
independent pure-numpy restatement of the same reference sources (see make_golden.py for the
provenance).  Runs on CPU."""
import os

import numpy as np
import pytest

from util import assert_close

G = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")


def load(name):
    return np.load(os.path.join(G, name))


@pytest.mark.parametrize("kb", [8, 4])
def test_hash_index_matches_golden(oracle, kb):
    z = load(f"hash_index_k{kb}.npz")
    assert (oracle.hash_keys(z["batch1"], kb) == z["hash1"]).all()
    ht = oracle.HashTable(int(z["capacity"]), kb)
    assert ht.table_size() == int(z["slots"])
    assert (ht.get_insert(z["batch1"]) == z["vi1"]).all()
    assert (ht.get_insert(z["batch2"]) == z["vi2"]).all()
    assert (ht.get_mark(z["eval"]) == z["vi_eval"]).all()
    assert ht.size() == int(z["size"])


def _opt(oracle, kind, lr, scaler, b1=0.9, b2=0.999, eps=1e-7, times=0):
    o = oracle.OptParamsC()
    o.optimizer, o.update_type, o.lr = kind, oracle.UPDATE_LOCAL, lr
    o.beta1, o.beta2, o.epsilon = b1, b2, eps
    o.momentum_factor, o.scaler, o.times = 0.0, scaler, times
    return o


@pytest.mark.parametrize("name", ["mean_multihot", "sum_onehot"])
def test_embedding_matches_golden(oracle, name):
    z = load(f"embedding_{name}.npz")
    D, comb = int(z["D"]), int(z["combiner"])
    ro, vi = z["row_offset"], z["value_index"]
    ht = oracle.HashTable(int(z["V"]), 8)
    assert (ht.get_insert(z["keys"]) == vi).all()
    out = oracle.forward(ro, vi, z["table"], D, comb)
    assert (out.view(np.uint32) == z["out"].view(np.uint32)).all(), "forward not bit-exact"
    wg = oracle.backward(ro, z["top_grad"], D, comb)
    assert (wg.view(np.uint32) == z["wgrad"].view(np.uint32)).all(), "wgrad not bit-exact"
    sc = float(z["scaler"])
    t = z["table"].copy()
    oracle.update_params(ro, vi, wg, _opt(oracle, oracle.OPT_SGD, float(z["sgd_lr"]), sc), t)
    assert_close(t, z["table_sgd"], 1e-6, 1e-7, "sgd")
    lr, b1, b2, eps, times = z["adam"]
    t, m, v = z["table"].copy(), z["m0"].copy(), z["v0"].copy()
    oracle.update_params(ro, vi, wg, _opt(oracle, oracle.OPT_ADAM, lr, sc, b1, b2, eps, int(times)),
                         t, m, v)
    assert_close(t, z["table_adam"], 1e-5, 1e-7, "adam table")
    assert_close(m, z["m1"], 1e-5, 1e-8, "adam m")
    assert_close(v, z["v1"], 1e-5, 1e-9, "adam v")
    lr, eps = z["adagrad"]
    t, a = z["table"].copy(), z["a0"].copy()
    oracle.update_params(ro, vi, wg, _opt(oracle, oracle.OPT_ADAGRAD, lr, sc, eps=eps), t, a)
    assert_close(t, z["table_adagrad"], 1e-5, 1e-7, "adagrad table")
    assert_close(a, z["a1"], 1e-5, 1e-8, "adagrad accum")


@pytest.mark.parametrize("name", ["dlrm", "small"])
def test_interaction_matches_golden(oracle, name):
    z = load(f"interaction_{name}.npz")
    out = oracle.interaction_fwd(z["mlp"], z["emb"])
    assert_close(out, z["out"], 1e-5, 1e-4, "interaction fwd")
    assert (out[:, -1] == 0).all()
    dm, de = oracle.interaction_bwd(z["mlp"], z["emb"], z["top_grad"])
    assert_close(dm, z["dmlp"], 1e-5, 1e-4, "interaction dmlp")
    assert_close(de, z["demb"], 1e-5, 1e-4, "interaction demb")


def test_cross_v1_matches_golden(oracle):
    z = load("cross_v1.npz")
    outputs, _ = oracle.cross_v1_fwd(z["x0"], z["kernels"], z["biases"])
    assert_close(outputs[-1], z["out"], 1e-5, 1e-5, "cross v1")
