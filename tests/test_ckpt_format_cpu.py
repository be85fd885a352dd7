"""Interchange formats (graph JSON, dense model file, sparse model directory) written by
hugectr_amd on an MI355X (fixture tests/golden/ckpt, made by tests/golden/make_ckpt_fixture.py):
* read back by the REFERENCE's own loader (R/onnx_converter/hugectr2onnx/hugectr_loader.py) and
  compared value by value -- runs wherever the reference checkout is mounted (the build container);
* and by an independent restatement of the documented layout, everywhere."""
import json
import os
import sys

import numpy as np
import pytest

HERE = os.path.dirname(os.path.abspath(__file__))
CKPT = os.path.join(HERE, "golden", "ckpt")
sys.path.insert(0, os.path.join(HERE, "golden"))


@pytest.mark.parametrize("name", ["dcn", "dlrm"])
def test_reference_loader_reads_our_checkpoint(name):
    import check_ckpt_with_reference as chk
    if not os.path.exists(chk.REF_LOADER):
        pytest.skip("reference checkout not mounted")
    seen, dims = chk.check(name)
    assert "embedding" in seen and len(seen) >= 7


@pytest.mark.parametrize("name", ["dcn", "dlrm"])
def test_documented_layout(name):
    """dense file = per layer, in graph order, weight [in, out] then bias (InnerProduct, each MLP
    sub-layer), w then b per cross layer; sparse directory = key int64[], emb_vector fp32[n][vec]
    (+ slot_id u64[] for localized embeddings)"""
    g = json.load(open(os.path.join(CKPT, f"{name}.json")))["layers"]
    truth = np.load(os.path.join(CKPT, f"{name}_truth.npz"))
    dense = np.fromfile(os.path.join(CKPT, f"{name}_dense_5.model"), dtype="<f4")
    assert g[0]["type"] == "Data" and isinstance(g[0]["sparse"][0]["nnz_per_slot"], list)
    width = {g[0]["dense"]["top"]: g[0]["dense"]["dense_dim"]}
    off = 0

    def take(n):
        nonlocal off
        v = dense[off:off + n]
        off += n
        return v

    for L in g[1:]:
        t = L["type"]
        if t.endswith("SparseEmbeddingHash"):
            hp = L["sparse_embedding_hparam"]
            vec = hp["embedding_vec_size"]
            slots = g[0]["sparse"][0]["slot_num"]
            width[L["top"]] = (slots, vec)
            d = os.path.join(CKPT, f"{name}0_sparse_5.model")
            keys = np.fromfile(os.path.join(d, "key"), dtype="<i8")
            vecs = np.fromfile(os.path.join(d, "emb_vector"), dtype="<f4").reshape(-1, vec)
            assert keys.size == vecs.shape[0] == np.unique(keys).size
            assert hp["max_vocabulary_size_global"] > keys.max()
            order = np.argsort(keys)
            torder = np.argsort(truth["emb_keys"])
            assert (keys[order] == truth["emb_keys"][torder]).all()
            assert (vecs[order] == truth["emb_vectors"][torder]).all()
            if t.startswith("Localized"):
                slot = np.fromfile(os.path.join(d, "slot_id"), dtype="<u8")
                assert slot.size == keys.size and slot.max() < slots
                assert L["optimizer"]["type"] == "SGD"
        elif t == "InnerProduct":
            i, o = width[L["bottom"]], L["fc_param"]["num_output"]
            assert (take(i * o).reshape(i, o) == truth[L["top"] + "_weight"]).all()
            assert (take(o) == truth[L["top"] + "_bias"].reshape(-1)).all()
            width[L["top"]] = o
        elif t == "MLP":
            i = width[L["bottom"]]
            for j, o in enumerate(L["mlp_param"]["num_outputs"]):
                assert (take(i * o).reshape(i, o) == truth[f"{L['top']}{j}_weight"]).all()
                assert (take(o) == truth[f"{L['top']}{j}_bias"].reshape(-1)).all()
                i = o
            width[L["top"]] = i
        elif t == "MultiCross":
            w = width[L["bottom"]]
            for l in range(L["mc_param"]["num_layers"]):
                assert (take(w) == truth[L["top"] + "_weights"][l]).all()
                assert (take(w) == truth[L["top"] + "_biases"][l]).all()
            width[L["top"]] = w
        elif t == "Concat":
            width[L["top"]] = sum(int(np.prod(width[b])) for b in L["bottom"])
        elif t == "Reshape":
            width[L["top"]] = L["leading_dim"]
        elif t == "Interaction":
            n, v = width[L["bottom"][1]]
            width[L["top"]] = v + (n + 1) * n // 2 + 1
        elif t in ("ReLU", "Dropout"):
            assert isinstance(L["bottom"], str)  # single names are strings in the reference schema
            width[L["top"]] = width[L["bottom"]]
    assert off == dense.size
