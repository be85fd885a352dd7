"""Loads the checkpoint fixture (tests/golden/ckpt, written by make_ckpt_fixture.py on an MI355X
through hugectr_amd) with the REFERENCE's own reader of these formats,
R/onnx_converter/hugectr2onnx/hugectr_loader.py (plain Python + numpy), and compares every tensor it
returns with truth.npz.  Run in the build container (the reference checkout is absent on the GPU
box); tests/test_ckpt_format_cpu.py runs the same check whenever /root/reference is present."""
import importlib.util
import os
import sys

import numpy as np

REF_LOADER = "/root/reference/onnx_converter/hugectr2onnx/hugectr_loader.py"
HERE = os.path.dirname(os.path.abspath(__file__))


def reference_loader():
    spec = importlib.util.spec_from_file_location("hugectr_loader_ref", REF_LOADER)
    mod = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(mod)
    return mod.HugeCTRLoader


def check(name, ckpt=os.path.join(HERE, "ckpt")):
    Loader = reference_loader()
    truth = np.load(os.path.join(ckpt, f"{name}_truth.npz"))
    ld = Loader(os.path.join(ckpt, f"{name}.json"), os.path.join(ckpt, f"{name}_dense_5.model"), True,
                [os.path.join(ckpt, f"{name}0_sparse_5.model")], None)
    seen = set()
    dims = {}
    for _ in range(ld.layers):
        params, weights, dims = ld.load_layer()
        for k, v in weights.items():
            if k == "key_to_indice_hash_all_tables":
                continue
            if k == "embedding_table":
                # row `indice` of the loader's table = vector of the key that maps to it
                hash_table = weights["hash_table"]
                keys, vecs = truth["emb_keys"], truth["emb_vectors"]
                idx = hash_table[keys]
                assert (idx > 0).all() and np.unique(idx).size == keys.size
                assert (v[idx] == vecs).all(), "embedding vectors"
                assert (v[0] == 0).all()
                seen.add("embedding")
            elif k == "hash_table":
                continue
            elif k.endswith("_weights") or k.endswith("_biases"):  # MultiCross: list per layer
                got = np.stack([np.asarray(x).reshape(-1) for x in v])
                assert (got == truth[k]).all(), k
                seen.add(k)
            else:
                assert v.shape == truth[k].shape, (k, v.shape, truth[k].shape)
                assert (v == truth[k]).all(), k
                seen.add(k)
    want = {k for k in truth.files if not k.startswith("emb_")} | {"embedding"}
    assert seen == want, (sorted(seen), sorted(want))
    # the loader consumed the dense file to its last byte
    assert ld._HugeCTRLoader__offset == os.path.getsize(os.path.join(ckpt, f"{name}_dense_5.model"))
    return sorted(seen), dims


if __name__ == "__main__":
    for n in ("dcn", "dlrm"):
        seen, dims = check(n)
        print(n, "ok:", seen)
        print("   tensor dims per the reference loader:", dims)
