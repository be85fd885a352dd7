"""Drop-in check of the `hugectr` Python surface against the reference's OWN scripts: every
`hugectr.<name>` the sample / test scripts under R/samples, R/test/embedding_collection_test and
R/test/pybind_test touch must exist in hugectr_amd.hugectr, and every keyword they pass to
`hugectr.<callable>(...)` must be accepted.  Layer types of model families outside the hot-path scope
(attention / sequence models: DIN, BST, MMoE) are listed explicitly.  Runs where the reference
checkout is mounted (the build container); skipped elsewhere."""
import ast
import collections
import dataclasses
import glob
import inspect
import os

import pytest

REF = "/root/reference"
# dense layers of model families outside SURVEY 8 (DIN / BST / MMoE attention and sequence blocks)
OUT_OF_SCOPE_LAYERS = {"FusedReshapeConcat", "LayerNorm", "MatrixMultiply", "MultiHeadAttention",
                       "PReLU_Dice", "ReduceMean", "Scale", "SequenceMask"}


def _chain(node):
    parts = []
    while isinstance(node, ast.Attribute):
        parts.append(node.attr)
        node = node.value
    if isinstance(node, ast.Name):
        parts.append(node.id)
        return list(reversed(parts))
    return None


def _usage():
    files = sorted(glob.glob(REF + "/samples/**/*.py", recursive=True) +
                   glob.glob(REF + "/test/embedding_collection_test/*.py") +
                   glob.glob(REF + "/test/pybind_test/*.py"))
    names, calls, n_files = collections.Counter(), collections.defaultdict(set), 0
    for f in files:
        try:
            tree = ast.parse(open(f).read())
        except SyntaxError:
            continue
        hit = False
        for node in ast.walk(tree):
            if isinstance(node, ast.Attribute):
                c = _chain(node)
                if c and c[0] == "hugectr" and len(c) >= 2:
                    names[".".join(c[1:])] += 1
                    hit = True
            if isinstance(node, ast.Call):
                c = _chain(node.func)
                if c and c[0] == "hugectr" and len(c) >= 2:
                    calls[".".join(c[1:])].update(k.arg for k in node.keywords if k.arg)
        n_files += hit
    return names, calls, n_files


@pytest.mark.skipif(not os.path.isdir(REF + "/samples"), reason="reference checkout not mounted")
def test_every_name_the_reference_scripts_use_exists():
    import hugectr_amd.hugectr as H
    names, calls, n_files = _usage()
    assert n_files >= 20 and len(names) >= 60
    missing = []
    for n in names:
        obj = H
        for part in n.split("."):
            if not hasattr(obj, part):
                missing.append(n)
                break
            obj = getattr(obj, part)
    assert not missing, f"hugectr names used by the reference's scripts but absent here: {missing}"
    # layer types: supported by Model, or on the explicit out-of-scope list
    src = inspect.getsource(H.Model._build_layer)
    used_layers = {n.split(".")[1] for n in names if n.startswith("Layer_t.")}
    unsupported = {l for l in used_layers if f"Layer_t.{l}" not in src}
    assert unsupported == OUT_OF_SCOPE_LAYERS & used_layers, unsupported ^ (OUT_OF_SCOPE_LAYERS & used_layers)
    # keyword arguments of the callables
    bad = {}
    for name, kws in calls.items():
        obj = H
        for part in name.split("."):
            obj = getattr(obj, part)
        if name == "DenseLayer":
            accepted = set(H.DenseLayer(H.Layer_t.ReLU, ["a"], ["b"]).__dict__)
        elif name == "CreateSolver":
            accepted = {f.name for f in dataclasses.fields(H.Solver)}
        else:
            params = inspect.signature(obj).parameters
            if any(p.kind == p.VAR_KEYWORD for p in params.values()):
                continue
            accepted = set(params)
        extra = kws - accepted
        if extra:
            bad[name] = sorted(extra)
    assert not bad, f"keyword arguments the reference's scripts pass that are not accepted: {bad}"


def test_training_callback_hooks_are_called_in_the_reference_order():
    """Model.fit drives TrainingCallback as R/HugeCTR/src/pybind/model.cpp:869-994 does (checked on
    the host logic only: no device work)"""
    import hugectr_amd.hugectr as H
    assert {m for m in ("on_training_start", "on_training_end", "on_eval_start", "on_eval_end")
            if callable(getattr(H.TrainingCallback, m))} == {"on_training_start", "on_training_end",
                                                             "on_eval_start", "on_eval_end"}
    assert H.AllReduceAlgo.NCCL.name == "NCCL" and H.AllReduceAlgo.OneShot.name == "OneShot"
    s = H.CreateSolver(batchsize=8, vvgpu=[[0]], training_callbacks=[H.TrainingCallback()],
                       all_reduce_algo=H.AllReduceAlgo.NCCL)
    assert len(s.training_callbacks) == 1
