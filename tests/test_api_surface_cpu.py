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
                       "PReLU_Dice", "ReduceMean", "SequenceMask"}


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


@pytest.mark.skipif(not os.path.isdir(REF + "/samples"), reason="reference checkout not mounted")
def test_every_model_method_the_reference_scripts_call_exists():
    """`model.<verb>(...)` / `ebc_config.<verb>(...)` calls of the same scripts: every verb exists and
    accepts the keywords used"""
    import hugectr_amd.hugectr as H
    files = sorted(glob.glob(REF + "/samples/**/*.py", recursive=True) +
                   glob.glob(REF + "/test/embedding_collection_test/*.py") +
                   glob.glob(REF + "/test/pybind_test/*.py"))
    owner = {"model": H.Model, "ebc_config": H.EmbeddingCollectionConfig}
    used, kws = collections.Counter(), collections.defaultdict(set)
    for f in files:
        try:
            tree = ast.parse(open(f).read())
        except SyntaxError:
            continue
        for node in ast.walk(tree):
            if (isinstance(node, ast.Call) and isinstance(node.func, ast.Attribute) and
                    isinstance(node.func.value, ast.Name) and node.func.value.id in owner):
                k = (node.func.value.id, node.func.attr)
                used[k] += 1
                kws[k].update(a.arg for a in node.keywords if a.arg)
    assert used[("model", "add")] > 500 and used[("model", "fit")] >= 20
    missing, bad = [], {}
    for (obj, verb) in used:
        fn = getattr(owner[obj], verb, None)
        if fn is None:
            missing.append(f"{obj}.{verb}")
            continue
        params = inspect.signature(fn).parameters
        if not any(p.kind == p.VAR_KEYWORD for p in params.values()):
            extra = kws[(obj, verb)] - set(params)
            if extra:
                bad[f"{obj}.{verb}"] = sorted(extra)
    assert missing == [], missing
    assert not bad, bad


def test_learning_rate_scheduler_known_answers():
    """LearningRateScheduler::get_next (R/HugeCTR/include/learning_rate_scheduler.hpp:66-88)"""
    import hugectr_amd.hugectr as H
    s = H.LearningRateScheduler(1.0, warmup_steps=4, decay_start=6, decay_steps=4, decay_power=2.0,
                                end_lr=0.1)
    got = [round(s.get_next(), 6) for _ in range(12)]
    # warm-up 1/4 .. 4/4, flat to step 6, ((10 - step) / 4)^2 down to end_lr, then end_lr
    assert got == [0.25, 0.5, 0.75, 1.0, 1.0, 1.0, 0.5625, 0.25, 0.1, 0.1, 0.1, 0.1]
    flat = H.LearningRateScheduler(0.3)
    assert [flat.get_next() for _ in range(3)] == [0.3, 0.3, 0.3] and flat.get_step() == 3
    with pytest.raises(RuntimeError):
        H.LearningRateScheduler(0.1, decay_power=0.5)
