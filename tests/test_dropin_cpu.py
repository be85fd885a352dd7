"""Drop-in boundary, CPU side: `import hugectr` / `import sparse_operation_kit` are the module names
of the reference (R/HugeCTR/src/pybind/module_main.cpp:36-48), and every call the reference's own
UNMODIFIED scripts make -- recorded in tests/golden/script_traces.json by running those scripts
against a recording stand-in -- is accepted with its exact arguments.  (The same calls are executed
for real on the GPU in tests/test_dropin_gpu.py.)"""
import json
import os
import sys

import pytest

from dropin_replay import load_traces, replay

HERE = os.path.dirname(os.path.abspath(__file__))
REF = "/root/reference"


def test_module_names_of_the_reference_import():
    import hugectr
    import sparse_operation_kit as sok
    from hugectr.tools import DataGenerator, DataGeneratorParams  # noqa: F401  (R/README.md:62)
    from hugectr_amd import hugectr as impl
    assert hugectr.Model is impl.Model and hugectr.CreateSolver is impl.CreateSolver
    assert hugectr.Layer_t.Interaction is impl.Layer_t.Interaction
    for name in ("init", "Variable", "DynamicVariable", "lookup_sparse", "OptimizerWrapper"):
        assert hasattr(sok, name), name


@pytest.mark.skipif(not os.path.isdir(REF), reason="needs the reference checkout")
def test_committed_traces_are_what_the_reference_scripts_do_today():
    sys.path.insert(0, os.path.join(HERE, "golden"))
    import make_script_traces as mk
    fresh = json.loads(json.dumps(mk.record_all(), sort_keys=True))
    assert fresh == load_traces(), "re-run tests/golden/make_script_traces.py"


class _ModelStandIn:
    """takes the place of hugectr.Model (which needs a GPU): checks what the scripts hand it"""

    def __init__(self, solver, reader, optimizer):
        from hugectr_amd import hugectr as impl
        assert isinstance(solver, impl.Solver) and isinstance(reader, impl.DataReaderParams)
        assert isinstance(optimizer, impl.OptParamsPy)
        self.items = []

    def add(self, item):
        from hugectr_amd import hugectr as impl
        assert isinstance(item, (impl.Input, impl.SparseEmbedding, impl.DenseLayer,
                                 impl.EmbeddingCollectionConfig)), type(item)
        self.items.append(item)

    def __getattr__(self, name):  # compile / summary / fit / graph_to_json ...: exist on Model?
        from hugectr_amd import hugectr as impl
        if not hasattr(impl.Model, name):
            raise AttributeError(f"hugectr.Model has no method {name}")
        return lambda *a, **k: None


@pytest.mark.parametrize("script", sorted(load_traces()))
def test_every_call_of_the_reference_script_is_accepted(script, monkeypatch):
    import hugectr
    monkeypatch.setenv("WORLD_SIZE", "1")  # (no relaunch: CreateSolver may name several GPUs)
    tr = load_traces()[script]
    res = replay(hugectr, tr["calls"],
                 substitute=lambda t: _ModelStandIn if t == "Model" else None)
    models = [r for r in res if isinstance(r, _ModelStandIn)]
    assert len(models) == 1 and len(models[0].items) >= 4
