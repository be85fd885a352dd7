"""Records what the reference's own training scripts ASK of the `hugectr` module, as data.

The GPU box has no reference checkout (and reference sources are never copied into this repo), so
the drop-in tests cannot `runpy` R/samples/dcn/dcn_parquet.py there.  Instead this script -- run in
the build container, where /root/reference exists -- executes each script UNMODIFIED against a
recording stand-in for `hugectr` (no GPU, no implementation: every attribute is a name, every call
is written down) and stores the sequence of calls with their exact arguments in
tests/golden/script_traces.json.  tests/test_dropin_gpu.py replays those calls, one for one,
against the real `import hugectr` of this repo on an MI355X; tests/test_dropin_cpu.py re-records
when the checkout is present and demands the committed file be identical.

Encoding: {"name": "Layer_t.MLP"} = the module attribute hugectr.Layer_t.MLP (enum value, class,
function); {"ref": i} = the object returned by call i; calls are {"call": name | {"ref", "method"},
"args": [...], "kwargs": {...}}; dicts whose keys are not strings are {"pairs": [[k, v], ...]}.
"""
import json
import os
import runpy
import sys
import types

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
REF = "/root/reference"

SCRIPTS = {
    # name: (path under the reference, argv)
    "dcn_parquet": ("samples/dcn/dcn_parquet.py", []),
    "deepfm_parquet": ("samples/deepfm/deepfm_parquet.py", []),
    "wdl_1gpu": ("samples/wdl/wdl_1gpu.py", []),
    # BASELINE configs[4] names MMoE: two labels, two BinaryCrossEntropyLoss layers,
    # compile(loss_names, loss_weights), Softmax gates, Scale / ElementwiseMultiply expert mixing
    "mmoe_parquet": ("samples/mmoe/mmoe_parquet.py", []),
    "dgx_a100_one_hot": ("test/embedding_collection_test/dgx_a100_one_hot.py",
                         ["--num_gpus_per_node", "1", "--batchsize", "8192", "--batchsize_eval",
                          "8192", "--max_iter", "24", "--eval_interval", "12",
                          "--max_eval_batches", "2", "--display_interval", "8", "--lr", "0.5",
                          "--warmup_steps", "4", "--decay_start", "12", "--decay_steps", "8"]),
    # MLPerf DLRM-DCNv2 (BASELINE configs[4]'s API family: embedding_collection, multi-hot inputs,
    # the sample's own sharding planner -- imported from the checkout while recording -- and a
    # hugectr.TrainingCallback subclass of the sample's mlperf_logger package)
    "dlrm_mlperf_dcnv2": ("samples/dlrm/train.py",
                          ["--num_gpus_per_node", "1", "--batchsize", "8192", "--batchsize_eval",
                           "8192", "--max_iter", "24", "--eval_interval", "12",
                           "--max_eval_batches", "2", "--display_interval", "8", "--lr", "0.005",
                           "--sharding_plan", "auto", "--memory_cap_for_embedding", "250",
                           "--use_mixed_precision", "--scaler", "1024"]),
}


class Recorder:
    def __init__(self):
        self.calls = []

    def enc(self, v):
        if isinstance(v, _Name):
            return {"name": v._path}
        if isinstance(v, _Obj):
            return {"ref": v._idx}
        if isinstance(v, _CallbackBase):
            # an instance of a script-side subclass of hugectr.TrainingCallback: code cannot
            # travel as data, so the class name and its plain public attributes do; the replaying
            # test supplies a TrainingCallback of its own in that place
            return {"callback": type(v).__name__,
                    "attrs": {k: x for k, x in sorted(vars(v).items())
                              if not k.startswith("_") and isinstance(x, (int, float, str, bool))}}
        if isinstance(v, (list, tuple)):
            return [self.enc(x) for x in v]
        if isinstance(v, dict):
            if all(isinstance(k, str) for k in v):
                return {"dict": {k: self.enc(x) for k, x in v.items()}}
            return {"pairs": [[self.enc(k), self.enc(x)] for k, x in v.items()]}
        if isinstance(v, (int, float, str, bool)) or v is None:
            return v
        raise TypeError(f"cannot record {type(v)}")

    def record(self, target, args, kwargs):
        self.calls.append({"call": target, "args": [self.enc(a) for a in args],
                           "kwargs": {k: self.enc(x) for k, x in kwargs.items()}})
        return _Obj(self, len(self.calls) - 1)


class _Name:
    """hugectr.<path>: an enum value / class / function of the module, or a call of it"""

    def __init__(self, rec, path):
        object.__setattr__(self, "_rec", rec)
        object.__setattr__(self, "_path", path)

    def __getattr__(self, attr):
        if attr.startswith("__"):
            raise AttributeError(attr)
        return _Name(self._rec, f"{self._path}.{attr}")

    def __call__(self, *args, **kwargs):
        return self._rec.record(self._path, args, kwargs)

    def __hash__(self):
        return hash(self._path)

    def __eq__(self, other):
        return isinstance(other, _Name) and other._path == self._path


class _CallbackBase:
    """hugectr.TrainingCallback while recording: scripts subclass it (samples/dlrm/mlperf_logger)"""


class _Quiet:
    """stands in for MLPerf's logging objects (mlperf_common / mlperf_logging are not installed
    and are no part of the hugectr surface): every attribute is a callable that returns another"""

    def __init__(self, *a, **k):
        pass

    def __getattr__(self, attr):
        if attr.startswith("__"):
            raise AttributeError(attr)
        return _Quiet()

    def __call__(self, *a, **k):
        return _Quiet()


def _mlperf_stubs():
    mods = {}
    for name in ("mlperf_logging", "mlperf_logging.mllog", "mlperf_logging.mllog.constants",
                 "mlperf_common", "mlperf_common.frameworks", "mlperf_common.frameworks.hugectr",
                 "mlperf_common.logging"):
        m = types.ModuleType(name)
        m.__path__ = []
        mods[name] = m
    mods["mlperf_logging.mllog.constants"].__getattr__ = lambda attr: attr  # a constant = its name
    mods["mlperf_logging.mllog"].constants = mods["mlperf_logging.mllog.constants"]
    mods["mlperf_logging"].mllog = mods["mlperf_logging.mllog"]
    mods["mlperf_common.frameworks.hugectr"].HCTRCommunicationHandler = _Quiet
    mods["mlperf_common.logging"].MLLoggerWrapper = _Quiet
    return mods


class _Obj:
    """the result of a recorded call; its methods are recorded too"""

    def __init__(self, rec, idx):
        object.__setattr__(self, "_rec", rec)
        object.__setattr__(self, "_idx", idx)

    def __getattr__(self, attr):
        if attr.startswith("__"):
            raise AttributeError(attr)
        rec, idx = self._rec, self._idx
        return lambda *a, **k: rec.record({"ref": idx, "method": attr}, a, k)


def _stub_modules(rec):
    m = types.ModuleType("hugectr")
    m.__getattr__ = lambda attr: (_CallbackBase if attr == "TrainingCallback"
                                  else _Name(rec, attr))  # PEP 562
    m.__path__ = []
    t = types.ModuleType("hugectr.tools")
    t.__getattr__ = lambda attr: _Name(rec, f"tools.{attr}")
    m.tools = t
    return {"hugectr": m, "hugectr.tools": t}


def record(script_path, argv):
    rec = Recorder()
    stubs = {**_stub_modules(rec), **_mlperf_stubs()}
    before = set(sys.modules)
    saved = {k: sys.modules.get(k) for k in list(stubs) + ["mpi4py"]}
    saved_argv, saved_getsize = sys.argv, os.path.getsize
    sys.modules.update(stubs)
    if saved["mpi4py"] is None:
        sys.path.insert(0, os.path.join(ROOT, "hugectr_amd", "compat"))
    # packages that sit next to the script (samples/dlrm/sharding, mlperf_logger)
    sys.path.insert(0, os.path.dirname(script_path))
    # scripts log the row count of dataset files that only exist on a training machine
    os.path.getsize = lambda f: saved_getsize(f) if os.path.exists(f) else 0
    try:
        sys.argv = [script_path] + list(argv)
        runpy.run_path(script_path, run_name="__main__")
    finally:
        sys.argv, os.path.getsize = saved_argv, saved_getsize
        sys.path.remove(os.path.dirname(script_path))
        for k in set(sys.modules) - before:  # script-side packages must not outlive the recording
            sys.modules.pop(k, None)
        for k, v in saved.items():
            if v is not None:
                sys.modules[k] = v
        if saved["mpi4py"] is None:
            sys.path.remove(os.path.join(ROOT, "hugectr_amd", "compat"))
    return rec.calls


def record_all():
    out = {}
    for name, (rel, argv) in SCRIPTS.items():
        out[name] = {"script": rel, "argv": argv, "calls": record(os.path.join(REF, rel), argv)}
    return out


if __name__ == "__main__":
    traces = record_all()
    with open(os.path.join(HERE, "script_traces.json"), "w") as f:
        json.dump(traces, f, indent=0, sort_keys=True)
    for k, v in traces.items():
        print(k, len(v["calls"]), "calls")
