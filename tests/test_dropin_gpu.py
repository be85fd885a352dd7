"""Drop-in boundary on the GPU: the exact calls the reference's own UNMODIFIED training scripts make
(tests/golden/script_traces.json, recorded from R/samples/dcn/dcn_parquet.py,
R/samples/deepfm/deepfm_parquet.py, R/samples/wdl/wdl_1gpu.py,
R/samples/mmoe/mmoe_parquet.py (two tasks, two losses),
R/test/embedding_collection_test/dgx_a100_one_hot.py and the MLPerf DLRM-DCNv2 sample
R/samples/dlrm/train.py by tests/golden/make_script_traces.py) are
executed against `import hugectr` -- this repo's module of the reference's name -- on an MI355X,
with generated data where the scripts expect theirs.  The GPU box has no reference checkout, which
is why the scripts travel as recorded calls; where the checkout exists the script files themselves
are run with runpy as well.  Only test-speed caps are applied (iterations, evaluation batches)."""
import gc
import math
import os
import subprocess
import sys

import numpy as np
import pytest

from dropin_replay import load_traces, replay

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
REF = "/root/reference"


def _find(calls, name):
    return next(c for c in calls if c["call"] == name)


def _make_data(hugectr, calls, tmp_path):
    solver = _find(calls, "CreateSolver")["kwargs"]
    reader = _find(calls, "DataReaderParams")["kwargs"]
    B, Be = solver["batchsize"], solver["batchsize_eval"]
    sizes = reader["slot_size_array"]
    fmt = reader["data_reader_type"]["name"]
    if fmt == "DataReaderType_t.Parquet":
        inp = _find(calls, "Input")["kwargs"]
        label_dim = sum(inp["label_dims"]) if "label_dims" in inp else inp.get("label_dim", 1)
        hugectr.tools.DataGenerator(hugectr.tools.DataGeneratorParams(
            format=hugectr.DataReaderType_t.Parquet, label_dim=label_dim,
            dense_dim=inp.get("dense_dim", 13),
            num_slot=len(sizes), i64_input_key=bool(solver.get("i64_input_key", False)),
            source=reader["source"][0], eval_source=reader["eval_source"], slot_size_array=sizes,
            dist_type=hugectr.Distribution_t.PowerLaw, power_law_type=hugectr.PowerLaw_t.Short,
            num_files=1, eval_num_files=1, num_samples_per_file=B * 3, num_samples=B * 3,
            eval_num_samples=Be)).generate()
        return
    assert fmt == "DataReaderType_t.RawAsync"
    # Raw file of the multi-hot async reader: per sample {label i32, dense f32 x 13, then for
    # every table its hotness u32 keys (within the table; the reader adds the table offsets)}
    hot = [c["args"][1] for c in calls if c["call"] == "DataReaderSparseParam"]
    assert len(hot) == len(sizes)
    rng = np.random.default_rng(0)
    for path, n in ((reader["source"][0], B * 4), (reader["eval_source"], Be * 2)):
        f = os.path.join(str(tmp_path), path.lstrip("/"))
        os.makedirs(os.path.dirname(f), exist_ok=True)
        a = np.zeros((n, 1 + 13 + sum(hot)), dtype="<u4")
        keys = np.concatenate([np.minimum(rng.zipf(1.3, (n, h)) - 1, v - 1)
                               for v, h in zip(sizes, hot)], 1)
        a[:, 0] = (keys[:, sum(hot[:2])] % 2).astype("<i4").view("<u4")  # a learnable label
        a[:, 1:14] = rng.random((n, 13), dtype=np.float32).view("<u4")
        a[:, 14:] = keys.astype("<u4")
        a.tofile(f)


def _caps(i, target, args, kwargs):
    if target == "CreateSolver":
        kwargs["max_eval_batches"] = min(kwargs.get("max_eval_batches", 2), 2)
    if isinstance(target, dict) and target["method"] == "fit":
        kwargs["max_iter"] = min(kwargs.get("max_iter", 24), 24)
        kwargs["eval_interval"] = 12
        kwargs["display"] = 8
        kwargs["snapshot"] = 10 ** 9


@pytest.mark.parametrize("script", sorted(load_traces()))
def test_reference_script_calls_run_on_the_gpu(script, tmp_path, monkeypatch, capsys):
    import hugectr
    import torch
    tr = load_traces()[script]
    monkeypatch.chdir(tmp_path)
    monkeypatch.setenv("HCTR_DATA_ROOT", str(tmp_path))
    _make_data(hugectr, tr["calls"], tmp_path)
    res = replay(hugectr, tr["calls"], before=_caps)
    model = next(r for r in res if isinstance(r, hugectr.Model))
    torch.cuda.synchronize()
    assert model._iter == 24
    out = capsys.readouterr().out
    assert "Finish 24 iterations" in out and "Evaluation, AUC" in out
    for cb in _find(tr["calls"], "CreateSolver")["kwargs"].get("training_callbacks", []):
        # the script's hugectr.TrainingCallback subclass (samples/dlrm/mlperf_logger): its stand-in
        # must have been driven through the whole protocol (training_callback.hpp:20-26)
        live = next(c for c in model.solver.training_callbacks if c.name == cb["callback"])
        kinds = [e[0] for e in live.events]
        assert kinds[0] == "training_start" and kinds[-1] == "training_end"
        evals = [e for e in live.events if e[0] == "eval_end"]
        assert len(evals) == 2 and all("AUC" in e[2] for e in evals)
    solver = _find(tr["calls"], "CreateSolver")["kwargs"]
    if solver.get("gen_loss_summary", True):
        assert math.isfinite(model.get_current_loss())
    del model, res
    gc.collect()  # (Model <-> layers reference cycles hold 100+ GB tables until collected)
    torch.cuda.empty_cache()


@pytest.mark.skipif(not os.path.isdir(REF), reason="the reference checkout is not on the GPU box")
def test_the_script_file_itself_runs(tmp_path, monkeypatch):
    """where the checkout and a GPU are both present: runpy of the unmodified file"""
    import runpy
    import hugectr
    tr = load_traces()["dcn_parquet"]
    monkeypatch.chdir(tmp_path)
    _make_data(hugectr, tr["calls"], tmp_path)
    fit = hugectr.Model.fit
    monkeypatch.setattr(hugectr.Model, "fit",
                        lambda self, **k: fit(self, **{**k, "max_iter": 24, "eval_interval": 0}))
    if "mpi4py" not in sys.modules:
        monkeypatch.syspath_prepend(os.path.join(ROOT, "hugectr_amd", "compat"))
    runpy.run_path(os.path.join(REF, tr["script"]), run_name="__main__")


_TWO_GPU_SCRIPT = '''
import os, sys
import hugectr
solver = hugectr.CreateSolver(max_eval_batches=1, batchsize_eval=512, batchsize=512, lr=0.01,
                              vvgpu=[[0, 1]], repeat_dataset=True, i64_input_key=True)
open("started.%s" % os.environ.get("RANK", "parent"), "w").close()
reader = hugectr.DataReaderParams(data_reader_type=hugectr.DataReaderType_t.Parquet,
                                  source=["./data/train/_file_list.txt"],
                                  eval_source="./data/val/_file_list.txt",
                                  slot_size_array=SIZES, check_type=hugectr.Check_t.Non)
optimizer = hugectr.CreateOptimizer(optimizer_type=hugectr.Optimizer_t.SGD,
                                    update_type=hugectr.Update_t.Local)
model = hugectr.Model(solver, reader, optimizer)
model.add(hugectr.Input(label_dim=1, label_name="label", dense_dim=13, dense_name="dense",
                        data_reader_sparse_param_array=[
                            hugectr.DataReaderSparseParam("data1", 1, True, len(SIZES))]))
model.add(hugectr.SparseEmbedding(
    embedding_type=hugectr.Embedding_t.LocalizedSlotSparseEmbeddingHash,
    workspace_size_per_gpu_in_mb=8, embedding_vec_size=16, combiner="sum",
    sparse_embedding_name="emb", bottom_name="data1", slot_size_array=SIZES, optimizer=optimizer))
model.add(hugectr.DenseLayer(layer_type=hugectr.Layer_t.Reshape, bottom_names=["emb"],
                             top_names=["flat"], leading_dim=16 * len(SIZES)))
model.add(hugectr.DenseLayer(layer_type=hugectr.Layer_t.Concat, bottom_names=["flat", "dense"],
                             top_names=["cat"]))
model.add(hugectr.DenseLayer(layer_type=hugectr.Layer_t.MLP, bottom_names=["cat"],
                             top_names=["mlp"], num_outputs=[64, 1],
                             activations=[hugectr.Activation_t.Relu, hugectr.Activation_t.Non]))
model.add(hugectr.DenseLayer(layer_type=hugectr.Layer_t.BinaryCrossEntropyLoss,
                             bottom_names=["mlp", "label"], top_names=["loss"]))
model.compile()
model.fit(max_iter=20, display=10, eval_interval=0, snapshot=0)
open("done.%s.%d" % (os.environ["RANK"], model.world), "w").close()
'''


def test_python_script_with_two_gpus_in_vvgpu_starts_its_own_ranks(tmp_path):
    """`python train.py` with vvgpu=[[0, 1]] (the reference: one process, one OpenMP thread per
    GPU): CreateSolver starts one process per GPU itself.  Both ranks share this box's one GPU and
    talk over gloo here (HCTR_RANKS_ON_ONE_GPU / HCTR_DIST_BACKEND are test knobs; on a multi-GPU
    node the same path runs RCCL)."""
    import hugectr
    sizes = [50, 7, 120, 33, 4, 90]
    d = tmp_path
    hugectr.tools.DataGenerator(hugectr.tools.DataGeneratorParams(
        format=hugectr.DataReaderType_t.Parquet, label_dim=1, dense_dim=13, num_slot=len(sizes),
        i64_input_key=True, source=str(d / "data" / "train" / "_file_list.txt"),
        eval_source=str(d / "data" / "val" / "_file_list.txt"), slot_size_array=sizes,
        dist_type=hugectr.Distribution_t.PowerLaw, power_law_type=hugectr.PowerLaw_t.Short,
        num_files=1, eval_num_files=1, num_samples_per_file=2048, num_samples=2048,
        eval_num_samples=512)).generate()
    (d / "train.py").write_text(f"SIZES = {sizes}\n" + _TWO_GPU_SCRIPT)
    env = dict(os.environ, PYTHONPATH=ROOT + os.pathsep + os.environ.get("PYTHONPATH", ""),
               HCTR_DIST_BACKEND="gloo", HCTR_RANKS_ON_ONE_GPU="1")
    for k in ("RANK", "WORLD_SIZE", "LOCAL_RANK", "MASTER_ADDR", "MASTER_PORT"):
        env.pop(k, None)
    r = subprocess.run([sys.executable, "train.py"], cwd=str(d), env=env, capture_output=True,
                       text=True, timeout=600)
    assert r.returncode == 0, r.stdout[-3000:] + r.stderr[-3000:]
    assert (d / "done.0.2").exists() and (d / "done.1.2").exists()
    assert not (d / "started.parent").exists()  # the parent stopped at CreateSolver
    assert "Finish 20 iterations" in r.stdout
