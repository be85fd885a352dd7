"""The first box with two GPUs produces parity evidence on its own: 2 ranks on 2 devices over the
`nccl` backend (= RCCL over xGMI) train the DLRM graph at a global batch of 8192 and must end where
ONE rank ends on the same batches -- losses, every embedding vector (as a key -> vector map), the
dense weights -- with the all-to-all of pooled rows (the reference's exchange,
R/HugeCTR/src/embeddings/all2all_forward_functor.cu:157-264) and with the unique-row payload, the
asynchronous schedule on (all-to-all under the bottom MLP, gradient all-to-all from backward).
Skipped on 1-GPU boxes (every `gpurun` box and the driver's GPU tier so far: no N > 1 RCCL run
exists in this repository's history).  HCTR_MULTI_GPU_BACKEND=gloo runs the same assertions with
both ranks on device 0 over gloo -- a check of the TEST, not of RCCL."""
import os

import numpy as np
import pytest
import torch

from test_model_gpu import SIZES, _gen, _parity_run

pytestmark = pytest.mark.gpu

_STANDIN = os.environ.get("HCTR_MULTI_GPU_BACKEND") == "gloo"
B, D, STEPS = 8192, 32, 5


def _two_devices():
    return torch.cuda.is_available() and torch.cuda.device_count() >= 2


def _model(hugectr, folder, world, overlap, mixed):
    solver = hugectr.CreateSolver(batchsize=B, batchsize_eval=B, lr=0.05, vvgpu=[list(range(world))],
                                  i64_input_key=True, max_eval_batches=1, use_mixed_precision=mixed,
                                  scaler=128.0 if mixed else 1.0,
                                  train_intra_iteration_overlap=overlap,
                                  train_inter_iteration_overlap=overlap)
    reader = hugectr.DataReaderParams(
        data_reader_type=hugectr.DataReaderType_t.Parquet,
        source=[os.path.join(folder, "train", "_file_list.txt")],
        eval_source=os.path.join(folder, "val", "_file_list.txt"), slot_size_array=SIZES,
        check_type=hugectr.Check_t.Non)
    opt = hugectr.CreateOptimizer(optimizer_type=hugectr.Optimizer_t.SGD,
                                  update_type=hugectr.Update_t.Local, atomic_update=False)
    model = hugectr.Model(solver, reader, opt)
    model.add(hugectr.Input(label_dim=1, label_name="label", dense_dim=13, dense_name="dense",
                            data_reader_sparse_param_array=[
                                hugectr.DataReaderSparseParam("data1", 1, True, 26)]))
    L, T, A = hugectr.DenseLayer, hugectr.Layer_t, hugectr.Activation_t
    model.add(hugectr.SparseEmbedding(
        embedding_type=hugectr.Embedding_t.LocalizedSlotSparseEmbeddingHash,
        slot_size_array=SIZES, embedding_vec_size=D, combiner="sum",
        sparse_embedding_name="emb", bottom_name="data1", optimizer=opt))
    model.add(L(layer_type=T.MLP, bottom_names=["dense"], top_names=["mlp1"], num_outputs=[64, D],
                act_type=A.Relu))
    model.add(L(layer_type=T.Interaction, bottom_names=["mlp1", "emb"], top_names=["inter"]))
    model.add(L(layer_type=T.MLP, bottom_names=["inter"], top_names=["mlp2"],
                num_outputs=[128, 64, 1], activations=[A.Relu, A.Relu, A.Non]))
    model.add(L(layer_type=T.BinaryCrossEntropyLoss, bottom_names=["mlp2", "label"],
                top_names=["loss"]))
    model.compile()
    model.load_sparse_weights([os.path.join(folder, "init_sparse")])
    return model


def _worker(rank, world, port, folder, backend, exchange, mixed, ret):
    import torch.distributed as dist
    local = rank if backend == "nccl" else 0
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), LOCAL_RANK=str(local),
                      HCTR_EXCHANGE=exchange, HSA_ENABLE_IPC_MODE_LEGACY="0")
    torch.cuda.set_device(local)
    kw = dict(device_id=torch.device("cuda", local)) if backend == "nccl" else {}
    dist.init_process_group(backend, rank=rank, world_size=world, **kw)
    try:
        assert dist.get_backend() == backend
        import hugectr_amd.hugectr as hugectr
        torch.manual_seed(5)
        model = _model(hugectr, folder, world, True, mixed)
        losses, k, v, dense = _parity_run(model, STEPS)
        rep = model.exchange_report()["emb"]
        ret[rank] = ("ok", losses, k, v, dense, rep, torch.cuda.current_device())
    except Exception as ex:
        import traceback
        ret[rank] = ("".join(traceback.format_exception(type(ex), ex, ex.__traceback__)),)
    finally:
        dist.destroy_process_group()


@pytest.mark.skipif(not (_two_devices() or _STANDIN),
                    reason="needs 2 GPUs (HCTR_MULTI_GPU_BACKEND=gloo: the stand-in on one)")
@pytest.mark.parametrize("mixed", [False, True], ids=["fp32", "fp16"])
@pytest.mark.parametrize("exchange", ["rows", "unique"])
def test_two_ranks_on_two_gpus_train_like_one_rank(tmp_path, exchange, mixed):
    import hugectr_amd.hugectr as hugectr
    import torch.multiprocessing as mp
    from numpy.testing import assert_allclose
    backend = "nccl" if _two_devices() and not _STANDIN else "gloo"
    _gen(tmp_path, hugectr, n_train=2 * B, n_eval=B)
    rng = np.random.default_rng(3)
    V = sum(SIZES)
    d = tmp_path / "init_sparse"
    d.mkdir()
    np.arange(V, dtype="<i8").tofile(d / "key")
    np.repeat(np.arange(26), SIZES).astype("<u8").tofile(d / "slot_id")
    (rng.standard_normal((V, D)) * 0.1).astype("<f4").tofile(d / "emb_vector")
    torch.manual_seed(5)
    one = _parity_run(_model(hugectr, str(tmp_path), 1, False, mixed), STEPS)
    ctx = mp.get_context("spawn")
    ret = ctx.Manager().dict()
    port = 29500 + os.getpid() % 2000 + (31 if exchange == "rows" else 37) + (2 if mixed else 0)
    procs = [ctx.Process(target=_worker, args=(r, 2, port, str(tmp_path), backend, exchange, mixed, ret))
             for r in range(2)]
    for p in procs:
        p.start()
    for p in procs:
        p.join(900)
    for r in range(2):
        assert ret.get(r) is not None and ret[r][0] == "ok", ret.get(r)
    if backend == "nccl":
        assert {ret[0][6], ret[1][6]} == {0, 1}, "the two ranks did not run on two devices"
    for r in range(2):
        assert ret[r][5]["payload"] == exchange and ret[r][5]["intra_iteration_overlap"]
    tol = dict(rtol=2e-2, atol=2e-3) if mixed else dict(rtol=2e-4, atol=2e-6)
    # rank-local losses average to the global-batch loss
    two_loss = (np.array(ret[0][1]) + np.array(ret[1][1])) / 2
    assert_allclose(two_loss, np.array(one[0]), rtol=5e-3 if mixed else 2e-5)
    for r in range(2):
        assert_allclose(ret[r][4], one[3], **tol)  # dense weights: the same on both ranks
    assert (ret[0][4] == ret[1][4]).all()
    k2 = np.concatenate([ret[0][2], ret[1][2]])
    v2 = np.concatenate([ret[0][3], ret[1][3]])
    assert len(np.unique(k2)) == k2.size == one[1].size  # every key on exactly one rank
    o1, o2 = np.argsort(one[1]), np.argsort(k2)
    assert (one[1][o1] == k2[o2]).all()
    assert_allclose(v2[o2], one[2][o1], **tol)
    moved = np.abs(one[2][o1] - np.fromfile(d / "emb_vector", "<f4").reshape(V, D)).max()
    assert moved > 1e-5, "the embedding did not train"  # (mean over 8192 samples, 5 steps)
