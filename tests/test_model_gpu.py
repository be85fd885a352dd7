"""GPU: existing-style hugectr scripts run through the hugectr_amd.hugectr surface (config 1:
DCN on README-style synthetic parquet; DLRM-style model with Interaction)."""
import os

import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu

SIZES = [203, 185, 140, 70, 189, 4, 63, 12, 49, 186, 71, 67, 11, 21, 73, 61, 4, 93, 15, 204, 141,
         199, 60, 91, 71, 34]


def _gen(tmp_path, hugectr, n_train=8192, n_eval=2048, nnz=None):
    p = hugectr.tools.DataGeneratorParams(
        format=hugectr.DataReaderType_t.Parquet, label_dim=1, dense_dim=13, num_slot=26,
        i64_input_key=True, source=str(tmp_path / "train" / "_file_list.txt"),
        eval_source=str(tmp_path / "val" / "_file_list.txt"), slot_size_array=SIZES,
        nnz_array=nnz or [], dist_type=hugectr.Distribution_t.PowerLaw,
        power_law_type=hugectr.PowerLaw_t.Short, num_files=2, eval_num_files=1,
        num_samples_per_file=n_train // 2, num_samples=n_train, eval_num_samples=n_eval)
    hugectr.tools.DataGenerator(p).generate()
    # make the label learnable (the generator draws coin flips): label = parity of feature C1
    import glob
    import pyarrow as pa
    import pyarrow.parquet as pq
    for f in glob.glob(str(tmp_path / "*" / "*.parquet")):
        t = pq.read_table(f)
        one_hot = next(f"C{i + 1}" for i in range(26) if not nnz or nnz[i] == 1)
        lab = (t[one_hot].to_numpy() % 2).astype(np.float32)
        t = t.set_column(t.schema.get_field_index("label"), "label", pa.array(lab, type=pa.float32()))
        pq.write_table(t, f)
    return p


def test_dcn_script_trains(tmp_path, capsys):
    import hugectr_amd.hugectr as hugectr
    p = _gen(tmp_path, hugectr)
    solver = hugectr.CreateSolver(max_eval_batches=2, batchsize_eval=1024, batchsize=1024, lr=0.001,
                                  vvgpu=[[0]], repeat_dataset=True, i64_input_key=True)
    reader = hugectr.DataReaderParams(data_reader_type=hugectr.DataReaderType_t.Parquet,
                                      source=[p.source], eval_source=p.eval_source,
                                      slot_size_array=SIZES, check_type=hugectr.Check_t.Non)
    optimizer = hugectr.CreateOptimizer(optimizer_type=hugectr.Optimizer_t.Adam,
                                        update_type=hugectr.Update_t.Global)
    model = hugectr.Model(solver, reader, optimizer)
    model.add(hugectr.Input(label_dim=1, label_name="label", dense_dim=13, dense_name="dense",
                            data_reader_sparse_param_array=[
                                hugectr.DataReaderSparseParam("data1", 1, False, 26)]))
    model.add(hugectr.SparseEmbedding(
        embedding_type=hugectr.Embedding_t.DistributedSlotSparseEmbeddingHash,
        workspace_size_per_gpu_in_mb=2, embedding_vec_size=16, combiner="sum",
        sparse_embedding_name="sparse_embedding1", bottom_name="data1", optimizer=optimizer))
    model.add(hugectr.DenseLayer(layer_type=hugectr.Layer_t.Reshape,
                                 bottom_names=["sparse_embedding1"], top_names=["reshape1"],
                                 leading_dim=416))
    model.add(hugectr.DenseLayer(layer_type=hugectr.Layer_t.Concat,
                                 bottom_names=["reshape1", "dense"], top_names=["concat1"]))
    model.add(hugectr.DenseLayer(layer_type=hugectr.Layer_t.MultiCross, bottom_names=["concat1"],
                                 top_names=["multicross1"], num_layers=6))
    model.add(hugectr.DenseLayer(layer_type=hugectr.Layer_t.InnerProduct, bottom_names=["concat1"],
                                 top_names=["fc1"], num_output=256))
    model.add(hugectr.DenseLayer(layer_type=hugectr.Layer_t.ReLU, bottom_names=["fc1"],
                                 top_names=["relu1"]))
    model.add(hugectr.DenseLayer(layer_type=hugectr.Layer_t.Dropout, bottom_names=["relu1"],
                                 top_names=["dropout1"], dropout_rate=0.5))
    model.add(hugectr.DenseLayer(layer_type=hugectr.Layer_t.Concat,
                                 bottom_names=["dropout1", "multicross1"], top_names=["concat2"]))
    model.add(hugectr.DenseLayer(layer_type=hugectr.Layer_t.InnerProduct, bottom_names=["concat2"],
                                 top_names=["fc2"], num_output=1))
    model.add(hugectr.DenseLayer(layer_type=hugectr.Layer_t.BinaryCrossEntropyLoss,
                                 bottom_names=["fc2", "label"], top_names=["loss"]))
    model.compile()
    model.summary()
    model.train()
    first = model.get_current_loss()
    model.fit(max_iter=120, display=40, eval_interval=60, snapshot=120,
              snapshot_prefix=str(tmp_path / "dcn"))
    last = model.get_current_loss()
    assert np.isfinite(first) and np.isfinite(last)
    assert last < first and last < 0.5  # label = parity of C1 is learnable from the embedding
    out = capsys.readouterr().out
    assert "Iter: 40" in out and "Evaluation, AUC" in out
    # checkpoint in the reference's directory layout, then reload into a fresh model
    d = str(tmp_path / "dcn0_sparse_120.model")
    assert os.path.exists(os.path.join(d, "key")) and os.path.exists(os.path.join(d, "emb_vector"))
    nkeys = os.path.getsize(os.path.join(d, "key")) // 8
    assert os.path.getsize(os.path.join(d, "emb_vector")) == nkeys * 16 * 4
    model.graph_to_json(str(tmp_path / "dcn.json"))
    assert "MultiCross" in open(tmp_path / "dcn.json").read()


def test_dlrm_style_script_trains(tmp_path):
    import hugectr_amd.hugectr as hugectr
    p = _gen(tmp_path, hugectr, n_train=4096, n_eval=1024)
    solver = hugectr.CreateSolver(batchsize=512, batchsize_eval=512, lr=0.01, vvgpu=[[0]],
                                  i64_input_key=True, max_eval_batches=1)
    reader = hugectr.DataReaderParams(data_reader_type=hugectr.DataReaderType_t.Parquet,
                                      source=[p.source], eval_source=p.eval_source,
                                      slot_size_array=SIZES, check_type=hugectr.Check_t.Non)
    opt = hugectr.CreateOptimizer(optimizer_type=hugectr.Optimizer_t.Adam,
                                  update_type=hugectr.Update_t.Local)
    model = hugectr.Model(solver, reader, opt)
    model.add(hugectr.Input(label_dim=1, label_name="label", dense_dim=13, dense_name="dense",
                            data_reader_sparse_param_array=[
                                hugectr.DataReaderSparseParam("data1", 1, True, 26)]))
    model.add(hugectr.SparseEmbedding(
        embedding_type=hugectr.Embedding_t.LocalizedSlotSparseEmbeddingHash,
        slot_size_array=SIZES, embedding_vec_size=32, combiner="sum",
        sparse_embedding_name="sparse_embedding1", bottom_name="data1", optimizer=opt))
    model.add(hugectr.DenseLayer(layer_type=hugectr.Layer_t.MLP, bottom_names=["dense"],
                                 top_names=["mlp1"], num_outputs=[64, 32],
                                 act_type=hugectr.Activation_t.Relu, use_bias=True))
    model.add(hugectr.DenseLayer(layer_type=hugectr.Layer_t.Interaction,
                                 bottom_names=["mlp1", "sparse_embedding1"],
                                 top_names=["interaction1"]))
    model.add(hugectr.DenseLayer(layer_type=hugectr.Layer_t.MLP, bottom_names=["interaction1"],
                                 top_names=["mlp2"], num_outputs=[128, 1],
                                 activations=[hugectr.Activation_t.Relu, hugectr.Activation_t.Non]))
    model.add(hugectr.DenseLayer(layer_type=hugectr.Layer_t.BinaryCrossEntropyLoss,
                                 bottom_names=["mlp2", "label"], top_names=["loss"]))
    model.compile()
    model.train()
    first = model.get_current_loss()
    model.fit(max_iter=300, display=100, eval_interval=0, snapshot=0)
    assert model.get_current_loss() < min(first, 0.6)


def test_deepfm_script_trains(tmp_path):
    """BASELINE config 2: the layer graph of R/samples/deepfm/deepfm_parquet.py (embedding_vec_size
    11 -> Slice 10 + 1, WeightMultiply, FmOrder2, ReduceSum, Add) on a Criteo-Kaggle-shaped
    synthetic Parquet set, Adam with the Global update, as the sample configures it."""
    import hugectr_amd.hugectr as hugectr
    _gen(tmp_path, hugectr)
    solver = hugectr.CreateSolver(max_eval_batches=4, batchsize_eval=512, batchsize=512, lr=0.002,
                                  vvgpu=[[0]], repeat_dataset=True, i64_input_key=True)
    reader = hugectr.DataReaderParams(
        data_reader_type=hugectr.DataReaderType_t.Parquet,
        source=[str(tmp_path / "train" / "_file_list.txt")],
        eval_source=str(tmp_path / "val" / "_file_list.txt"), slot_size_array=SIZES,
        check_type=hugectr.Check_t.Non)
    opt = hugectr.CreateOptimizer(optimizer_type=hugectr.Optimizer_t.Adam,
                                  update_type=hugectr.Update_t.Global, beta1=0.9, beta2=0.999,
                                  epsilon=1e-7)
    model = hugectr.Model(solver, reader, opt)
    L, D = hugectr.Layer_t, hugectr.DenseLayer
    model.add(hugectr.Input(label_dim=1, label_name="label", dense_dim=13, dense_name="dense",
                            data_reader_sparse_param_array=[
                                hugectr.DataReaderSparseParam("data1", 1, False, 26)]))
    model.add(hugectr.SparseEmbedding(
        embedding_type=hugectr.Embedding_t.LocalizedSlotSparseEmbeddingHash,
        workspace_size_per_gpu_in_mb=8, embedding_vec_size=11, combiner="sum",
        sparse_embedding_name="sparse_embedding1", bottom_name="data1", slot_size_array=SIZES,
        optimizer=opt))
    model.add(D(layer_type=L.Reshape, bottom_names=["sparse_embedding1"], top_names=["reshape1"],
                leading_dim=11))
    model.add(D(layer_type=L.Slice, bottom_names=["reshape1"], top_names=["slice11", "slice12"],
                ranges=[(0, 10), (10, 11)]))
    model.add(D(layer_type=L.Reshape, bottom_names=["slice11"], top_names=["reshape2"],
                leading_dim=260))
    model.add(D(layer_type=L.Reshape, bottom_names=["slice12"], top_names=["reshape3"],
                leading_dim=26))
    model.add(D(layer_type=L.WeightMultiply, bottom_names=["dense"],
                top_names=["weight_multiply1"], weight_dims=[13, 10]))
    model.add(D(layer_type=L.WeightMultiply, bottom_names=["dense"],
                top_names=["weight_multiply2"], weight_dims=[13, 1]))
    model.add(D(layer_type=L.Concat, bottom_names=["reshape2", "weight_multiply1"],
                top_names=["concat1"]))
    prev = "concat1"
    for i in (1, 2, 3):
        model.add(D(layer_type=L.InnerProduct, bottom_names=[prev], top_names=[f"fc{i}"],
                    num_output=64))
        model.add(D(layer_type=L.ReLU, bottom_names=[f"fc{i}"], top_names=[f"relu{i}"]))
        model.add(D(layer_type=L.Dropout, bottom_names=[f"relu{i}"], top_names=[f"dropout{i}"],
                    dropout_rate=0.1))
        prev = f"dropout{i}"
    model.add(D(layer_type=L.InnerProduct, bottom_names=[prev], top_names=["fc4"], num_output=1))
    model.add(D(layer_type=L.FmOrder2, bottom_names=["concat1"], top_names=["fmorder2"],
                out_dim=10))
    model.add(D(layer_type=L.ReduceSum, bottom_names=["fmorder2"], top_names=["reducesum1"],
                axis=1))
    model.add(D(layer_type=L.Concat, bottom_names=["reshape3", "weight_multiply2"],
                top_names=["concat2"]))
    model.add(D(layer_type=L.ReduceSum, bottom_names=["concat2"], top_names=["reducesum2"],
                axis=1))
    model.add(D(layer_type=L.Add, bottom_names=["fc4", "reducesum1", "reducesum2"],
                top_names=["add"]))
    model.add(D(layer_type=L.BinaryCrossEntropyLoss, bottom_names=["add", "label"],
                top_names=["loss"]))
    model.compile()
    model.summary()
    model.train()
    first = model.get_current_loss()
    model.fit(max_iter=150, display=50, eval_interval=75, snapshot=1000000,
              snapshot_prefix=str(tmp_path / "deepfm"))
    last = model.get_current_loss()
    assert np.isfinite(first) and np.isfinite(last) and last < first and last < 0.55
    auc = dict(model.get_eval_metrics()).get("AUC")
    assert auc is not None and auc > 0.7


@pytest.mark.parametrize("plan,dynamic", [("round_robin", False), ("auto", False), ("auto", True)])
def test_dcnv2_embedding_collection_script_trains(tmp_path, plan, dynamic):
    """the graph of R/samples/dlrm/train.py (MLPerf DLRM-DCNv2): one multi-hot input per table, an
    embedding_collection sharded by the sample's planner, bottom MLP, concat, MultiCross with a
    projection, top MLP -- at a small synthetic shape"""
    import hugectr_amd.hugectr as hugectr
    from hugectr_amd import sharding
    hot = [1, 2, 1, 2, 6, 1, 1, 1, 1, 7, 3, 8, 1, 6, 9, 5, 1, 1, 1, 12, 10, 7, 4, 3, 1, 1]  # (C1 one-hot: _gen derives the label from it)
    p = _gen(tmp_path, hugectr, n_train=4096, n_eval=1024, nnz=hot)
    solver = hugectr.CreateSolver(batchsize=512, batchsize_eval=256, lr=0.003 if dynamic else 0.05,
                                  vvgpu=[[0]], i64_input_key=True, max_eval_batches=2,
                                  use_embedding_collection=True)
    reader = hugectr.DataReaderParams(data_reader_type=hugectr.DataReaderType_t.Parquet,
                                      source=[p.source], eval_source=p.eval_source,
                                      slot_size_array=SIZES, check_type=hugectr.Check_t.Non)
    # dynamic hash tables (max_vocabulary_size = -1, BASELINE config 5) take any optimizer
    opt = hugectr.CreateOptimizer(optimizer_type=hugectr.Optimizer_t.Adam if dynamic
                                  else hugectr.Optimizer_t.AdaGrad,
                                  update_type=hugectr.Update_t.Global, initial_accu_value=0.0)
    model = hugectr.Model(solver, reader, opt)
    model.add(hugectr.Input(label_dim=1, label_name="label", dense_dim=13, dense_name="dense",
                            data_reader_sparse_param_array=[
                                hugectr.DataReaderSparseParam(f"data{i}", hot[i], True, 1)
                                for i in range(26)]))
    tables = [hugectr.EmbeddingTableConfig(name=str(i), ev_size=16,
                                           max_vocabulary_size=-1 if dynamic else SIZES[i])
              for i in range(26)]
    args = sharding.mi355x_args(sharding_plan=plan, ev_size=16, optimizer="adagrad",
                                num_gpus_per_node=1)
    shard_matrix, shard_strategy = sharding.generate_plan(SIZES, hot, 1, 1, args, False)
    ebc = hugectr.EmbeddingCollectionConfig(use_exclusive_keys=True,
                                            comm_strategy=hugectr.CommunicationStrategy.Uniform)
    ebc.embedding_lookup(table_config=tables, bottom_name=[f"data{i}" for i in range(26)],
                         top_name="sparse_embedding", combiner=["sum"] * 26)
    ebc.shard(shard_matrix=shard_matrix, shard_strategy=shard_strategy)
    model.add(ebc)
    cc = hugectr.DenseLayerComputeConfig(async_wgrad=True, fuse_wb=False)
    model.add(hugectr.DenseLayer(layer_type=hugectr.Layer_t.MLP, bottom_names=["dense"],
                                 top_names=["mlp1"], num_outputs=[64, 16],
                                 act_type=hugectr.Activation_t.Relu, compute_config=cc))
    model.add(hugectr.DenseLayer(layer_type=hugectr.Layer_t.Concat,
                                 bottom_names=["mlp1", "sparse_embedding"], top_names=["concat1"]))
    model.add(hugectr.DenseLayer(layer_type=hugectr.Layer_t.MultiCross, bottom_names=["concat1"],
                                 top_names=["interaction1"], projection_dim=32, num_layers=2))
    model.add(hugectr.DenseLayer(layer_type=hugectr.Layer_t.MLP, bottom_names=["interaction1"],
                                 top_names=["mlp2"], num_outputs=[64, 1],
                                 activations=[hugectr.Activation_t.Relu, hugectr.Activation_t.Non]))
    model.add(hugectr.DenseLayer(layer_type=hugectr.Layer_t.BinaryCrossEntropyLoss,
                                 bottom_names=["mlp2", "label"], top_names=["loss"]))
    model.compile()
    model.summary()
    before = None if dynamic else model._ebc[0]["train"].table.clone()
    model.train()
    first = model.get_current_loss()
    model.fit(max_iter=200, display=100, eval_interval=100, snapshot=200,
              snapshot_prefix=str(tmp_path / "dcnv2"))
    assert model.get_current_loss() < min(first, 0.62)
    assert dict(model.get_eval_metrics())["AUC"] > 0.6
    if dynamic:
        e = model._ebc[0]["train"]
        assert e.det.size() > 500 and os.path.exists(
            tmp_path / "dcnv2_ebc0_sparse_200.model" / "emb_vector.table0.rank0")
    else:
        assert (model._ebc[0]["train"].table != before).any()
        assert os.path.exists(tmp_path / "dcnv2_ebc0_sparse_200.model" / "emb_vector.rank0")
    model.graph_to_json(str(tmp_path / "graph.json"))


@pytest.mark.parametrize("name", ["dcn", "dlrm"])
def test_load_reference_format_checkpoint(tmp_path, name):
    """the fixture under tests/golden/ckpt is in the reference's formats (verified with the
    reference's own loader, tests/test_ckpt_format_cpu.py): load_dense_weights /
    load_sparse_weights must bring a fresh model to exactly those weights, and saving again must
    reproduce the files byte for byte"""
    import sys
    here = os.path.dirname(os.path.abspath(__file__))
    sys.path.insert(0, os.path.join(here, "golden"))
    import make_ckpt_fixture as fx
    ck = os.path.join(here, "golden", "ckpt")
    p = fx.gen(str(tmp_path / "data"))
    m = getattr(fx, name)(p)
    m.compile()
    m.load_dense_weights(os.path.join(ck, f"{name}_dense_5.model"))
    m.load_sparse_weights([os.path.join(ck, f"{name}0_sparse_5.model")])
    want = np.load(os.path.join(ck, f"{name}_truth.npz"))
    got = fx.truth(m)
    for k in want.files:
        if k.startswith("emb_"):
            continue
        assert (got[k] == want[k]).all(), k
    o, w = np.argsort(got["emb_keys"]), np.argsort(want["emb_keys"])
    assert (got["emb_keys"][o] == want["emb_keys"][w]).all()
    assert (got["emb_vectors"][o] == want["emb_vectors"][w]).all()
    m.save_params_to_files(str(tmp_path / name), 5)
    a = np.fromfile(tmp_path / f"{name}_dense_5.model", dtype="<f4")
    b = np.fromfile(os.path.join(ck, f"{name}_dense_5.model"), dtype="<f4")
    assert (a == b).all()
    m.graph_to_json(str(tmp_path / "g.json"))
    import json
    assert json.load(open(tmp_path / "g.json")) == json.load(open(os.path.join(ck, f"{name}.json")))


def _ebc_model_worker(rank, world, port, folder, ret, overlap=False, iters=200):
    import torch.distributed as dist
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), LOCAL_RANK="0")
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        import hugectr_amd.hugectr as hugectr
        from hugectr_amd import sharding
        hot = [1, 2, 1, 2, 6, 1, 1, 1, 1, 7, 3, 8, 1, 6, 9, 5, 1, 1, 1, 12, 10, 7, 4, 3, 1, 1]
        solver = hugectr.CreateSolver(batchsize=256, batchsize_eval=256, lr=0.05, vvgpu=[[0, 1]],
                                      i64_input_key=True, max_eval_batches=1,
                                      use_embedding_collection=True,
                                      train_intra_iteration_overlap=overlap,
                                      train_inter_iteration_overlap=overlap)
        reader = hugectr.DataReaderParams(
            data_reader_type=hugectr.DataReaderType_t.Parquet,
            source=[os.path.join(folder, "train", "_file_list.txt")],
            eval_source=os.path.join(folder, "val", "_file_list.txt"), slot_size_array=SIZES,
            check_type=hugectr.Check_t.Non)
        opt = hugectr.CreateOptimizer(optimizer_type=hugectr.Optimizer_t.AdaGrad,
                                      update_type=hugectr.Update_t.Global, initial_accu_value=0.0)
        model = hugectr.Model(solver, reader, opt)
        model.add(hugectr.Input(label_dim=1, label_name="label", dense_dim=13, dense_name="dense",
                                data_reader_sparse_param_array=[
                                    hugectr.DataReaderSparseParam(f"data{i}", hot[i], True, 1)
                                    for i in range(26)]))
        tables = [hugectr.EmbeddingTableConfig(name=str(i), max_vocabulary_size=SIZES[i], ev_size=16)
                  for i in range(26)]
        # tables under 78 rows (5e-6 GB at ev 16, fp32) are replicated: ("dp", [...]) in the plan
        args = sharding.mi355x_args(sharding_plan="auto", ev_size=16, num_gpus_per_node=2,
                                    dp_sharding_threshold=5e-6)
        sm, ss = sharding.generate_plan(SIZES, hot, 1, 2, args, False)
        assert dict(ss)["dp"] == [str(i) for i, v in enumerate(SIZES) if v < 78]
        ebc = hugectr.EmbeddingCollectionConfig()
        ebc.embedding_lookup(table_config=tables, bottom_name=[f"data{i}" for i in range(26)],
                             top_name="sparse_embedding",
                             combiner=["concat" if h == 1 else "sum" for h in hot])
        ebc.shard(shard_matrix=sm, shard_strategy=ss)
        model.add(ebc)
        D, T = hugectr.DenseLayer, hugectr.Layer_t
        # (the reshape of R/test/embedding_collection_test/dgx_a100_one_hot.py:309-315)
        model.add(D(layer_type=T.Reshape, bottom_names=["sparse_embedding"],
                    top_names=["sparse_embedding1"], shape=[-1, 26, 16]))
        model.add(D(layer_type=T.MLP, bottom_names=["dense"], top_names=["mlp1"], num_outputs=[32, 16],
                    act_type=hugectr.Activation_t.Relu))
        model.add(D(layer_type=T.Interaction, bottom_names=["mlp1", "sparse_embedding1"],
                    top_names=["interaction1"]))
        model.add(D(layer_type=T.MLP, bottom_names=["interaction1"], top_names=["mlp2"],
                    num_outputs=[64, 1],
                    activations=[hugectr.Activation_t.Relu, hugectr.Activation_t.Non]))
        model.add(D(layer_type=T.BinaryCrossEntropyLoss, bottom_names=["mlp2", "label"],
                    top_names=["loss"]))
        model.compile()
        from hugectr_amd.embedding_collection import DataParallelCollection
        dp = [rt["train"] for rt in model._ebc if isinstance(rt["train"], DataParallelCollection)]
        mp_ = [rt["train"] for rt in model._ebc if not isinstance(rt["train"], DataParallelCollection)]
        assert len(dp) == 1 and len(mp_) == 1 and dp[0].L == sum(v < 78 for v in SIZES)
        e = mp_[0]
        owned = [t for t in range(len(e.tables)) if rank in e.owners[t]]
        assert owned and len(owned) < len(e.tables)   # the planner split the big tables over the ranks
        before = e.table.clone()
        dp_before = dp[0].table.clone()
        model.train()
        first = model.get_current_loss()
        model.fit(max_iter=iters, display=0, eval_interval=0, snapshot=0)
        last = model.get_current_loss()
        assert iters < 200 or last < min(first, 0.64), (first, last)
        assert (e.table != before).any() and (dp[0].table != dp_before).any()
        # the replicated tables stay identical on both ranks
        rep = dp[0].table.detach().flatten().cpu()
        both = [torch.empty_like(rep) for _ in range(world)]
        dist.all_gather(both, rep)
        assert torch.equal(both[0], both[1])
        # data-parallel dense weights stay identical on both ranks
        flat = torch.cat([q.detach().flatten().float() for q in model._dense_params]).cpu()
        both = [torch.empty_like(flat) for _ in range(world)]
        dist.all_gather(both, flat)
        assert torch.equal(both[0], both[1])
        ret[(rank, "state")] = (last, e.table.detach().cpu().numpy(), dp[0].table.detach().cpu().numpy(),
                                flat.numpy())
        ret[rank] = "ok"
    except Exception as ex:
        import traceback
        ret[rank] = "".join(traceback.format_exception(type(ex), ex, ex.__traceback__))
    finally:
        dist.destroy_process_group()


def test_embedding_collection_model_overlap_is_bit_equal(tmp_path):
    """the same 2-rank embedding_collection model with train_intra/inter_iteration_overlap on: the
    pooled vectors' all-to-all runs asynchronously under the bottom MLP, the gradients' starts from
    inside backward -- losses, model-parallel and replicated tables and the dense weights must
    equal the blocking schedule's bit for bit"""
    import hugectr_amd.hugectr as hugectr
    import torch.multiprocessing as mp
    hot = [1, 2, 1, 2, 6, 1, 1, 1, 1, 7, 3, 8, 1, 6, 9, 5, 1, 1, 1, 12, 10, 7, 4, 3, 1, 1]
    _gen(tmp_path, hugectr, n_train=4096, n_eval=512, nnz=hot)
    ctx = mp.get_context("spawn")
    states = {}
    for k, overlap in enumerate((False, True)):
        ret = ctx.Manager().dict()
        port = 29500 + os.getpid() % 2000 + 23 + k
        procs = [ctx.Process(target=_ebc_model_worker,
                             args=(r, 2, port, str(tmp_path), ret, overlap, 12)) for r in range(2)]
        for p in procs:
            p.start()
        for p in procs:
            p.join(600)
        for r in range(2):
            assert ret.get(r) == "ok", ret.get(r)
        states[overlap] = [ret[(r, "state")] for r in range(2)]
    for r in range(2):
        a, b = states[False][r], states[True][r]
        assert a[0] == b[0]
        for x, y in zip(a[1:], b[1:]):
            assert (x == y).all()


def test_embedding_collection_model_two_ranks_on_one_gpu(tmp_path):
    """the N > 1 path of the embedding_collection model (tables placed by the planner, key routing
    on the replicated batch, all-to-all of the pooled vectors, data-parallel dense tower): 2
    processes over gloo on this one GPU"""
    import hugectr_amd.hugectr as hugectr
    import torch.multiprocessing as mp
    hot = [1, 2, 1, 2, 6, 1, 1, 1, 1, 7, 3, 8, 1, 6, 9, 5, 1, 1, 1, 12, 10, 7, 4, 3, 1, 1]
    _gen(tmp_path, hugectr, n_train=4096, n_eval=512, nnz=hot)
    ctx = mp.get_context("spawn")
    ret = ctx.Manager().dict()
    port = 29500 + os.getpid() % 2000
    procs = [ctx.Process(target=_ebc_model_worker, args=(r, 2, port, str(tmp_path), ret))
             for r in range(2)]
    for p in procs:
        p.start()
    for p in procs:
        p.join(600)
    for r in range(2):
        if ret.get(r) != "ok":
            print(f"--- rank {r} ---\n{ret.get(r)}")
    assert ret.get(0) == "ok" and ret.get(1) == "ok"


def test_wdl_embedding_collection_mixed_vector_sizes(tmp_path):
    """Wide & Deep on ONE embedding_collection config that mixes vector sizes (wide tables ev = 1,
    deep tables ev = 16, BASELINE config 5's model family): dynamic hash tables
    (max_vocabulary_size = -1), per-lookup top names + Concat as in
    R/samples/ftrl/dlrm_train_ftrl.py:222-245"""
    import hugectr_amd.hugectr as hugectr
    hot = [1, 2] + [1] * 24
    p = _gen(tmp_path, hugectr, n_train=4096, n_eval=512, nnz=hot)
    solver = hugectr.CreateSolver(batchsize=512, batchsize_eval=512, lr=0.05, vvgpu=[[0]],
                                  i64_input_key=True, max_eval_batches=1,
                                  use_embedding_collection=True)
    reader = hugectr.DataReaderParams(data_reader_type=hugectr.DataReaderType_t.Parquet,
                                      source=[p.source], eval_source=p.eval_source,
                                      slot_size_array=SIZES, check_type=hugectr.Check_t.Non)
    opt = hugectr.CreateOptimizer(optimizer_type=hugectr.Optimizer_t.AdaGrad,
                                  update_type=hugectr.Update_t.Global)
    model = hugectr.Model(solver, reader, opt)
    model.add(hugectr.Input(label_dim=1, label_name="label", dense_dim=13, dense_name="dense",
                            data_reader_sparse_param_array=[
                                hugectr.DataReaderSparseParam(f"data{i}", hot[i], True, 1)
                                for i in range(26)]))
    wide, deep = [0, 1], list(range(2, 10))
    ebc = hugectr.EmbeddingCollectionConfig()
    for i in wide:
        ebc.embedding_lookup(table_config=hugectr.EmbeddingTableConfig(f"w{i}", -1, 1),
                             bottom_name=f"data{i}", top_name=f"wide{i}", combiner="sum")
    for i in deep:
        ebc.embedding_lookup(table_config=hugectr.EmbeddingTableConfig(f"d{i}", -1, 16),
                             bottom_name=f"data{i}", top_name=f"deep{i}", combiner="sum")
    ebc.shard(shard_matrix=[[f"w{i}" for i in wide] + [f"d{i}" for i in deep]],
              shard_strategy=[("mp", [f"w{i}" for i in wide] + [f"d{i}" for i in deep])])
    model.add(ebc)
    D, T = hugectr.DenseLayer, hugectr.Layer_t
    model.add(D(layer_type=T.Concat, bottom_names=[f"deep{i}" for i in deep] + ["dense"],
                top_names=["concat1"]))
    model.add(D(layer_type=T.MLP, bottom_names=["concat1"], top_names=["mlp1"], num_outputs=[64, 1],
                activations=[hugectr.Activation_t.Relu, hugectr.Activation_t.Non]))
    model.add(D(layer_type=T.Concat, bottom_names=[f"wide{i}" for i in wide], top_names=["wide"]))
    model.add(D(layer_type=T.ReduceSum, bottom_names=["wide"], top_names=["wide_sum"], axis=1))
    model.add(D(layer_type=T.Add, bottom_names=["mlp1", "wide_sum"], top_names=["add1"]))
    model.add(D(layer_type=T.BinaryCrossEntropyLoss, bottom_names=["add1", "label"],
                top_names=["loss"]))
    model.compile()
    assert len(model._ebc) == 2 and {rt["train"].ev for rt in model._ebc} == {1, 16}
    model.train()
    first = model.get_current_loss()
    model.fit(max_iter=200, display=0, eval_interval=0, snapshot=0)
    assert model.get_current_loss() < min(first, 0.55)
    # the wide table of feature C1 (which decides the label) has learnt a per-key bias
    wide_rt = [rt for rt in model._ebc if rt["train"].ev == 1][0]["train"]
    k, v = wide_rt.det.export(wide_rt.class_of_table[0])
    assert k.numel() > 10 and float(v.abs().max()) > 0.1


def test_training_callbacks_and_early_stop(tmp_path):
    """fit() drives hugectr.TrainingCallback as the reference does (model.cpp:869-994):
    start -> (eval start, eval end(results))* -> end; on_eval_end returning True stops training"""
    import sys
    import hugectr_amd.hugectr as hugectr
    here = os.path.dirname(os.path.abspath(__file__))
    sys.path.insert(0, os.path.join(here, "golden"))
    import make_ckpt_fixture as fx
    log = []

    class CB(hugectr.TrainingCallback):
        def on_training_start(self):
            log.append("start")

        def on_training_end(self, it):
            log.append(("end", it))

        def on_eval_start(self, it):
            log.append(("eval_start", it))
            return False

        def on_eval_end(self, it, results):
            log.append(("eval_end", it, sorted(results)))
            return it >= 19            # stop at the second evaluation

    m = fx.dlrm(fx.gen(str(tmp_path / "d")))
    m.solver.training_callbacks = [CB()]
    m.compile()
    m.fit(max_iter=100, display=0, eval_interval=10, snapshot=0)
    assert log == ["start", ("eval_start", 9), ("eval_end", 9, ["AUC", "AverageLoss"]),
                   ("eval_start", 19), ("eval_end", 19, ["AUC", "AverageLoss"]), ("end", 19)]
    assert m._iter == 20


def _legacy_model_worker(rank, world, port, folder, kind, ret):
    import torch.distributed as dist
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), LOCAL_RANK="0")
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        import hugectr_amd.hugectr as hugectr
        solver = hugectr.CreateSolver(batchsize=256, batchsize_eval=256, lr=0.01, vvgpu=[[0, 1]],
                                      i64_input_key=True, max_eval_batches=1)
        reader = hugectr.DataReaderParams(
            data_reader_type=hugectr.DataReaderType_t.Parquet,
            source=[os.path.join(folder, "train", "_file_list.txt")],
            eval_source=os.path.join(folder, "val", "_file_list.txt"), slot_size_array=SIZES,
            check_type=hugectr.Check_t.Non)
        opt = hugectr.CreateOptimizer(optimizer_type=hugectr.Optimizer_t.Adam,
                                      update_type=hugectr.Update_t.Local)
        model = hugectr.Model(solver, reader, opt)
        model.add(hugectr.Input(label_dim=1, label_name="label", dense_dim=13, dense_name="dense",
                                data_reader_sparse_param_array=[
                                    hugectr.DataReaderSparseParam("data1", 1, True, 26)]))
        D, T = hugectr.DenseLayer, hugectr.Layer_t
        localized = kind == "localized"
        model.add(hugectr.SparseEmbedding(
            embedding_type=(hugectr.Embedding_t.LocalizedSlotSparseEmbeddingHash if localized
                            else hugectr.Embedding_t.DistributedSlotSparseEmbeddingHash),
            workspace_size_per_gpu_in_mb=4, slot_size_array=SIZES if localized else [],
            embedding_vec_size=16, combiner="sum", sparse_embedding_name="emb", bottom_name="data1",
            optimizer=opt))
        model.add(D(layer_type=T.MLP, bottom_names=["dense"], top_names=["mlp1"], num_outputs=[32, 16],
                    act_type=hugectr.Activation_t.Relu))
        model.add(D(layer_type=T.Interaction, bottom_names=["mlp1", "emb"], top_names=["inter"]))
        model.add(D(layer_type=T.MLP, bottom_names=["inter"], top_names=["mlp2"], num_outputs=[64, 1],
                    activations=[hugectr.Activation_t.Relu, hugectr.Activation_t.Non]))
        model.add(D(layer_type=T.BinaryCrossEntropyLoss, bottom_names=["mlp2", "label"],
                    top_names=["loss"]))
        model.compile()
        model.train()
        first = model.get_current_loss()
        # (the distributed exchange is staged through the host here: fewer iterations)
        model.fit(max_iter=200 if localized else 60, display=0, eval_interval=50, snapshot=0)
        last = model.get_current_loss()
        assert last < (min(first, 0.6) if localized else first), (first, last)
        flat = torch.cat([q.detach().flatten().float() for q in model._dense_params]).cpu()
        both = [torch.empty_like(flat) for _ in range(world)]
        dist.all_gather(both, flat)
        assert torch.equal(both[0], both[1])
        ret[rank] = "ok"
    except Exception as ex:
        import traceback
        ret[rank] = "".join(traceback.format_exception(type(ex), ex, ex.__traceback__))
    finally:
        dist.destroy_process_group()


@pytest.mark.parametrize("kind", ["localized", "distributed"])
def test_legacy_embedding_model_two_ranks_on_one_gpu(tmp_path, kind):
    """hugectr.Model with a Localized / Distributed SparseEmbedding on 2 processes (gloo, this one
    GPU): slot- / key-sharded tables, the exchange of the reference (all-to-all + reorder, or
    reduce-scatter / all-gather), data-parallel dense tower"""
    import hugectr_amd.hugectr as hugectr
    import torch.multiprocessing as mp
    _gen(tmp_path, hugectr, n_train=4096, n_eval=512)
    ctx = mp.get_context("spawn")
    ret = ctx.Manager().dict()
    port = 29500 + os.getpid() % 2000 + (7 if kind == "localized" else 11)
    procs = [ctx.Process(target=_legacy_model_worker, args=(r, 2, port, str(tmp_path), kind, ret))
             for r in range(2)]
    for p in procs:
        p.start()
    for p in procs:
        p.join(600)
    for r in range(2):
        if ret.get(r) != "ok":
            print(f"--- rank {r} ---\n{ret.get(r)}")
    assert ret.get(0) == "ok" and ret.get(1) == "ok"


def test_freeze_and_optimizer_state_files(tmp_path):
    """IEmbedding::freeze / Model.freeze_embedding / freeze_dense, and the sparse optimizer state
    file (<prefix><i>_opt_sparse_<iter>.model: m then v, raw [max_vocabulary_size_per_gpu, D]
    arrays -- opt_states_functor.cu:24-128)"""
    import sys
    import hugectr_amd.hugectr as hugectr
    here = os.path.dirname(os.path.abspath(__file__))
    sys.path.insert(0, os.path.join(here, "golden"))
    import make_ckpt_fixture as fx
    m = fx.dcn(fx.gen(str(tmp_path / "d")))          # Adam, Distributed embedding, vec 4
    m.compile()
    m.fit(max_iter=3, display=0, eval_interval=0, snapshot=0)
    h = list(m._emb.values())[0][2]
    dense0 = [q.detach().clone() for q in m._dense_params]
    table0 = h.table().clone()
    m.freeze_embedding()
    assert not h.is_trainable()
    m.fit(max_iter=2, display=0, eval_interval=0, snapshot=0)
    assert torch.equal(h.table(), table0)                               # embedding stood still
    assert any(not torch.equal(a, b) for a, b in zip(dense0, m._dense_params))
    m.unfreeze_embedding("sparse_embedding1")
    m.freeze_dense()
    dense1 = [q.detach().clone() for q in m._dense_params]
    m.fit(max_iter=2, display=0, eval_interval=0, snapshot=0)
    assert all(torch.equal(a, b) for a, b in zip(dense1, m._dense_params))
    assert not torch.equal(h.table(), table0)
    m.save_params_to_files(str(tmp_path / "ck"), 7)
    f = tmp_path / "ck0_opt_sparse_7.model"
    V, D = h.get_max_vocabulary_size(), 4
    raw = np.fromfile(f, dtype="<f4")
    assert raw.size == 2 * V * D
    assert (raw[:V * D].reshape(V, D) == h.opt_state(0).cpu().numpy()).all()      # m first
    assert (raw[V * D:].reshape(V, D) == h.opt_state(1).cpu().numpy()).all()      # then v
    s0 = h.opt_state(0).clone()
    h.opt_state(0).zero_()
    m.load_sparse_optimizer_states([str(f)])
    assert torch.equal(h.opt_state(0), s0)


def test_low_level_training_loop(tmp_path):
    """the loop of R/test/pybind_test/model_test.py:1209-1230: start_data_reading, the solver's
    learning-rate schedule (warm-up + decay) applied by hand, train / eval / get_eval_metrics"""
    import sys
    import hugectr_amd.hugectr as hugectr
    here = os.path.dirname(os.path.abspath(__file__))
    sys.path.insert(0, os.path.join(here, "golden"))
    import make_ckpt_fixture as fx
    m = fx.dlrm(fx.gen(str(tmp_path / "d")))
    m.solver.warmup_steps, m.solver.decay_start, m.solver.decay_steps = 5, 20, 10
    m.solver.end_lr = 0.001
    m.compile()
    m.start_data_reading()
    sch = m.get_learning_rate_scheduler()
    lrs = []
    for it in range(40):
        lr = sch.get_next()
        m.set_learning_rate(lr)
        lrs.append(lr)
        assert m.train()
        if it % 10 == 0 and it:
            m._eval_buf = []
            for _ in range(1):
                m.eval()
            names = [n for n, _ in m.get_eval_metrics()]
            assert names == ["AUC", "AverageLoss"]
    assert lrs[0] == pytest.approx(0.01 / 5) and lrs[4] == pytest.approx(0.01)
    assert lrs[19] == pytest.approx(0.01) and lrs[24] == pytest.approx(0.01 * 0.25)
    assert lrs[-1] == pytest.approx(0.001)
    h = list(m._emb.values())[0][2]
    assert np.isfinite(m.get_current_loss()) and h.get_vocabulary_size() > 0


# ---- N ranks == 1 rank (the embeddings' gradients must be shares of the GLOBAL-batch
# mean, loss.cu:242-249; VERDICT r1: distributed + mean divides by the global key count) ------------
_PAR_HOT = [3, 1, 2, 1, 4, 1, 1, 2, 1, 3, 1, 1, 1, 2, 1, 1, 1, 5, 1, 2, 1, 1, 2, 1, 1, 1]


def _parity_model(hugectr, folder, kind, combiner, world, mixed=False):
    solver = hugectr.CreateSolver(batchsize=256, batchsize_eval=256, lr=0.05,
                                  vvgpu=[list(range(world))], i64_input_key=True,
                                  max_eval_batches=1, use_mixed_precision=mixed,
                                  scaler=128.0 if mixed else 1.0)
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
                                hugectr.DataReaderSparseParam("data1", _PAR_HOT, False, 26)]))
    D, T = hugectr.DenseLayer, hugectr.Layer_t
    localized = kind == "localized"
    model.add(hugectr.SparseEmbedding(
        embedding_type=(hugectr.Embedding_t.LocalizedSlotSparseEmbeddingHash if localized
                        else hugectr.Embedding_t.DistributedSlotSparseEmbeddingHash),
        workspace_size_per_gpu_in_mb=4, slot_size_array=SIZES if localized else [],
        embedding_vec_size=16, combiner=combiner, sparse_embedding_name="emb", bottom_name="data1",
        optimizer=opt))
    model.add(D(layer_type=T.Reshape, bottom_names=["emb"], top_names=["flat"], leading_dim=26 * 16))
    model.add(D(layer_type=T.Concat, bottom_names=["flat", "dense"], top_names=["cat"]))
    model.add(D(layer_type=T.MLP, bottom_names=["cat"], top_names=["mlp"], num_outputs=[64, 32, 1],
                activations=[hugectr.Activation_t.Relu, hugectr.Activation_t.Relu,
                             hugectr.Activation_t.Non]))
    model.add(D(layer_type=T.BinaryCrossEntropyLoss, bottom_names=["mlp", "label"],
                top_names=["loss"]))
    model.compile()
    model.load_sparse_weights([os.path.join(folder, "init_sparse")])
    return model


def _parity_run(model, steps):
    losses = []
    for _ in range(steps):
        assert model.train()
        losses.append(model.get_current_loss())
    h = list(model._emb.values())[0][2]
    k, _, v = h.dump_parameters()
    dense = torch.cat([q.detach().flatten().float() for q in model._dense_params]).cpu()
    return losses, k.cpu().numpy(), v.cpu().numpy(), dense.numpy()


def _parity_worker(rank, world, port, folder, kind, combiner, steps, ret):
    import torch.distributed as dist
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), LOCAL_RANK="0")
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        import hugectr_amd.hugectr as hugectr
        model = _parity_model(hugectr, folder, kind, combiner, world)
        losses, k, v, dense = _parity_run(model, steps)
        # the merged checkpoint directory written by all ranks loads back into the same tables
        model.save_params_to_files(os.path.join(folder, "ck_"), 3)
        ret[rank] = ("ok", losses, k, v, dense)
    except Exception as ex:
        import traceback
        ret[rank] = ("".join(traceback.format_exception(type(ex), ex, ex.__traceback__)),)
    finally:
        dist.destroy_process_group()


@pytest.mark.parametrize("kind,combiner", [("localized", "sum"), ("localized", "mean"),
                                           ("distributed", "sum"), ("distributed", "mean")])
def test_two_ranks_train_like_one_rank(tmp_path, kind, combiner):
    """The same model, initial weights and global batches on 1 rank and on 2 ranks (gloo, this one
    GPU): after 6 SGD steps the losses, every embedding vector (as a key -> vector map) and the
    dense weights agree to fp32 rounding.  Catches gradients that are a share of the per-GPU mean
    instead of the global one, and a mean combiner that divides by a rank's key count."""
    import hugectr_amd.hugectr as hugectr
    import torch.multiprocessing as mp
    from numpy.testing import assert_allclose
    _gen(tmp_path, hugectr, n_train=2048, n_eval=512, nnz=_PAR_HOT)
    rng = np.random.default_rng(3)
    V = sum(SIZES)
    d = tmp_path / "init_sparse"
    d.mkdir()
    np.arange(V, dtype="<i8").tofile(d / "key")
    np.repeat(np.arange(26), SIZES).astype("<u8").tofile(d / "slot_id")
    (rng.standard_normal((V, 16)) * 0.1).astype("<f4").tofile(d / "emb_vector")
    steps = 6
    one = _parity_run(_parity_model(hugectr, str(tmp_path), kind, combiner, 1), steps)
    ctx = mp.get_context("spawn")
    ret = ctx.Manager().dict()
    port = 29500 + os.getpid() % 2000 + 13
    procs = [ctx.Process(target=_parity_worker,
                         args=(r, 2, port, str(tmp_path), kind, combiner, steps, ret))
             for r in range(2)]
    for p in procs:
        p.start()
    for p in procs:
        p.join(600)
    for r in range(2):
        assert ret.get(r) is not None and ret[r][0] == "ok", ret.get(r)
    # rank-local losses average to the global-batch loss
    two_loss = (np.array(ret[0][1]) + np.array(ret[1][1])) / 2
    assert_allclose(two_loss, np.array(one[0]), rtol=2e-5)
    assert_allclose(ret[0][4], one[3], rtol=2e-4, atol=2e-6)      # dense weights
    assert_allclose(ret[1][4], one[3], rtol=2e-4, atol=2e-6)
    k2 = np.concatenate([ret[0][2], ret[1][2]])
    v2 = np.concatenate([ret[0][3], ret[1][3]])
    assert len(np.unique(k2)) == k2.size == one[1].size
    o1, o2 = np.argsort(one[1]), np.argsort(k2)
    assert (one[1][o1] == k2[o2]).all()
    assert_allclose(v2[o2], one[2][o1], rtol=2e-4, atol=2e-6)
    moved = np.abs(one[2][o1] - np.fromfile(d / "emb_vector", "<f4").reshape(V, 16)).max()
    assert moved > 1e-4, "the embedding did not train"
    # merged multi-rank checkpoint: one directory, every key once, same vectors
    ck = tmp_path / "ck_0_sparse_3.model"
    kk = np.fromfile(ck / "key", "<i8")
    vv = np.fromfile(ck / "emb_vector", "<f4").reshape(-1, 16)
    assert kk.size == k2.size and (np.sort(kk) == k2[o2]).all()
    assert (vv[np.argsort(kk)] == v2[o2]).all()


# ---- the N > 1 training step of the product: solver.train_intra / inter_iteration_overlap and the
# exchange payload (R/HugeCTR/src/pybind/model_pipeline.cpp:299-346) ------------
def _overlap_model(hugectr, folder, world, overlap, mixed):
    solver = hugectr.CreateSolver(batchsize=256, batchsize_eval=256, lr=0.05,
                                  vvgpu=[list(range(world))], i64_input_key=True,
                                  max_eval_batches=1, use_mixed_precision=mixed,
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
    D, T, A = hugectr.DenseLayer, hugectr.Layer_t, hugectr.Activation_t
    model.add(hugectr.SparseEmbedding(
        embedding_type=hugectr.Embedding_t.LocalizedSlotSparseEmbeddingHash,
        slot_size_array=SIZES, embedding_vec_size=32, combiner="sum",
        sparse_embedding_name="emb", bottom_name="data1", optimizer=opt))
    model.add(D(layer_type=T.MLP, bottom_names=["dense"], top_names=["mlp1"], num_outputs=[64, 32],
                act_type=A.Relu))
    model.add(D(layer_type=T.Interaction, bottom_names=["mlp1", "emb"], top_names=["inter"]))
    model.add(D(layer_type=T.MLP, bottom_names=["inter"], top_names=["mlp2"],
                num_outputs=[128, 64, 1], activations=[A.Relu, A.Relu, A.Non]))
    model.add(D(layer_type=T.BinaryCrossEntropyLoss, bottom_names=["mlp2", "label"],
                top_names=["loss"]))
    model.compile()
    model.load_sparse_weights([os.path.join(folder, "init_sparse")])
    return model


def _overlap_worker(rank, world, port, folder, overlap, exchange, mixed, steps, ret):
    import torch.distributed as dist
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), LOCAL_RANK="0")
    if exchange:
        os.environ["HCTR_EXCHANGE"] = exchange
    else:
        os.environ.pop("HCTR_EXCHANGE", None)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        import hugectr_amd.hugectr as hugectr
        torch.manual_seed(5)
        model = _overlap_model(hugectr, folder, world, overlap, mixed)
        losses, k, v, dense = _parity_run(model, steps)
        model._eval_buf = []
        model.eval()  # (drains an index stage that ran ahead)
        auc = dict(model.get_eval_metrics())["AUC"]
        rep = model.exchange_report()["emb"]
        ret[rank] = ("ok", losses, k, v, dense, rep, auc)
    except Exception as ex:
        import traceback
        ret[rank] = ("".join(traceback.format_exception(type(ex), ex, ex.__traceback__)),)
    finally:
        dist.destroy_process_group()


def _overlap_run(tmp_path, overlap, exchange, mixed, steps, salt):
    import torch.multiprocessing as mp
    ctx = mp.get_context("spawn")
    ret = ctx.Manager().dict()
    port = 29500 + os.getpid() % 2000 + 17 + salt
    procs = [ctx.Process(target=_overlap_worker,
                         args=(r, 2, port, str(tmp_path), overlap, exchange, mixed, steps, ret))
             for r in range(2)]
    for p in procs:
        p.start()
    for p in procs:
        p.join(600)
    for r in range(2):
        assert ret.get(r) is not None and ret[r][0] == "ok", ret.get(r)
    return [ret[0], ret[1]]


@pytest.mark.parametrize("mixed", [False, True])
def test_two_ranks_overlap_schedule(tmp_path, mixed):
    """2 ranks (gloo, this one GPU), the DLRM graph: (a) train_intra/inter_iteration_overlap off =
    blocking collectives in line; (b) overlap on with the rows payload (asynchronous all-to-all
    under the bottom MLP, gradient all-to-all from inside backward) must equal (a) BIT FOR BIT;
    (c) the unique-row payload (distinct rows once per destination, per-row gradient sums, next
    batch's index stage on a side stream; with 16-bit vectors the Interaction reads the rows
    through the index table) agrees to fp32 rounding; (d) HCTR_EXCHANGE=auto runs its selection
    over real training steps, settles on one payload on both ranks and trains the same model."""
    import hugectr_amd.hugectr as hugectr
    from numpy.testing import assert_allclose
    _gen(tmp_path, hugectr, n_train=8192, n_eval=512)
    rng = np.random.default_rng(3)
    V = sum(SIZES)
    d = tmp_path / "init_sparse"
    d.mkdir()
    np.arange(V, dtype="<i8").tofile(d / "key")
    np.repeat(np.arange(26), SIZES).astype("<u8").tofile(d / "slot_id")
    (rng.standard_normal((V, 32)) * 0.1).astype("<f4").tofile(d / "emb_vector")
    steps = 20
    a = _overlap_run(tmp_path, False, None, mixed, steps, 0)
    b = _overlap_run(tmp_path, True, "rows", mixed, steps, 1)
    for r in range(2):
        assert a[r][5]["payload"] == "rows" and not a[r][5]["intra_iteration_overlap"]
        assert b[r][5]["payload"] == "rows" and b[r][5]["intra_iteration_overlap"]
        assert a[r][1] == b[r][1], "losses differ between overlap off and on"
        oa, ob = np.argsort(a[r][2]), np.argsort(b[r][2])  # (tables compare as key -> vector maps)
        assert (a[r][2][oa] == b[r][2][ob]).all() and (a[r][3][oa] == b[r][3][ob]).all()
        assert (a[r][4] == b[r][4]).all()
    tol = dict(rtol=2e-2, atol=2e-3) if mixed else dict(rtol=2e-4, atol=2e-6)
    for exchange, salt in (("unique", 2), ("auto", 3)):
        c = _overlap_run(tmp_path, True, exchange, mixed, steps, salt)
        assert c[0][5]["payload"] == c[1][5]["payload"]
        if exchange == "unique":
            assert c[0][5]["payload"] == "unique" and c[0][5]["distinct_rows_out"] > 0
        else:
            assert c[0][5]["payload"] in ("rows", "unique")
            assert c[0][5]["selection_ms_per_step"] is not None
        for r in range(2):
            assert_allclose(c[r][1], a[r][1], rtol=5e-3 if mixed else 2e-5)
            o1, o2 = np.argsort(a[r][2]), np.argsort(c[r][2])
            assert (a[r][2][o1] == c[r][2][o2]).all()
            assert_allclose(c[r][3][o2], a[r][3][o1], **tol)
            assert_allclose(c[r][4], a[r][4], **tol)
            assert abs(c[r][6] - a[r][6]) < 2e-2


# ---- solver.use_cuda_graph: the dense tower replayed from a HIP graph ------------------------------
def _graph_model(hugectr, folder, mixed, opt_type):
    solver = hugectr.CreateSolver(batchsize=512, batchsize_eval=512, lr=0.02, vvgpu=[[0]],
                                  i64_input_key=True, max_eval_batches=1, use_mixed_precision=mixed,
                                  scaler=128.0 if mixed else 1.0)
    reader = hugectr.DataReaderParams(
        data_reader_type=hugectr.DataReaderType_t.Parquet,
        source=[os.path.join(folder, "train", "_file_list.txt")],
        eval_source=os.path.join(folder, "val", "_file_list.txt"), slot_size_array=SIZES,
        check_type=hugectr.Check_t.Non)
    opt = hugectr.CreateOptimizer(optimizer_type=opt_type, update_type=hugectr.Update_t.Local,
                                  atomic_update=False)
    model = hugectr.Model(solver, reader, opt)
    model.add(hugectr.Input(label_dim=1, label_name="label", dense_dim=13, dense_name="dense",
                            data_reader_sparse_param_array=[
                                hugectr.DataReaderSparseParam("data1", 1, True, 26)]))
    D, T, A = hugectr.DenseLayer, hugectr.Layer_t, hugectr.Activation_t
    model.add(hugectr.SparseEmbedding(
        embedding_type=hugectr.Embedding_t.LocalizedSlotSparseEmbeddingHash,
        slot_size_array=SIZES, embedding_vec_size=32, combiner="sum",
        sparse_embedding_name="emb", bottom_name="data1", optimizer=opt))
    model.add(D(layer_type=T.MLP, bottom_names=["dense"], top_names=["mlp1"], num_outputs=[64, 32],
                act_type=A.Relu))
    model.add(D(layer_type=T.Interaction, bottom_names=["mlp1", "emb"], top_names=["inter"]))
    model.add(D(layer_type=T.MLP, bottom_names=["inter"], top_names=["mlp2"],
                num_outputs=[128, 64, 1], activations=[A.Relu, A.Relu, A.Non]))
    model.add(D(layer_type=T.BinaryCrossEntropyLoss, bottom_names=["mlp2", "label"],
                top_names=["loss"]))
    model.compile()
    model.load_sparse_weights([os.path.join(folder, "init_sparse")])
    return model


@pytest.mark.parametrize("mixed,opt_name", [(True, "SGD"), (False, "Adam"), (False, "SGD")])
def test_hip_graph_replay_trains_like_eager_launches(tmp_path, monkeypatch, mixed, opt_name):
    """use_cuda_graph (default on, batch <= 8192, one GPU): after three eager steps the dense
    tower's forward + loss + backward + optimizer step are captured once and replayed; the
    embedding's index stage / gather (fused into the interaction in mixed precision) and sparse
    update stay eager.  12 steps with the graph and 12 without must give the same losses, tables and
    dense weights (same kernels in the same order: to rounding of the library's GEMM choice)."""
    import hugectr_amd.hugectr as hugectr
    from numpy.testing import assert_allclose
    _gen(tmp_path, hugectr, n_train=8192, n_eval=512)
    rng = np.random.default_rng(3)
    V = sum(SIZES)
    d = tmp_path / "init_sparse"
    d.mkdir()
    np.arange(V, dtype="<i8").tofile(d / "key")
    np.repeat(np.arange(26), SIZES).astype("<u8").tofile(d / "slot_id")
    (rng.standard_normal((V, 32)) * 0.1).astype("<f4").tofile(d / "emb_vector")
    res = {}
    for mode in ("0", "auto"):
        monkeypatch.setenv("HCTR_HIP_GRAPH", mode)
        torch.manual_seed(5)
        m = _graph_model(hugectr, str(tmp_path), mixed, getattr(hugectr.Optimizer_t, opt_name))
        losses, k, v, dense = _parity_run(m, 12)
        assert (m._graph is not None) == (mode == "auto")
        m._eval_buf = []
        m.eval()
        res[mode] = (losses, k, v, dense, dict(m.get_eval_metrics())["AUC"])
    a, b = res["0"], res["auto"]
    # (Adam: the captured step is torch's `capturable` form, whose arithmetic differs from the
    #  eager form in the last bit; 12 fast-moving steps amplify that to 1e-3)
    tol = dict(rtol=2e-3, atol=2e-4) if mixed else \
        dict(rtol=2e-2, atol=2e-3) if opt_name == "Adam" else dict(rtol=2e-5, atol=1e-6)
    assert_allclose(b[0][:5], a[0][:5], rtol=1e-3 if mixed else 1e-5)
    assert_allclose(b[0], a[0], rtol=tol["rtol"])
    oa, ob = np.argsort(a[1]), np.argsort(b[1])
    assert (a[1][oa] == b[1][ob]).all()
    assert_allclose(b[2][ob], a[2][oa], **tol)
    assert_allclose(b[3], a[3], **tol)
    assert abs(a[4] - b[4]) < 1e-2


def _wdl_two_task_model(hugectr, tmp_path, p, hot, two_tasks):
    solver = hugectr.CreateSolver(batchsize=512, batchsize_eval=512, lr=0.05, vvgpu=[[0]],
                                  i64_input_key=True, max_eval_batches=1, seed=11,
                                  use_mixed_precision=True, scaler=1024.0,
                                  use_embedding_collection=True)
    reader = hugectr.DataReaderParams(data_reader_type=hugectr.DataReaderType_t.Parquet,
                                      source=[p.source], eval_source=p.eval_source,
                                      slot_size_array=SIZES, check_type=hugectr.Check_t.Non)
    opt = hugectr.CreateOptimizer(optimizer_type=hugectr.Optimizer_t.AdaGrad,
                                  update_type=hugectr.Update_t.Global)
    model = hugectr.Model(solver, reader, opt)
    model.add(hugectr.Input(label_dim=1, label_name="label", dense_dim=13, dense_name="dense",
                            data_reader_sparse_param_array=[
                                hugectr.DataReaderSparseParam(f"data{i}", hot[i], True, 1)
                                for i in range(26)]))
    wide, deep = [0, 1], list(range(2, 10))
    ebc = hugectr.EmbeddingCollectionConfig()
    for i in wide:
        ebc.embedding_lookup(table_config=hugectr.EmbeddingTableConfig(f"w{i}", -1, 1),
                             bottom_name=f"data{i}", top_name=f"wide{i}", combiner="sum")
    for i in deep:
        ebc.embedding_lookup(table_config=hugectr.EmbeddingTableConfig(f"d{i}", -1, 16),
                             bottom_name=f"data{i}", top_name=f"deep{i}", combiner="sum")
    names = [f"w{i}" for i in wide] + [f"d{i}" for i in deep]
    ebc.shard(shard_matrix=[names], shard_strategy=[("mp", names)])
    model.add(ebc)
    D, T, A = hugectr.DenseLayer, hugectr.Layer_t, hugectr.Activation_t
    model.add(D(layer_type=T.Concat, bottom_names=[f"deep{i}" for i in deep], top_names=["emb"]))
    model.add(D(layer_type=T.Slice, bottom_names=["emb"], top_names=["inA", "inB"],
                ranges=[(0, 128), (0, 128)]))
    model.add(D(layer_type=T.MLP, bottom_names=["inA"], top_names=["mlpA"], num_outputs=[64, 1],
                activations=[A.Relu, A.Non]))
    model.add(D(layer_type=T.Concat, bottom_names=[f"wide{i}" for i in wide], top_names=["wide"]))
    model.add(D(layer_type=T.ReduceSum, bottom_names=["wide"], top_names=["wide_sum"], axis=1))
    model.add(D(layer_type=T.Add, bottom_names=["mlpA", "wide_sum"], top_names=["logitA"]))
    model.add(D(layer_type=T.BinaryCrossEntropyLoss, bottom_names=["logitA", "label"],
                top_names=["lossA"]))
    if two_tasks:  # a second tower on the same label: two BinaryCrossEntropyLoss layers
        model.add(D(layer_type=T.MLP, bottom_names=["inB"], top_names=["logitB"],
                    num_outputs=[32, 1], activations=[A.Relu, A.Non]))
        model.add(D(layer_type=T.BinaryCrossEntropyLoss, bottom_names=["logitB", "label"],
                    top_names=["lossB"]))
    model.compile()
    return model


@pytest.mark.parametrize("two_tasks", [False, True])
def test_hip_graph_replay_of_an_embedding_collection_model(tmp_path, monkeypatch, two_tasks):
    """use_cuda_graph on a model written with EmbeddingCollectionConfig (Wide & Deep on dynamic
    tables, mixed vector sizes, fp16 tower, one or two losses -- BASELINE configs[4]'s family):
    the collections' lookups and their backward + update stay eager launches around the replayed
    dense tower.  The same steps with and without the graph give the same losses and tables."""
    import hugectr_amd.hugectr as hugectr
    from numpy.testing import assert_allclose
    hot = [1, 2] + [1] * 24
    p = _gen(tmp_path, hugectr, n_train=4096, n_eval=512, nnz=hot)
    res = {}
    for mode in ("0", "auto"):
        monkeypatch.setenv("HCTR_HIP_GRAPH", mode)
        torch.manual_seed(5)
        m = _wdl_two_task_model(hugectr, tmp_path, p, hot, two_tasks)
        losses = []
        for _ in range(10):
            assert m.train()
            losses.append(m.get_current_loss())
        if os.environ.get("HCTR_EMU") != "1":  # (the host interpreter has no graphs: eager both times)
            assert (m._graph is not None) == (mode == "auto")
        tabs = {}
        for rt in m._ebc:
            e = rt["train"]
            for t in range(len(e.class_of_table)):
                k, v = e.det.export(e.class_of_table[t])
                o = torch.argsort(k)
                tabs[(e.ev, t)] = (k[o].cpu().numpy(), v[o].float().cpu().numpy())
        res[mode] = (np.array(losses), tabs)
    a, b = res["0"], res["auto"]
    assert np.isfinite(b[0]).all() and b[0][-1] < b[0][0]
    assert_allclose(b[0], a[0], rtol=2e-3)
    assert a[1].keys() == b[1].keys()
    for key in a[1]:
        assert (a[1][key][0] == b[1][key][0]).all()
        assert_allclose(b[1][key][1], a[1][key][1], rtol=5e-3, atol=5e-4)
