"""CPU: the hugectr-shaped Python surface (solver/optimizer defaults, enums) and the dataset
formats either side of the hot path (Parquet layout, Norm framing)."""
import json
import os

import numpy as np
import pytest
import torch


def test_create_solver_defaults_match_reference():
    # R/HugeCTR/include/pybind/solver_wrapper.hpp:127-150
    import hugectr_amd.hugectr as hugectr
    s = hugectr.CreateSolver()
    assert (s.lr, s.batchsize, s.batchsize_eval, s.max_eval_batches) == (0.001, 2048, 2048, 100)
    assert s.vvgpu == [[0]] and s.repeat_dataset and not s.use_mixed_precision
    assert s.scaler == 1.0 and not s.i64_input_key and not s.use_embedding_collection
    assert s.num_iterations_statistics == 20 and s.drop_incomplete_batch
    s = hugectr.CreateSolver(batchsize=16384, vvgpu=[[0]], i64_input_key=True, lr=0.01)
    assert s.batchsize == 16384 and s.i64_input_key
    with pytest.raises(RuntimeError):
        hugectr.CreateSolver(no_such_option=1)
    with pytest.raises(RuntimeError):  # mixed precision needs a supported loss scaler
        hugectr.CreateSolver(use_mixed_precision=True, scaler=3.0)


def test_create_optimizer_defaults_and_enums():
    # R/HugeCTR/include/pybind/optimizer_wrapper.hpp:35-40 ; common.hpp:82-94,145-149
    import hugectr_amd.hugectr as hugectr
    o = hugectr.CreateOptimizer()
    assert o.optimizer_type == hugectr.Optimizer_t.Adam and o.update_type == hugectr.Update_t.Global
    assert (o.beta1, o.beta2, o.epsilon, o.atomic_update) == (0.9, 0.999, 1e-7, True)
    assert [int(x) for x in (hugectr.Optimizer_t.Ftrl, hugectr.Optimizer_t.Adam,
                              hugectr.Optimizer_t.RMSProp, hugectr.Optimizer_t.AdaGrad,
                              hugectr.Optimizer_t.Nesterov, hugectr.Optimizer_t.MomentumSGD,
                              hugectr.Optimizer_t.SGD)] == list(range(7))
    assert int(hugectr.Embedding_t.DistributedSlotSparseEmbeddingHash) == 0
    assert int(hugectr.Embedding_t.LocalizedSlotSparseEmbeddingHash) == 1
    # workspace -> max_vocabulary_size_per_gpu (model.cpp:186-196): 267 MB, Adam (2 states), D=16
    from hugectr_amd.hugectr import _max_vocab_from_workspace
    assert _max_vocab_from_workspace(267, o, 16) == 267 * 2**20 // (3 * 4 * 16)


def test_parquet_dataset_roundtrip(tmp_path):
    import hugectr_amd.hugectr as hugectr
    from hugectr_amd import data
    sizes = [50, 7, 1000, 3]
    p = hugectr.tools.DataGeneratorParams(
        format=hugectr.DataReaderType_t.Parquet, label_dim=1, dense_dim=5, num_slot=4,
        i64_input_key=True, source=str(tmp_path / "train" / "_file_list.txt"),
        eval_source=str(tmp_path / "val" / "_file_list.txt"), slot_size_array=sizes,
        dist_type=hugectr.Distribution_t.PowerLaw, power_law_type=hugectr.PowerLaw_t.Short,
        num_files=2, eval_num_files=1, num_samples_per_file=256, num_samples=512,
        eval_num_samples=256)
    hugectr.tools.DataGenerator(p).generate()
    meta = json.load(open(tmp_path / "train" / "_metadata.json"))
    assert [c["col_name"] for c in meta["cats"]] == ["C1", "C2", "C3", "C4"]
    assert meta["file_stats"][0]["num_rows"] == 256 and len(meta["file_stats"]) == 2
    inp = hugectr.Input(label_dim=1, label_name="label", dense_dim=5, dense_name="dense",
                        data_reader_sparse_param_array=[
                            hugectr.DataReaderSparseParam("data1", 1, True, 4)])
    r = data.ParquetReader(p.source, inp, sizes, 128, 0, 1, torch.device("cpu"), True, False)
    n = 0
    while True:
        b = r.next_batch()
        if b is None:
            break
        n += 1
        ro, keys = b["sparse"]["data1"]
        assert ro.tolist() == list(range(128 * 4 + 1)) and keys.numel() == 128 * 4
        k = keys.view(128, 4).numpy()
        offs = np.array([0, 50, 57, 1057])
        assert ((k - offs) >= 0).all() and ((k - offs) < np.array(sizes)).all()  # slot offsets added
        assert b["dense"].shape == (128, 5) and b["label"].shape == (128, 1)
    assert n == 4  # 2 files x 256 rows / 128
    # two ranks see the same keys but different dense/label slices
    r0 = data.ParquetReader(p.source, inp, sizes, 128, 0, 2, torch.device("cpu"), True, False)
    r1 = data.ParquetReader(p.source, inp, sizes, 128, 1, 2, torch.device("cpu"), True, False)
    b0, b1 = r0.next_batch(), r1.next_batch()
    assert torch.equal(b0["sparse"]["data1"][1], b1["sparse"]["data1"][1])
    assert b0["dense"].shape == (64, 5) and not torch.equal(b0["dense"], b1["dense"])


@pytest.mark.parametrize("check_sum", [False, True])
@pytest.mark.parametrize("i64", [True, False])
def test_norm_format_roundtrip(tmp_path, check_sum, i64):
    from hugectr_amd import data
    rng = np.random.default_rng(0)
    n, L, Dn, S = 20, 1, 3, 4
    label = rng.random((n, L), dtype=np.float32)
    dense = rng.random((n, Dn), dtype=np.float32)
    cats = [rng.integers(0, 1000, size=(n, h)).astype(np.int64) for h in (1, 3, 2, 1)]
    path = str(tmp_path / "a.data")
    data.write_norm(path, label, dense, cats, i64_key=i64, check_sum=check_sum)
    l2, d2, ro, keys = data.read_norm(path, i64_key=i64)
    assert np.array_equal(l2, label) and np.array_equal(d2, dense)
    assert ro[-1] == n * 7 and ro.size == n * S + 1
    want = np.concatenate([np.concatenate([c[i] for c in cats]) for i in range(n)])
    assert np.array_equal(keys, want)
    # header layout: DataSetHeader = 8 x int64 (common.hpp:184-191)
    raw = open(path, "rb").read()
    off = 4 if check_sum else 0
    hdr = np.frombuffer(raw, "<i8", 8, off)
    assert hdr.tolist() == [1 if check_sum else 0, n, L, Dn, S, 0, 0, 0]
    if check_sum:
        b = bytearray(raw)
        b[80] ^= 0xFF  # corrupt one payload byte of the first record
        open(path, "wb").write(bytes(b))
        with pytest.raises(AssertionError):
            data.read_norm(path, i64_key=i64)


@pytest.mark.parametrize("float_ld", [True, False])
def test_raw_format_reader(tmp_path, float_ld):
    """Raw (RawAsync) dataset: fixed-size 4-byte records, static multi-hot, uint32 keys + slot
    offsets, integer dense -> log(x + 1) (docs/source/api/python_interface.md "Raw")"""
    import hugectr_amd.hugectr as hugectr
    from hugectr_amd import data
    rng = np.random.default_rng(1)
    n, L, Dn = 300, 1, 3
    hot = [2, 1, 3]
    sizes = [40, 9, 100]
    label = rng.integers(0, 2, size=(n, L))
    dense = rng.integers(0, 50, size=(n, Dn)) if not float_ld else rng.random((n, Dn))
    cats = np.concatenate([rng.integers(0, v, size=(n, h)) for v, h in zip(sizes, hot)], axis=1)
    path = str(tmp_path / "train.bin")
    data.write_raw(path, label, dense, cats, float_label_dense=float_ld)
    assert os.path.getsize(path) == n * (L + Dn + sum(hot)) * 4
    inp = hugectr.Input(label_dim=L, label_name="label", dense_dim=Dn, dense_name="dense",
                        data_reader_sparse_param_array=[
                            hugectr.DataReaderSparseParam("wide", [2, 1], True, 2),
                            hugectr.DataReaderSparseParam("deep", 3, True, 1)])
    r = data.RawReader(path, inp, sizes, 64, 1, 2, torch.device("cpu"), 0, float_ld, False)
    nb = 0
    while True:
        b = r.next_batch()
        if b is None:
            break
        a = nb * 64
        ro, keys = b["sparse"]["wide"]
        assert ro.tolist() == np.concatenate([[0], np.cumsum(np.tile([2, 1], 64))]).tolist()
        want = cats[a:a + 64, :3] + np.array([0, 0, 40])
        assert (keys.view(64, 3).numpy() == want).all()
        ro2, keys2 = b["sparse"]["deep"]
        assert (keys2.view(64, 3).numpy() == cats[a:a + 64, 3:] + 49).all()
        want_d = dense[a + 32:a + 64] if float_ld else np.log(dense[a + 32:a + 64] + 1.0)
        assert np.allclose(b["dense"].numpy(), want_d.astype(np.float32), rtol=1e-6)
        assert (b["label"].numpy() == label[a + 32:a + 64]).all()     # rank 1 of 2
        nb += 1
    assert nb == 4  # 300 // 64, tail dropped
    # keys AND row offsets carry the model's key type: solver.i64_input_key = False hands the
    # u32 embedding 4-byte values (int64 arrays were read as uint32 by the C ABI)
    r32 = data.RawReader(path, inp, sizes, 64, 1, 2, torch.device("cpu"), 0, float_ld, False,
                         i64_key=False)
    b32 = r32.next_batch()
    ro, keys = b32["sparse"]["wide"]
    assert ro.dtype == torch.int32 and keys.dtype == torch.int32
    assert (keys.view(64, 3).numpy() == cats[:64, :3] + np.array([0, 0, 40])).all()


def test_raw_reader_reads_the_reference_converters_file():
    """tests/golden/raw_mlperf_val.bin was written by the reference's own converter
    (R/samples/dlrm/preprocessing/convert_to_raw.py, see tests/golden/make_raw_golden.py) in the
    layout the MLPerf DLRM-DCNv2 sample trains from: the reader configured like the sample
    (train.py:350-378: RawAsync, AsyncParam(multi_hot_reader, is_dense_float), one multi-hot
    DataReaderSparseParam per table) must return the converter's inputs"""
    import hugectr_amd.hugectr as hugectr
    from hugectr_amd import data
    here = os.path.dirname(os.path.abspath(__file__))
    src = np.load(os.path.join(here, "golden", "raw_mlperf_inputs.npz"))
    hot = [src[str(i)].shape[1] for i in range(26)]
    n = src["labels"].size
    inp = hugectr.Input(label_dim=1, label_name="label", dense_dim=13, dense_name="dense",
                        data_reader_sparse_param_array=[
                            hugectr.DataReaderSparseParam(f"data{i}", hot[i], True, 1)
                            for i in range(26)])
    ap = hugectr.AsyncParam(num_threads=1, num_batches_per_thread=16, shuffle=False,
                            multi_hot_reader=True, is_dense_float=True)
    path = os.path.join(here, "golden", "raw_mlperf_val.bin")
    assert os.path.getsize(path) == n * (1 + 13 + sum(hot)) * 4
    B, world = 32, 2
    for rank in range(world):
        r = data.RawReader(path, inp, [], B, rank, world, torch.device("cpu"), 0, False, False, ap)
        for nb in range(n // B):
            b = r.next_batch()
            a = nb * B
            mine = slice(a + rank * (B // world), a + (rank + 1) * (B // world))
            assert (b["label"].numpy().reshape(-1) == src["labels"][mine].astype(np.float32)).all()
            assert (b["dense"].numpy() == src["dense"][mine]).all()      # float words, bit for bit
            for i in range(26):
                ro, keys = b["sparse"][f"data{i}"]                        # the full batch
                assert ro.tolist() == list(range(0, B * hot[i] + 1, hot[i]))
                assert (keys.view(B, hot[i]).numpy() == src[str(i)][a:a + B]).all()
        assert r.next_batch() is None


def test_parquet_multi_hot_columns(tmp_path):
    """list-typed categorical columns (multi-hot) next to scalar ones: bucket order (sample, slot),
    slot offsets added, one DataReaderSparseParam per column or one over several columns"""
    import pyarrow.parquet as pq
    import hugectr_amd.hugectr as hugectr
    from hugectr_amd import data
    sizes = [50, 7, 1000, 3]
    hot = [3, 1, 2, 1]
    p = hugectr.tools.DataGeneratorParams(
        format=hugectr.DataReaderType_t.Parquet, label_dim=1, dense_dim=2, num_slot=4,
        i64_input_key=True, source=str(tmp_path / "train" / "_file_list.txt"),
        eval_source=str(tmp_path / "val" / "_file_list.txt"), slot_size_array=sizes, nnz_array=hot,
        dist_type=hugectr.Distribution_t.PowerLaw, power_law_type=hugectr.PowerLaw_t.Short,
        num_files=1, eval_num_files=1, num_samples_per_file=96, num_samples=96, eval_num_samples=32)
    hugectr.tools.DataGenerator(p).generate()
    t = pq.read_table(str(tmp_path / "train" / "gen_0.parquet"))
    cols = [t[f"C{i + 1}"].to_pylist() for i in range(4)]
    offs = np.array([0, 50, 57, 1057])
    for params in ([hugectr.DataReaderSparseParam("all", hot, True, 4)],
                   [hugectr.DataReaderSparseParam(f"d{i}", hot[i], True, 1) for i in range(4)]):
        inp = hugectr.Input(label_dim=1, label_name="label", dense_dim=2, dense_name="dense",
                            data_reader_sparse_param_array=params)
        r = data.ParquetReader(p.source, inp, sizes, 32, 0, 1, torch.device("cpu"), True, False)
        for nb in range(3):
            b = r.next_batch()
            a = nb * 32
            if len(params) == 1:
                ro, keys = b["sparse"]["all"]
                want, lens = [], []
                for i in range(a, a + 32):
                    for s in range(4):
                        v = cols[s][i] if isinstance(cols[s][i], list) else [cols[s][i]]
                        want += [k + offs[s] for k in v]
                        lens.append(len(v))
                assert keys.tolist() == want
                assert ro.tolist() == np.concatenate([[0], np.cumsum(lens)]).tolist()
            else:
                for s in range(4):
                    ro, keys = b["sparse"][f"d{s}"]
                    want = []
                    for i in range(a, a + 32):
                        v = cols[s][i] if isinstance(cols[s][i], list) else [cols[s][i]]
                        want += [k + offs[s] for k in v]
                    assert keys.tolist() == want and ro.numel() == 33


def test_auc_shares_ranks_among_tied_scores():
    """hugectr.Model's AUC: ties between a positive and a negative count one half"""
    import torch
    from sklearn.metrics import roc_auc_score
    from hugectr_amd.hugectr import _auc
    rng = np.random.default_rng(0)
    for digits in (1, 2, 6):
        p = np.round(rng.random(800), digits)
        y = (rng.random(800) < 0.35).astype(np.float64)
        assert abs(_auc(torch.from_numpy(p), torch.from_numpy(y)) - roc_auc_score(y, p)) < 1e-12
    assert _auc(torch.zeros(10), torch.tensor([0., 1.] * 5)) == 0.5


def _ebc_csr_from_inputs(b, names, offsets, B):
    """what Model._ebc_forward assembles from the per-input tensors when a reader has no
    ready-made CSR: feature-major, raw keys (slot offsets taken off again)"""
    ks, lens = [], []
    for n, o in zip(names, offsets):
        ro, k = b["sparse"][n]
        ks.append(k.to(torch.int64) - o)
        lens.append((ro[1:] - ro[:-1]).to(torch.int64))
    br = torch.zeros(len(names) * B + 1, dtype=torch.int64)
    br[1:] = torch.cumsum(torch.cat(lens), 0)
    return torch.cat(ks), br


def test_readers_hand_an_embedding_collection_its_global_csr(tmp_path):
    """Parquet (scalar + list columns) and Raw (static hotness): the `ebc` entry of a batch equals
    the CSR cut out of the per-input tensors, for lookups in a different order than the inputs"""
    import hugectr_amd.hugectr as hugectr
    from hugectr_amd import data
    sizes, hot = [50, 7, 1000, 3], [3, 1, 2, 1]
    offs = [0, 50, 57, 1057]
    params = [hugectr.DataReaderSparseParam(f"d{i}", hot[i], True, 1) for i in range(4)]
    inp = hugectr.Input(label_dim=1, label_name="label", dense_dim=2, dense_name="dense",
                        data_reader_sparse_param_array=params)
    groups = [["d2", "d0", "d3"], ["d1"]]
    goffs = [[offs[2], offs[0], offs[3]], [offs[1]]]
    p = hugectr.tools.DataGeneratorParams(
        format=hugectr.DataReaderType_t.Parquet, label_dim=1, dense_dim=2, num_slot=4,
        i64_input_key=True, source=str(tmp_path / "train" / "_file_list.txt"),
        eval_source=str(tmp_path / "val" / "_file_list.txt"), slot_size_array=sizes, nnz_array=hot,
        dist_type=hugectr.Distribution_t.PowerLaw, power_law_type=hugectr.PowerLaw_t.Short,
        num_files=1, eval_num_files=1, num_samples_per_file=96, num_samples=96, eval_num_samples=32)
    hugectr.tools.DataGenerator(p).generate()
    rng = np.random.default_rng(3)
    n = 96
    cats = np.concatenate([rng.integers(0, v, size=(n, h)) for v, h in zip(sizes, hot)], axis=1)
    raw = str(tmp_path / "train.bin")
    data.write_raw(raw, rng.integers(0, 2, size=(n, 1)), rng.random((n, 2)), cats, float_label_dense=True)
    readers = [data.ParquetReader(p.source, inp, sizes, 32, 0, 1, torch.device("cpu"), True, False),
               data.RawReader(raw, inp, sizes, 32, 0, 1, torch.device("cpu"), 0, True, False)]
    for r in readers:
        r.ebc_groups = groups
        for _ in range(3):
            b = r.next_batch()
            assert len(b["ebc"]) == 2
            for (gk, gbr), names, o in zip(b["ebc"], groups, goffs):
                wk, wbr = _ebc_csr_from_inputs(b, names, o, 32)
                assert gk.dtype == torch.int64 and torch.equal(gk, wk) and torch.equal(gbr, wbr)


def test_scale_layer_follows_the_reference_upscale_and_first_copy_downscale():
    """Layer_t.Scale (R/HugeCTR/src/layers/scale_layer.cu:30-66, the MMoE sample's gate mixing):
    fprop axis 0: out[b, j * factor + i] = in[b, j]; axis 1: out[b, i * n + j] = in[b, j]; bprop is
    the reference's downscale_kernel -- the gradient of the FIRST copy only, not the sum."""
    import numpy as np
    import torch
    from hugectr_amd.hugectr import _ScaleFn
    rng = np.random.default_rng(1)
    x = torch.from_numpy(rng.standard_normal((5, 3)).astype(np.float32)).requires_grad_()
    for axis, factor in ((0, 4), (1, 4), (0, 1)):
        x.grad = None
        y = _ScaleFn.apply(x, axis, factor)
        xn = x.detach().numpy()
        want = np.repeat(xn, factor, axis=1) if axis == 0 else np.tile(xn, (1, factor))
        assert y.shape == (5, 3 * factor) and (y.detach().numpy() == want).all()
        g = torch.from_numpy(rng.standard_normal((5, 3 * factor)).astype(np.float32))
        y.backward(g)
        gn = g.numpy()
        first = gn[:, ::factor] if axis == 0 else gn[:, :3]
        assert (x.grad.numpy() == first).all()


def test_shard_compression_strategy_is_checked_like_the_reference():
    """EmbeddingCollectionConfig.shard(..., compression_strategy): the sanity checks of
    EmbeddingCollectionParam's constructor (R/HugeCTR/embedding/common.cpp:280-307) -- a table under
    one strategy only, the strategies name exactly the model-parallel tables -- and the request is
    kept (not dropped): `Unique` is refused by the runtime on more than one GPU, where it would
    select an operator this library does not have, instead of silently running as `Reduction`."""
    import hugectr_amd.hugectr as hugectr
    from hugectr_amd.embedding_collection import EmbeddingCollectionConfig, EmbeddingTableConfig
    assert hugectr.CompressionStrategy.Reduction.name == "Reduction"
    assert hugectr.CompressionStrategy.Unique.name == "Unique"
    tabs = [EmbeddingTableConfig(f"t{i}", 100, 8) for i in range(3)]

    def cfg():
        c = EmbeddingCollectionConfig()
        c.embedding_lookup(tabs, ["a", "b", "c"], "emb", ["sum"] * 3)
        return c
    names = [["t0", "t1", "t2"]]
    c = cfg().shard(names, [("mp", ["t0", "t1"]), ("dp", ["t2"])],
                    [(hugectr.CompressionStrategy.Reduction, ["t0"]),
                     (hugectr.CompressionStrategy.Unique, ["t1"])])
    assert c.compression == {"t0": "reduction", "t1": "unique"}
    assert cfg().shard(names, [("mp", ["t0", "t1", "t2"])]).compression == {}
    with pytest.raises(RuntimeError, match="Duplicate table id"):
        cfg().shard(names, [("mp", ["t0", "t1", "t2"])],
                    [(hugectr.CompressionStrategy.Reduction, ["t0", "t1", "t2"]),
                     (hugectr.CompressionStrategy.Unique, ["t1"])])
    with pytest.raises(RuntimeError, match="does not match"):
        cfg().shard(names, [("mp", ["t0", "t1"]), ("dp", ["t2"])],
                    [(hugectr.CompressionStrategy.Reduction, ["t0"])])
    from hugectr_amd import _lib
    from hugectr_amd.embedding_collection import EmbeddingCollection
    with pytest.raises(_lib.HugeCTRAmdError, match="CompressionStrategy.Unique"):
        EmbeddingCollection.for_rank(0, 2, c, 64)
