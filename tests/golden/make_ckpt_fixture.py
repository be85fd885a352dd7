"""Writes a small trained model in the reference's interchange formats -- graph JSON
(Model.graph_to_json), dense model file, sparse model directories (key / slot_id / emb_vector) --
plus `truth.npz` with the same weights taken straight from the live modules, so that a reader can
be checked value by value.  Needs a GPU:

    gpurun -- 'python tests/golden/make_ckpt_fixture.py gpurun_out/ckpt'
    cp gpurun_out/ckpt/* tests/golden/ckpt/        # then: python tests/golden/check_ckpt_with_reference.py

Two models cover the layer kinds with weights: DCN (Distributed embedding, Reshape, Concat,
MultiCross v1, InnerProduct, ReLU, Dropout) and a DLRM-style one (Localized embedding, MLP,
Interaction)."""
import os
import sys

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import hugectr_amd.hugectr as hugectr  # noqa: E402

SIZES = [23, 5, 40, 7, 19, 4, 63, 12]


def gen(tmp):
    p = hugectr.tools.DataGeneratorParams(
        format=hugectr.DataReaderType_t.Parquet, label_dim=1, dense_dim=5, num_slot=len(SIZES),
        i64_input_key=True, source=os.path.join(tmp, "train", "_file_list.txt"),
        eval_source=os.path.join(tmp, "val", "_file_list.txt"), slot_size_array=SIZES,
        dist_type=hugectr.Distribution_t.PowerLaw, power_law_type=hugectr.PowerLaw_t.Short,
        num_files=1, eval_num_files=1, num_samples_per_file=512, num_samples=512, eval_num_samples=256)
    hugectr.tools.DataGenerator(p).generate()
    return p


def base(p, opt_type, update):
    solver = hugectr.CreateSolver(batchsize=128, batchsize_eval=128, lr=0.01, vvgpu=[[0]],
                                  i64_input_key=True, max_eval_batches=1, repeat_dataset=True)
    reader = hugectr.DataReaderParams(data_reader_type=hugectr.DataReaderType_t.Parquet,
                                      source=[p.source], eval_source=p.eval_source,
                                      slot_size_array=SIZES, check_type=hugectr.Check_t.Non)
    opt = hugectr.CreateOptimizer(optimizer_type=opt_type, update_type=update)
    m = hugectr.Model(solver, reader, opt)
    m.add(hugectr.Input(label_dim=1, label_name="label", dense_dim=5, dense_name="dense",
                        data_reader_sparse_param_array=[
                            hugectr.DataReaderSparseParam("data1", 1, True, len(SIZES))]))
    return m, opt


def dcn(p):
    m, opt = base(p, hugectr.Optimizer_t.Adam, hugectr.Update_t.Global)
    D = hugectr.DenseLayer
    T = hugectr.Layer_t
    m.add(hugectr.SparseEmbedding(embedding_type=hugectr.Embedding_t.DistributedSlotSparseEmbeddingHash,
                                  workspace_size_per_gpu_in_mb=1, embedding_vec_size=4, combiner="sum",
                                  sparse_embedding_name="sparse_embedding1", bottom_name="data1",
                                  optimizer=opt))
    m.add(D(layer_type=T.Reshape, bottom_names=["sparse_embedding1"], top_names=["reshape1"],
            leading_dim=32))
    m.add(D(layer_type=T.Concat, bottom_names=["reshape1", "dense"], top_names=["concat1"]))
    m.add(D(layer_type=T.MultiCross, bottom_names=["concat1"], top_names=["multicross1"], num_layers=3))
    m.add(D(layer_type=T.InnerProduct, bottom_names=["concat1"], top_names=["fc1"], num_output=16))
    m.add(D(layer_type=T.ReLU, bottom_names=["fc1"], top_names=["relu1"]))
    m.add(D(layer_type=T.Dropout, bottom_names=["relu1"], top_names=["dropout1"], dropout_rate=0.5))
    m.add(D(layer_type=T.Concat, bottom_names=["dropout1", "multicross1"], top_names=["concat2"]))
    m.add(D(layer_type=T.InnerProduct, bottom_names=["concat2"], top_names=["fc2"], num_output=1))
    m.add(D(layer_type=T.BinaryCrossEntropyLoss, bottom_names=["fc2", "label"], top_names=["loss"]))
    return m


def dlrm(p):
    m, opt = base(p, hugectr.Optimizer_t.SGD, hugectr.Update_t.Local)
    D = hugectr.DenseLayer
    T = hugectr.Layer_t
    m.add(hugectr.SparseEmbedding(embedding_type=hugectr.Embedding_t.LocalizedSlotSparseEmbeddingHash,
                                  slot_size_array=SIZES, embedding_vec_size=8, combiner="sum",
                                  sparse_embedding_name="sparse_embedding1", bottom_name="data1",
                                  optimizer=opt))
    m.add(D(layer_type=T.MLP, bottom_names=["dense"], top_names=["mlp1"], num_outputs=[16, 8],
            act_type=hugectr.Activation_t.Relu, use_bias=True))
    m.add(D(layer_type=T.Interaction, bottom_names=["mlp1", "sparse_embedding1"],
            top_names=["interaction1"]))
    m.add(D(layer_type=T.MLP, bottom_names=["interaction1"], top_names=["mlp2"], num_outputs=[24, 1],
            activations=[hugectr.Activation_t.Relu, hugectr.Activation_t.Non]))
    m.add(D(layer_type=T.BinaryCrossEntropyLoss, bottom_names=["mlp2", "label"], top_names=["loss"]))
    return m


def truth(m):
    """weights as the live modules hold them, in the reference loader's naming
    (hugectr_loader.py: <top>_weight / _bias, <top><i>_weight, <top>_weights[l])"""
    t = {}
    for i, L in enumerate(m.layers):
        mod = m._mods[f"l{i}"] if f"l{i}" in m._mods else None
        top = L.top_names[0]
        if L.layer_type == hugectr.Layer_t.InnerProduct:
            t[top + "_weight"] = mod.weight.detach().t().cpu().numpy()
            t[top + "_bias"] = mod.bias.detach().cpu().numpy().reshape(1, -1)
        elif L.layer_type == hugectr.Layer_t.MLP:
            for j, (w, b) in enumerate(zip(mod.weights, mod.biases)):
                t[f"{top}{j}_weight"] = w.detach().t().cpu().numpy()
                t[f"{top}{j}_bias"] = b.detach().cpu().numpy().reshape(1, -1)
        elif L.layer_type == hugectr.Layer_t.MultiCross:
            t[top + "_weights"] = mod.kernels.detach().cpu().numpy()
            t[top + "_biases"] = mod.biases.detach().cpu().numpy()
    for name, (se, p, h, _, _) in m._emb.items():
        keys, slot, vec = h.dump_parameters()
        t["emb_keys"] = keys.cpu().numpy().astype(np.int64)
        t["emb_vectors"] = vec.cpu().numpy()
    return t


def main():
    out = sys.argv[1]
    os.makedirs(out, exist_ok=True)
    torch.manual_seed(0)
    p = gen(os.path.join(out, "_data"))
    for name, build in (("dcn", dcn), ("dlrm", dlrm)):
        m = build(p)
        m.compile()
        m.fit(max_iter=5, display=0, eval_interval=0, snapshot=0)
        torch.cuda.synchronize()
        m.graph_to_json(os.path.join(out, f"{name}.json"))
        m.save_params_to_files(os.path.join(out, name), 5)
        np.savez(os.path.join(out, f"{name}_truth.npz"), **truth(m))
    import shutil
    shutil.rmtree(os.path.join(out, "_data"))
    print("wrote", sorted(os.listdir(out)))


if __name__ == "__main__":
    main()
