"""Loss-curve equivalence (north star: "loss-curve equivalent to reference"; the reference states
its own criterion the same way, R/docs/source/performance.md:18-27): a model trained through the
`hugectr` surface of this repo -- HIP embedding, HIP Interaction / MultiCross, library GEMMs --
against a FULLY INDEPENDENT fp32 path that shares no product code: the CPU oracle for the sparse
side (hash -> rows, pooling, gradient reduce, sparse optimizer: oracle/hctr_oracle.c, itself pinned
against the reference's CPU embedding built from the checkout) and plain torch fp32 on the CPU for
the dense tower, written out here layer by layer.  Same initial weights (sparse: one model
directory loaded by both; dense: read out of the compiled model), same batches, 60 steps:
per-step loss within 1e-3 relative in fp32, 1e-2 with use_mixed_precision (fp16 vectors / fp16
MLP GEMMs against the fp32 path; the reference's own fp16 embedding tolerance is 5e-3,
localized_slot_sparse_embedding_hash_test.cu:186-195)."""
import os

import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu

C1_SLOTS = [39884, 39043, 17289, 7420, 20263, 3, 7120, 1543, 39884, 39043, 17289, 7420, 20263, 3,
            7120, 1543, 63, 63, 39884, 39043, 17289, 7420, 20263, 3, 7120, 1543]  # R/README.md:72-74
STEPS = 60


def _gen(hugectr, d, sizes, batch, i64):
    hugectr.tools.DataGenerator(hugectr.tools.DataGeneratorParams(
        format=hugectr.DataReaderType_t.Parquet, label_dim=1, dense_dim=13, num_slot=len(sizes),
        i64_input_key=i64, source=str(d / "train" / "_file_list.txt"), eval_source="",
        slot_size_array=sizes, dist_type=hugectr.Distribution_t.PowerLaw,
        power_law_type=hugectr.PowerLaw_t.Short, num_files=1, eval_num_files=0,
        num_samples_per_file=batch * STEPS, num_samples=batch * STEPS,
        eval_num_samples=0)).generate()
    # learnable labels: a function of two features (the generator draws coin flips)
    import glob
    import pyarrow as pa
    import pyarrow.parquet as pq
    for f in glob.glob(str(d / "train" / "*.parquet")):
        t = pq.read_table(f)
        lab = ((t["C3"].to_numpy() + t["C7"].to_numpy()) % 2).astype(np.float32)
        t = t.set_column(t.schema.get_field_index("label"), "label", pa.array(lab, pa.float32()))
        pq.write_table(t, f)


def _sparse_model_dir(d, sizes, D, seed):
    rng = np.random.default_rng(seed)
    V = int(sum(sizes))
    os.makedirs(d, exist_ok=True)
    np.arange(V, dtype="<i8").tofile(os.path.join(d, "key"))
    np.repeat(np.arange(len(sizes)), sizes).astype("<u8").tofile(os.path.join(d, "slot_id"))
    vec = (rng.standard_normal((V, D)) * 0.05).astype("<f4")
    vec.tofile(os.path.join(d, "emb_vector"))
    return vec


def vec0_of(tmp_path, D):
    return np.fromfile(os.path.join(str(tmp_path), "sparse0", "emb_vector"), "<f4").reshape(-1, D)


class _IndependentTower:
    """the dense layers of the graph in plain torch fp32 on the CPU, formulas written out"""

    def __init__(self, model, hugectr):
        self.T = hugectr.Layer_t
        self.layers = model.layers
        self.p = {}
        for i, L in enumerate(model.layers):
            m = model._mods[f"l{i}"] if f"l{i}" in model._mods else None
            t = L.layer_type
            if t == self.T.InnerProduct:
                self.p[i] = [m.weight.detach().float().cpu().clone().requires_grad_(True),
                             m.bias.detach().float().cpu().clone().requires_grad_(True)]
            elif t == self.T.MLP:
                self.p[i] = [w.detach().float().cpu().clone().requires_grad_(True)
                             for pair in zip(m.weights, m.biases) for w in pair]
            elif t == self.T.MultiCross:
                self.p[i] = [m.kernels.detach().cpu().clone().requires_grad_(True),
                             m.biases.detach().cpu().clone().requires_grad_(True)]
        self.params = [q for i in sorted(self.p) for q in self.p[i]]

    def forward(self, tensors):
        T = self.T
        logit = None
        for i, L in enumerate(self.layers):
            t = L.layer_type
            x = [tensors[b] for b in L.bottom_names]
            if t == T.InnerProduct:
                w, b = self.p[i]
                y = x[0].reshape(x[0].shape[0], -1) @ w.t() + b
            elif t == T.MLP:
                y = x[0].reshape(x[0].shape[0], -1)
                ps = self.p[i]
                acts = L.activations or [L.act_type] * len(L.num_outputs)
                for k in range(len(ps) // 2):
                    y = y @ ps[2 * k].t() + ps[2 * k + 1]
                    if getattr(acts[k], "name", "") == "Relu":
                        y = torch.relu(y)
            elif t == T.MultiCross:  # x_{l+1} = x0 * (x_l . w_l) + b_l + x_l
                ker, bia = self.p[i]
                x0 = x[0].reshape(x[0].shape[0], -1)
                xl = x0
                for l in range(ker.shape[0]):
                    xl = x0 * (xl * ker[l]).sum(1, keepdim=True) + bia[l] + xl
                y = xl
            elif t == T.Interaction:  # [mlp | strict lower triangle of X X^T, row major | 0]
                X = torch.cat([x[0].unsqueeze(1), x[1]], dim=1)
                Z = torch.bmm(X, X.transpose(1, 2))
                r, c = torch.tril_indices(X.shape[1], X.shape[1], -1)
                y = torch.cat([x[0], Z[:, r, c], torch.zeros(X.shape[0], 1)], dim=1)
            elif t == T.ReLU:
                y = torch.relu(x[0])
            elif t == T.Dropout:
                assert L.dropout_rate == 0.0
                y = x[0]
            elif t == T.Concat:
                y = torch.cat([v.reshape(v.shape[0], -1) for v in x], dim=1)
            elif t == T.Reshape:
                y = x[0].reshape(-1, L.leading_dim)
            elif t == T.BinaryCrossEntropyLoss:
                logit = x[0]
                continue
            else:
                raise AssertionError(t)
            tensors[L.top_names[0]] = y
        return logit


def _product_params(m, tower):
    """the product's dense parameters in the order of tower.params"""
    out = []
    for i in sorted(tower.p):
        mod = m._mods[f"l{i}"]
        t = m.layers[i].layer_type
        if t == tower.T.InnerProduct:
            out += [mod.weight, mod.bias]
        elif t == tower.T.MLP:
            out += [w for pair in zip(mod.weights, mod.biases) for w in pair]
        else:
            out += [mod.kernels, mod.biases]
    return out


def _run(hugectr, oracle, tmp_path, kind, mixed, opt_name="adam"):
    from hugectr_amd import _lib
    dcn = kind == "dcn"
    sizes = C1_SLOTS if dcn else [203, 18598, 140, 7012, 18977, 4, 6385, 1245, 49, 186, 713, 67288,
                                  11, 2168, 7338, 61]
    B, D, i64 = (1024, 16, False) if dcn else (2048, 32, True)
    _gen(hugectr, tmp_path, sizes, B, i64)
    vec0 = _sparse_model_dir(str(tmp_path / "sparse0"), sizes, D, 5)
    lr = 0.001 if opt_name == "adam" else 0.5
    solver = hugectr.CreateSolver(max_eval_batches=1, batchsize_eval=B, batchsize=B, lr=lr,
                                  vvgpu=[[0]], repeat_dataset=True, i64_input_key=i64,
                                  use_mixed_precision=mixed, scaler=1024.0 if mixed else 1.0)
    reader = hugectr.DataReaderParams(data_reader_type=hugectr.DataReaderType_t.Parquet,
                                      source=[str(tmp_path / "train" / "_file_list.txt")],
                                      eval_source="", slot_size_array=sizes,
                                      check_type=hugectr.Check_t.Non)
    upd = hugectr.Update_t.Global if dcn else hugectr.Update_t.Local
    optimizer = hugectr.CreateOptimizer(optimizer_type=hugectr.Optimizer_t.Adam, update_type=upd) \
        if opt_name == "adam" else hugectr.CreateOptimizer(
            optimizer_type=hugectr.Optimizer_t.SGD, update_type=hugectr.Update_t.Local)
    m = hugectr.Model(solver, reader, optimizer)
    L, T = hugectr.DenseLayer, hugectr.Layer_t
    S = len(sizes)
    m.add(hugectr.Input(label_dim=1, label_name="label", dense_dim=13, dense_name="dense",
                        data_reader_sparse_param_array=[
                            hugectr.DataReaderSparseParam("data1", 1, True, S)]))
    if dcn:  # the README's DCN (R/README.md:106-146), dropout off (it is random)
        m.add(hugectr.SparseEmbedding(
            embedding_type=hugectr.Embedding_t.DistributedSlotSparseEmbeddingHash,
            workspace_size_per_gpu_in_mb=75, embedding_vec_size=D, combiner="sum",
            sparse_embedding_name="emb", bottom_name="data1", optimizer=optimizer))
        m.add(L(layer_type=T.Reshape, bottom_names=["emb"], top_names=["reshape1"], leading_dim=S * D))
        m.add(L(layer_type=T.Concat, bottom_names=["reshape1", "dense"], top_names=["concat1"]))
        m.add(L(layer_type=T.MultiCross, bottom_names=["concat1"], top_names=["multicross1"],
                num_layers=6))
        m.add(L(layer_type=T.InnerProduct, bottom_names=["concat1"], top_names=["fc1"], num_output=1024))
        m.add(L(layer_type=T.ReLU, bottom_names=["fc1"], top_names=["relu1"]))
        m.add(L(layer_type=T.Dropout, bottom_names=["relu1"], top_names=["dropout1"], dropout_rate=0.0))
        m.add(L(layer_type=T.Concat, bottom_names=["dropout1", "multicross1"], top_names=["concat2"]))
        m.add(L(layer_type=T.InnerProduct, bottom_names=["concat2"], top_names=["fc2"], num_output=1))
        m.add(L(layer_type=T.BinaryCrossEntropyLoss, bottom_names=["fc2", "label"], top_names=["loss"]))
    else:   # DLRM: bottom MLP, dot interaction, top MLP (R/samples/dlrm/dlrm_kaggle_fp32.py shape)
        m.add(hugectr.SparseEmbedding(
            embedding_type=hugectr.Embedding_t.LocalizedSlotSparseEmbeddingHash,
            workspace_size_per_gpu_in_mb=64, embedding_vec_size=D, combiner="sum",
            sparse_embedding_name="emb", bottom_name="data1", slot_size_array=sizes,
            optimizer=optimizer))
        m.add(L(layer_type=T.MLP, bottom_names=["dense"], top_names=["mlp1"],
                num_outputs=[64, D], act_type=hugectr.Activation_t.Relu))
        m.add(L(layer_type=T.Interaction, bottom_names=["mlp1", "emb"], top_names=["inter"]))
        m.add(L(layer_type=T.MLP, bottom_names=["inter"], top_names=["mlp2"],
                num_outputs=[128, 64, 1],
                activations=[hugectr.Activation_t.Relu, hugectr.Activation_t.Relu,
                             hugectr.Activation_t.Non]))
        m.add(L(layer_type=T.BinaryCrossEntropyLoss, bottom_names=["mlp2", "label"], top_names=["loss"]))
    m.compile()
    m.load_sparse_weights([str(tmp_path / "sparse0")])
    h = list(m._emb.values())[0][2]
    # ---- the independent path -------------------------------------------------------------------
    tower = _IndependentTower(m, hugectr)
    dopt = torch.optim.Adam(tower.params, lr=lr, betas=(0.9, 0.999), eps=1e-7) \
        if opt_name == "adam" else torch.optim.SGD(tower.params, lr=lr)
    Vmax = h.get_max_vocabulary_size()
    table = np.zeros((Vmax, D), np.float32)
    table[:vec0.shape[0]] = vec0
    s0, s1 = np.zeros_like(table), np.zeros_like(table)
    ht = oracle.HashTable(Vmax, 8 if i64 else 4)
    ht.get_insert(np.arange(vec0.shape[0], dtype=np.int64 if i64 else np.uint32))  # file order
    o = oracle.OptParamsC()
    o.optimizer = oracle.OPT_ADAM if opt_name == "adam" else oracle.OPT_SGD
    o.update_type, o.lr = (1 if dcn and opt_name == "adam" else 0), lr
    o.beta1, o.beta2, o.epsilon, o.scaler = 0.9, 0.999, 1e-7, 1.0
    got, want = [], []
    for step in range(1, STEPS + 1):
        batch = m.reader.next_batch(True)
        ro, keys = batch["sparse"]["data1"]
        ro_np = ro.cpu().numpy().astype(np.int64)
        k_np = keys.cpu().numpy()
        k_np = k_np.astype(np.int64) if i64 else k_np.view(np.uint32) if k_np.dtype != np.uint32 else k_np
        # product path (the batch is handed over instead of being read a second time)
        loss, _ = m._run_batch(batch, True)
        got.append(float(loss))
        # independent path
        vi = ht.get_insert(k_np)
        E = oracle.forward(ro_np, vi, table, D, 0).reshape(B, S, D)
        Et = torch.from_numpy(E).requires_grad_(True)
        tensors = {"dense": batch["dense"].float().cpu(), "label": batch["label"].float().cpu(),
                   "emb": Et}
        logit = tower.forward(tensors)
        ref_loss = torch.nn.functional.binary_cross_entropy_with_logits(logit, tensors["label"])
        dopt.zero_grad()
        ref_loss.backward()
        want.append(float(ref_loss))
        o.times = step
        oracle.update_params(ro_np, vi, Et.grad.numpy().reshape(-1, D), o, table, s0, s1)
        dopt.step()
        if os.environ.get("HCTR_DEBUG_LOSS_CURVE"):  # where do the two paths part?
            torch.cuda.synchronize()
            rows = np.unique(vi)
            te = np.abs(h.table().cpu().numpy()[rows] - table[rows]).max()
            de = max(float((a.detach().float().cpu() - b.detach()).abs().max())
                     for a, b in zip(_product_params(m, tower), tower.params))
            ge = float((m._last_emb_grad.float().cpu() - Et.grad).abs().max()) \
                if hasattr(m, "_last_emb_grad") else -1.0
            print(f"step {step}: loss {got[-1]:.6f} / {want[-1]:.6f}  table err {te:.2e}  "
                  f"dense err {de:.2e}  emb grad err {ge:.2e} (|g| {float(Et.grad.abs().max()):.2e})")
    torch.cuda.synchronize()
    return np.array(got), np.array(want), h, table, ht


# Adam on the README's DCN (C1) and on a DLRM; SGD on the DLRM in fp32 and in the reference's mixed
# precision.  (Adam is not run in mixed precision: the reference keeps the sparse optimizer state
# in the embedding type, SURVEY q6, and fp16 cannot hold v = (1 - beta2) g^2 for the 1e-6-sized
# gradients of a fresh model -- v underflows to 0 and every element takes a m / epsilon step.  That
# is the reference's behaviour too; its mixed-precision DLRM configurations use SGD.)
@pytest.mark.parametrize("kind,mixed,opt_name", [("dcn", False, "adam"), ("dlrm", False, "adam"),
                                                 ("dlrm", False, "sgd"), ("dlrm", True, "sgd")])
def test_loss_curve_follows_the_independent_fp32_path(oracle, tmp_path, kind, mixed, opt_name):
    import hugectr
    got, want, h, table, ht = _run(hugectr, oracle, tmp_path, kind, mixed, opt_name)
    rel = np.abs(got - want) / np.abs(want)
    tol = 1e-2 if mixed else 1e-3
    assert rel.max() < tol, f"max rel loss difference {rel.max():.2e} at step {rel.argmax()}"
    if opt_name == "sgd":
        # (plain SGD barely moves this freshly initialised model in 60 steps; what the case adds
        # is the state itself: without Adam's normalisation the tables must agree element-wise)
        n = ht.size()
        err = np.abs(h.table().cpu().numpy()[:n] - table[:n]).max()
        tol = (1e-3 if mixed else 1e-4) * np.abs(table[:n]).max() + 1e-6
        assert err < tol, err
        assert np.abs(table[:n] - vec0_of(tmp_path, table.shape[1])[:n]).max() > 1e-5, "no update"
    else:
        assert want[-10:].mean() < want[:10].mean() - 0.003, "the model did not learn"
    if opt_name == "adam":
        # the embedding tables themselves after 60 steps of Adam on both sides
        n = ht.size()
        t_gpu = h.table().cpu().numpy()[:n]
        # Adam turns a gradient element that is pure rounding noise into a full +-lr step, so single
        # elements of the two runs walk apart by a few lr (measured: 2.5 lr after 60 steps) while
        # the loss stays within 1e-4; the bound is in units of the learning rate
        err = np.abs(t_gpu - table[:n]).max()
        assert err < 10 * 0.001, err
        assert np.abs(t_gpu - table[:n]).mean() < 0.2 * 0.001
