"""Dataset formats either side of the hot path: synthetic data generator and readers.

* Parquet dataset layout of the reference (`_file_list.txt`, `_metadata.json` with file_stats /
  cats / conts / labels; R/HugeCTR/src/data_readers/metadata.cpp:60-135) read with pyarrow.
  Keys get the cumulative `slot_size_array` offsets added, as the reference's Parquet reader does
  (R/HugeCTR/src/pybind/add_input.cpp:315-317), so keys are globally unique.
* Norm binary format (DataSetHeader 8 x int64, R/HugeCTR/include/common.hpp:184-191; per sample
  label[f32 x L] dense[f32 x Dn] then per slot `nnz:int32, keys[nnz]`; optional CheckSum framing
  `int32 length | payload | int8 sum`, R/HugeCTR/include/data_readers/check_sum.hpp:39-75) --
  write + read helpers (the reference's Python path marks Norm deprecated; its CPU test oracle
  still reads it, R/test/utest/embedding/sparse_embedding_hash_cpu.hpp:343-377).
* hugectr.tools.DataGenerator(DataGeneratorParams).generate()
  (R/HugeCTR/include/pybind/data_generator_wrapper.hpp:29-69; power law alpha: Long 0.9 /
  Medium 1.1 / Short 1.3, R/HugeCTR/src/data_generator.cpp:94-116).

Every rank reads the FULL batch of keys (the reference broadcasts the whole CSR to every GPU,
R/HugeCTR/src/data_readers/data_collector.cu:86-113) and its own slice of dense/label.
"""
from __future__ import annotations

import json
import os
import struct
from dataclasses import dataclass, field
from typing import Sequence

import numpy as np
import torch


def powerlaw_keys(rng: np.random.Generator, n: int, vocab: int, alpha: float) -> np.ndarray:
    """IntPowerLawDataSimulator (R/HugeCTR/include/data_generator.hpp:108-129), vectorised."""
    u = rng.random(n, dtype=np.float32).astype(np.float64)
    a = 1.0 - alpha
    y = ((float(vocab) ** a - 1.0) * u + 1.0) ** (1.0 / a)
    return np.clip(np.round(y) - 1, 0, vocab - 1).astype(np.int64)


@dataclass
class DataGeneratorParams:
    format: object
    label_dim: int
    dense_dim: int
    num_slot: int
    i64_input_key: bool
    source: str
    eval_source: str
    slot_size_array: Sequence[int]
    nnz_array: Sequence[int] = field(default_factory=list)
    check_type: object = None
    dist_type: object = None
    power_law_type: object = None
    alpha: float = 1.2
    num_files: int = 128
    eval_num_files: int = 32
    num_samples_per_file: int = 40960
    num_samples: int = 5242880
    eval_num_samples: int = 1310720
    float_label_dense: bool = False
    num_threads: int = 1


class DataGenerator:
    def __init__(self, data_generator_params: DataGeneratorParams):
        self.p = data_generator_params

    def _alpha(self) -> float:
        p = self.p
        dist = getattr(p.dist_type, "name", "PowerLaw")
        if dist == "Uniform":
            return 0.0
        plt = getattr(p.power_law_type, "name", "Specific")
        return {"Long": 0.9, "Medium": 1.1, "Short": 1.3}.get(plt, p.alpha)

    def _samples(self, rng, n):
        p = self.p
        alpha = self._alpha()
        nnz = list(p.nnz_array) if p.nnz_array else [1] * p.num_slot
        label = (rng.random((n, p.label_dim)) < 0.5).astype(np.float32)
        dense = rng.random((n, p.dense_dim), dtype=np.float32)
        cats = []
        for s in range(p.num_slot):
            v = int(p.slot_size_array[s])
            k = (powerlaw_keys(rng, n * nnz[s], v, alpha) if alpha > 0
                 else rng.integers(0, v, size=n * nnz[s]).astype(np.int64))
            cats.append(k.reshape(n, nnz[s]))
        return label, dense, cats

    def generate(self):
        p = self.p
        fmt = getattr(p.format, "name", str(p.format))
        rng = np.random.default_rng(20240923)
        for src, total, nfiles in ((p.source, p.num_samples, p.num_files),
                                   (p.eval_source, p.eval_num_samples, p.eval_num_files)):
            if not src:
                continue
            folder = os.path.dirname(src) or "."
            os.makedirs(folder, exist_ok=True)
            per_file = min(p.num_samples_per_file, max(1, total // max(nfiles, 1)))
            nfiles = max(1, total // per_file)
            names = []
            stats = []
            for f in range(nfiles):
                label, dense, cats = self._samples(rng, per_file)
                if fmt == "Parquet":
                    name = os.path.join(folder, f"gen_{f}.parquet")
                    write_parquet(name, label, dense, cats)
                    stats.append({"file_name": os.path.basename(name), "num_rows": per_file})
                else:
                    name = os.path.join(folder, f"gen_{f}.data")
                    write_norm(name, label, dense, cats, p.i64_input_key,
                               check_sum=getattr(p.check_type, "name", "Sum") == "Sum")
                names.append(name)
            with open(src, "w") as fl:
                fl.write(f"{len(names)}\n" + "\n".join(names) + "\n")
            if fmt == "Parquet":
                cols = parquet_columns(p.label_dim, p.dense_dim, p.num_slot)
                meta = {"file_stats": stats,
                        "labels": [{"col_name": c, "index": i} for i, c in enumerate(cols[0])],
                        "conts": [{"col_name": c, "index": p.label_dim + i}
                                  for i, c in enumerate(cols[1])],
                        "cats": [{"col_name": c, "index": p.label_dim + p.dense_dim + i}
                                 for i, c in enumerate(cols[2])]}
                with open(os.path.join(folder, "_metadata.json"), "w") as fm:
                    json.dump(meta, fm)


def parquet_columns(label_dim, dense_dim, num_slot):
    return ([f"label{i}" if label_dim > 1 else "label" for i in range(label_dim)],
            [f"I{i + 1}" for i in range(dense_dim)], [f"C{i + 1}" for i in range(num_slot)])


def write_parquet(path, label, dense, cats):
    import pyarrow as pa
    import pyarrow.parquet as pq
    lc, dc, cc = parquet_columns(label.shape[1], dense.shape[1], len(cats))
    arrays, names = [], []
    for i, c in enumerate(lc):
        arrays.append(pa.array(label[:, i], type=pa.float32()))
        names.append(c)
    for i, c in enumerate(dc):
        arrays.append(pa.array(dense[:, i], type=pa.float32()))
        names.append(c)
    for i, c in enumerate(cc):
        k = cats[i]
        if k.shape[1] == 1:
            arrays.append(pa.array(k[:, 0], type=pa.int64()))
        else:  # multi-hot: list<int64> column
            arrays.append(pa.array(k.tolist(), type=pa.list_(pa.int64())))
        names.append(c)
    pq.write_table(pa.Table.from_arrays(arrays, names=names), path)


def write_norm(path, label, dense, cats, i64_key=True, check_sum=False):
    n = label.shape[0]
    kfmt = "<i8" if i64_key else "<u4"
    with open(path, "wb") as f:
        hdr = struct.pack("<8q", 1 if check_sum else 0, n, label.shape[1], dense.shape[1], len(cats),
                          0, 0, 0)
        _w(f, hdr, check_sum)
        for i in range(n):
            rec = label[i].astype("<f4").tobytes() + dense[i].astype("<f4").tobytes()
            for k in cats:
                rec += struct.pack("<i", k.shape[1]) + k[i].astype(kfmt).tobytes()
            _w(f, rec, check_sum)


def _w(f, payload: bytes, check_sum: bool):
    if check_sum:
        s = np.frombuffer(payload, dtype=np.int8).sum(dtype=np.int64)
        f.write(struct.pack("<i", len(payload)) + payload + struct.pack("<b", int(np.int8(s))))
    else:
        f.write(payload)


def read_norm(path, i64_key=True):
    """-> (label [n,L], dense [n,Dn], row_offset [n*S+1], keys [nnz]) as the reference's
    read_a_batch builds them (sparse_embedding_hash_cpu.hpp:343-377)."""
    raw = open(path, "rb").read()
    pos = [0]
    framed = struct.unpack_from("<i", raw, 0)[0] == 64  # CheckSum framing wraps the 64-byte header

    def rd(nbytes):
        b = raw[pos[0]:pos[0] + nbytes]
        pos[0] += nbytes
        return b

    def record():
        if framed:
            ln = struct.unpack("<i", rd(4))[0]
            payload = rd(ln)
            chk = struct.unpack("<b", rd(1))[0]
            assert int(np.int8(np.frombuffer(payload, dtype=np.int8).sum(dtype=np.int64))) == chk, \
                "Norm file: checksum mismatch"
            return payload
        return None

    hdr = record() if framed else rd(64)
    err, n, L, Dn, S, _, _, _ = struct.unpack("<8q", hdr)
    ksz = 8 if i64_key else 4
    label = np.empty((n, L), np.float32)
    dense = np.empty((n, Dn), np.float32)
    ro = [0]
    keys = []
    for i in range(n):
        if framed:
            buf, o = record(), 0
        else:
            buf, o = raw, pos[0]
        label[i] = np.frombuffer(buf, "<f4", L, o)
        o += 4 * L
        dense[i] = np.frombuffer(buf, "<f4", Dn, o)
        o += 4 * Dn
        for _ in range(S):
            nnz = struct.unpack_from("<i", buf, o)[0]
            o += 4
            keys.append(np.frombuffer(buf, "<i8" if i64_key else "<u4", nnz, o).astype(np.int64))
            o += ksz * nnz
            ro.append(ro[-1] + nnz)
        if not framed:
            pos[0] = o
    return label, dense, np.asarray(ro, np.int64), (np.concatenate(keys) if keys else
                                                    np.empty(0, np.int64))


class ParquetReader:
    """Streams batches from a reference-layout Parquet dataset."""

    def __init__(self, file_list: str, inp, slot_size_array, batch, rank, world, device, i64_key,
                 repeat: bool):
        import pyarrow.parquet as pq
        self.pq = pq
        lines = [l.strip() for l in open(file_list) if l.strip()]
        self.files = lines[1:1 + int(lines[0])]
        folder = os.path.dirname(file_list) or "."
        self.files = [f if os.path.isabs(f) or os.path.exists(f) else os.path.join(folder, os.path.basename(f))
                      for f in self.files]
        meta_path = os.path.join(folder, "_metadata.json")
        if not os.path.exists(meta_path):
            raise RuntimeError(f"{meta_path} not found (Parquet datasets need _metadata.json)")
        meta = json.load(open(meta_path))
        self.label_cols = [c["col_name"] for c in meta["labels"]]
        self.cont_cols = [c["col_name"] for c in meta["conts"]]
        self.cat_cols = [c["col_name"] for c in meta["cats"]]
        self.inp, self.batch, self.rank, self.world = inp, batch, rank, world
        self.device, self.repeat = device, repeat
        self.key_dtype = torch.int64 if i64_key else torch.int32
        ssa = list(slot_size_array)
        self.slot_offsets = np.concatenate([[0], np.cumsum(ssa)[:-1]]).astype(np.int64) if ssa \
            else np.zeros(len(self.cat_cols), np.int64)
        self._file_idx, self._buf, self._pos = 0, None, 0
        self._epoch_done = False
        # embedding_collection models (see RawReader.ebc_groups): slot of every one-slot input
        self.ebc_groups = []
        self._slot_of = {}
        s0 = 0
        for p in inp.sparse_params:
            self._slot_of[p.top_name] = (s0, p.slot_num)
            s0 += p.slot_num

    def _load_next_file(self) -> bool:
        if self._file_idx >= len(self.files):
            if not self.repeat:
                return False
            self._file_idx = 0
        t = self.pq.read_table(self.files[self._file_idx])
        self._file_idx += 1
        label = np.stack([t[c].to_numpy() for c in self.label_cols], 1).astype(np.float32)
        dense = (np.stack([t[c].to_numpy() for c in self.cont_cols], 1).astype(np.float32)
                 if self.cont_cols else np.zeros((t.num_rows, 0), np.float32))
        cats = []
        for s, c in enumerate(self.cat_cols):
            col = t[c]
            if str(col.type).startswith("list"):
                # multi-hot column: (row offsets, flat keys) straight from the Arrow buffers
                arr = col.combine_chunks()
                off = arr.offsets.to_numpy().astype(np.int64)
                vals = arr.values.to_numpy().astype(np.int64)[off[0]:off[-1]] + self.slot_offsets[s]
                cats.append((off - off[0], vals))
            else:
                cats.append(col.to_numpy().astype(np.int64) + self.slot_offsets[s])
        self._buf, self._pos = (label, dense, cats), 0
        return True

    def next_batch(self):
        B = self.batch
        if self._buf is None or self._pos + B > self._buf[0].shape[0]:
            # the reference drops the incomplete tail of a file (drop_incomplete_batch=True)
            if not self._load_next_file():
                return None
            if self._buf[0].shape[0] < B:
                raise RuntimeError("parquet file holds fewer rows than one batch")
        label, dense, cats = self._buf
        a, b = self._pos, self._pos + B
        self._pos = b
        bpg = B // self.world
        sl = slice(a + self.rank * bpg, a + (self.rank + 1) * bpg)
        out = {"label": torch.from_numpy(label[sl]).to(self.device),
               "dense": torch.from_numpy(dense[sl]).to(self.device), "sparse": {}}
        # one DataReaderSparseParam may cover a contiguous group of slots
        s0 = 0
        for p in self.inp.sparse_params:
            group = cats[s0:s0 + p.slot_num]
            s0 += p.slot_num
            if all(isinstance(g, np.ndarray) for g in group):  # one-hot fast path
                keys = np.stack([g[a:b] for g in group], 1).reshape(-1)
                ro = np.arange(B * p.slot_num + 1, dtype=np.int64)
            else:
                # ragged group: bucket (sample, slot) order, vectorised per slot
                lens = np.empty((B, p.slot_num), np.int64)
                for s, g in enumerate(group):
                    lens[:, s] = (g[0][a + 1:b + 1] - g[0][a:b]) if isinstance(g, tuple) else 1
                ro = np.concatenate([[0], np.cumsum(lens.reshape(-1))]).astype(np.int64)
                keys = np.empty(int(ro[-1]), np.int64)
                starts = ro[:-1].reshape(B, p.slot_num)
                for s, g in enumerate(group):
                    if isinstance(g, tuple):
                        off, vals = g
                        seg = vals[off[a]:off[b]]
                        # destination of every key: its bucket's start + position inside the bucket
                        inner = np.arange(seg.size) - np.repeat(off[a:b] - off[a], lens[:, s])
                        keys[np.repeat(starts[:, s], lens[:, s]) + inner] = seg
                    else:
                        keys[starts[:, s]] = g[a:b]
            kt = torch.from_numpy(keys)
            rt = torch.from_numpy(ro)
            if self.key_dtype != torch.int64:
                kt, rt = kt.to(torch.int32), rt.to(torch.int32)
            out["sparse"][p.top_name] = (rt.to(self.device), kt.to(self.device))
        if self.ebc_groups:  # the collections' global feature-major CSRs, raw keys
            out["ebc"] = []
            for names in self.ebc_groups:
                ks, lens = [], []
                for n in names:
                    s, slots = self._slot_of[n]
                    assert slots == 1, "embedding_collection inputs carry one slot per lookup"
                    g = cats[s]
                    if isinstance(g, tuple):
                        off, vals = g
                        ks.append(vals[off[a]:off[b]] - self.slot_offsets[s])
                        lens.append(off[a + 1:b + 1] - off[a:b])
                    else:
                        ks.append(g[a:b] - self.slot_offsets[s])
                        lens.append(np.ones(B, np.int64))
                gbr = np.zeros(len(names) * B + 1, np.int64)
                np.cumsum(np.concatenate(lens), out=gbr[1:])
                out["ebc"].append((torch.from_numpy(np.concatenate(ks).astype(np.int64)).to(self.device),
                                   torch.from_numpy(gbr).to(self.device)))
        return out


# ---- Raw format (docs/source/api/python_interface.md "Raw"; R/HugeCTR/src/data_readers/multi_hot/) ---
# one binary file, fixed-size samples: label[label_dim] dense[dense_dim] keys[sum of hotness], every
# field 4 bytes: label/dense float32 (float_label_dense / is_dense_float) or int32/uint32 (dense is
# then fed through log(x + 1)), keys uint32 with STATIC hotness per slot taken from the Input layer.
def write_raw(path, label, dense, cats, float_label_dense=True):
    """cats: [num_samples, sum_hotness] keys in slot-major order"""
    n = label.shape[0]
    ld = (np.asarray(label, np.float32) if float_label_dense else np.asarray(label, np.int32))
    dd = (np.asarray(dense, np.float32) if float_label_dense else np.asarray(dense, np.uint32))
    rec = np.concatenate([ld.view(np.uint32).reshape(n, -1), dd.view(np.uint32).reshape(n, -1),
                          np.asarray(cats, np.uint32).reshape(n, -1)], axis=1)
    rec.astype("<u4").tofile(path)


class RawReader:
    """DataReaderType_t.RawAsync: streams batches of a Raw file (np.memmap; the reference uses
    Linux AIO worker threads -- I/O engines are outside the scope contract, the FORMAT is not)."""

    def __init__(self, path: str, inp, slot_size_array, batch, rank, world, device, num_samples,
                 float_label_dense: bool, repeat: bool, async_param=None, i64_key: bool = True):
        self.inp, self.batch, self.rank, self.world, self.device = inp, batch, rank, world, device
        self.repeat = repeat
        # keys AND row offsets carry the model's key type (solver.i64_input_key), as the
        # reference's SparseTensor<TypeKey> does
        self.key_dtype = torch.int64 if i64_key else torch.int32
        # two conventions exist in the reference:
        # * the multi-hot async reader (AsyncParam, split_batch.cu:28-66): the label word is ALWAYS
        #   an int32 (cast to float), dense words are floats when is_dense_float else ints fed
        #   through log(x + 1) -- the MLPerf raw files (samples/dlrm/preprocessing/convert_to_raw.py);
        # * files of the data generator (data_generator.hpp:977-1040): float_label_dense decides
        #   for label AND dense together.
        if async_param is not None and getattr(async_param, "multi_hot_reader", True):
            self.float_label, self.float_dense = False, bool(async_param.is_dense_float)
        else:
            self.float_label = self.float_dense = bool(float_label_dense)
        self.hot = []  # static hotness per slot, over all sparse params
        for p in inp.sparse_params:
            h = p.nnz_per_slot
            self.hot += list(h) if isinstance(h, (list, tuple)) else [int(h)] * p.slot_num
        self.width = inp.label_dim + inp.dense_dim + sum(self.hot)
        words = os.path.getsize(path) // 4
        n = words // self.width
        if num_samples:
            n = min(n, int(num_samples))
        if n < batch:
            raise RuntimeError("raw file holds fewer samples than one batch")
        self.data = np.memmap(path, dtype="<u4", mode="r", shape=(n, self.width))
        self.n, self._pos = n, 0
        ssa = list(slot_size_array)
        self.slot_offsets = (np.concatenate([[0], np.cumsum(ssa)[:-1]]).astype(np.int64) if ssa
                             else np.zeros(len(self.hot), np.int64))
        # embedding_collection models: Model.compile names, per collection, the inputs of its
        # lookups in lookup order; the reader then hands over the collection's global
        # feature-major CSR (raw keys, bucket = lookup * batch + sample) ready-made -- cutting it
        # out of 26 per-input tensors on the GPU costs ~100 tiny launches per step
        self.ebc_groups = []
        self._col_of = {}
        col, s0 = inp.label_dim + inp.dense_dim, 0
        for p in inp.sparse_params:
            hot = self.hot[s0:s0 + p.slot_num]
            self._col_of[p.top_name] = (col, sum(hot), p.slot_num)
            col += sum(hot)
            s0 += p.slot_num

    def next_batch(self):
        B = self.batch
        if self._pos + B > self.n:  # incomplete tail dropped, as the reference does
            if not self.repeat:
                return None
            self._pos = 0
        blk = np.asarray(self.data[self._pos:self._pos + B])
        self._pos += B
        L, Dn = self.inp.label_dim, self.inp.dense_dim
        bpg = B // self.world
        sl = slice(self.rank * bpg, (self.rank + 1) * bpg)
        label = (blk[:, :L].view(np.float32) if self.float_label
                 else blk[:, :L].view(np.int32).astype(np.float32))
        dense = (blk[:, L:L + Dn].view(np.float32) if self.float_dense
                 else np.log(blk[:, L:L + Dn].view(np.int32).astype(np.float32) + np.float32(1.0)))
        out = {"label": torch.from_numpy(np.ascontiguousarray(label[sl])).to(self.device),
               "dense": torch.from_numpy(np.ascontiguousarray(dense[sl])).to(self.device),
               "sparse": {}}
        col, s0 = L + Dn, 0
        for p in self.inp.sparse_params:
            hot = self.hot[s0:s0 + p.slot_num]
            w = sum(hot)
            keys = blk[:, col:col + w].astype(np.int64)
            offs = np.repeat(self.slot_offsets[s0:s0 + p.slot_num], hot)
            keys = (keys + offs[None, :]).reshape(-1)
            ro = np.concatenate([[0], np.cumsum(np.tile(hot, B))]).astype(np.int64)
            col += w
            s0 += p.slot_num
            out["sparse"][p.top_name] = (torch.from_numpy(ro).to(self.device, self.key_dtype),
                                         torch.from_numpy(keys).to(self.device, self.key_dtype))
        if self.ebc_groups:
            out["ebc"] = []
            for names in self.ebc_groups:
                ks, lens = [], []
                for n in names:
                    c0, w, slots = self._col_of[n]
                    assert slots == 1, "embedding_collection inputs carry one slot per lookup"
                    ks.append(blk[:, c0:c0 + w].astype(np.int64).reshape(-1))
                    lens.append(np.full(B, w, np.int64))
                gbr = np.zeros(len(names) * B + 1, np.int64)
                np.cumsum(np.concatenate(lens), out=gbr[1:])
                out["ebc"].append((torch.from_numpy(np.concatenate(ks)).to(self.device),
                                   torch.from_numpy(gbr).to(self.device)))
        return out


class _Readers:
    def __init__(self, train, evalr):
        self.train, self.evalr = train, evalr

    def next_batch(self, train: bool):
        r = self.train if train else self.evalr
        return None if r is None else r.next_batch()

    def has_eval(self) -> bool:
        return self.evalr is not None


def _resolve(path: str) -> str:
    """a dataset path of a script as is, or -- when it does not exist and HCTR_DATA_ROOT is set --
    under that root (scripts of the reference name absolute container paths like
    /data/train_data.bin, R/test/embedding_collection_test/dgx_a100_one_hot.py:236-237)"""
    root = os.environ.get("HCTR_DATA_ROOT")
    if path and root and not os.path.exists(path):
        cand = os.path.join(root, path.lstrip("/"))
        if os.path.exists(cand):
            return cand
    return path


def make_reader(rp, inp, solver, rank, world, device):
    fmt = getattr(rp.data_reader_type, "name", str(rp.data_reader_type))
    import copy
    rp = copy.copy(rp)
    rp.source = [_resolve(p) for p in rp.source]
    rp.eval_source = _resolve(rp.eval_source)
    if fmt == "RawAsync":
        train = RawReader(rp.source[0], inp, rp.slot_size_array, solver.batchsize, rank, world,
                          device, rp.num_samples, rp.float_label_dense, solver.repeat_dataset,
                          rp.async_param, solver.i64_input_key)
        evalr = None
        if rp.eval_source and os.path.exists(rp.eval_source) and solver.batchsize_eval > 0:
            evalr = RawReader(rp.eval_source, inp, rp.slot_size_array, solver.batchsize_eval, rank,
                              world, device, rp.eval_num_samples, rp.float_label_dense, True,
                              rp.async_param, solver.i64_input_key)
        return _Readers(train, evalr)
    if fmt != "Parquet":
        # the reference's Python path rejects Norm/Raw as deprecated (add_input.cpp:318-325)
        raise RuntimeError(f"DataReaderType_t.{fmt} is deprecated in the reference and not supported; "
                           "use Parquet")
    train = ParquetReader(rp.source[0], inp, rp.slot_size_array, solver.batchsize, rank, world,
                          device, solver.i64_input_key, solver.repeat_dataset)
    evalr = None
    if rp.eval_source and os.path.exists(rp.eval_source) and solver.batchsize_eval > 0:
        evalr = ParquetReader(rp.eval_source, inp, rp.slot_size_array, solver.batchsize_eval, rank,
                              world, device, solver.i64_input_key, True)
    return _Readers(train, evalr)
