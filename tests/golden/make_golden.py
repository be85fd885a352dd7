"""Generates the committed fixtures in tests/golden/*.npz.

PROVENANCE (read before trusting these as "reference outputs"): the reference (NVIDIA-Merlin/
HugeCTR, CUDA) cannot be built or imported in the build container (every translation unit needs
cublas/curand/nvml/mpi/nccl), and it ships no golden vectors for this path: its tests draw random
data and compare GPU vs CPU in-process.  These fixtures are therefore NOT outputs of the reference.
They come from a SECOND, independent restatement of the algorithms written below in plain
numpy / pure Python straight from the cited reference sources -- deliberately sharing no code with
oracle/hctr_oracle.c.  tests/test_golden_cpu.py checks the C oracle against them (two independent
readings of the reference agreeing), tests/test_golden_gpu.py checks the HIP path against them.

Run:  python tests/golden/make_golden.py          (rewrites the .npz files; deterministic)
"""
import os
import struct

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
f32 = np.float32


# ---- MurmurHash3_x86_32 (public domain, Austin Appleby) as used by R/HugeCTR/include/hashtable/
#      cudf/hash_functions.cuh:30-118 on the raw key bytes, seed 0 ----------------------------------
def rotl32(x, r):
    return ((x << r) | (x >> (32 - r))) & 0xFFFFFFFF


def murmur3_32(data: bytes, seed: int = 0) -> int:
    c1, c2 = 0xCC9E2D51, 0x1B873593
    h = seed
    nblocks = len(data) // 4
    for i in range(nblocks):
        k = struct.unpack_from("<I", data, 4 * i)[0]
        k = (k * c1) & 0xFFFFFFFF
        k = rotl32(k, 15)
        k = (k * c2) & 0xFFFFFFFF
        h ^= k
        h = rotl32(h, 13)
        h = (h * 5 + 0xE6546B64) & 0xFFFFFFFF
    tail = data[4 * nblocks:]
    k = 0
    if len(tail) >= 3:
        k ^= tail[2] << 16
    if len(tail) >= 2:
        k ^= tail[1] << 8
    if len(tail) >= 1:
        k ^= tail[0]
        k = (k * c1) & 0xFFFFFFFF
        k = rotl32(k, 15)
        k = (k * c2) & 0xFFFFFFFF
        h ^= k
    h ^= len(data)
    h ^= h >> 16
    h = (h * 0x85EBCA6B) & 0xFFFFFFFF
    h ^= h >> 13
    h = (h * 0xC2B2AE35) & 0xFFFFFFFF
    h ^= h >> 16
    return h


class PyHashTable:
    """nv_hashtable.cu:61-83,169-186 + concurrent_unordered_map.cuh:520-640, executed sequentially
    in array order: slots = (size_t)(capacity / 0.75f); start = hash % slots; linear probing;
    an unseen key takes row = counter++."""

    def __init__(self, capacity, key_bytes):
        self.slots = int(f32(capacity) / f32(0.75))  # float division, truncated
        self.keys = [None] * self.slots
        self.vals = [0] * self.slots
        self.counter = 0
        self.kb = key_bytes

    def _hash(self, key):
        b = struct.pack("<q", key) if self.kb == 8 else struct.pack("<I", key & 0xFFFFFFFF)
        return murmur3_32(b, 0)

    def get_insert(self, keys):
        out = []
        for k in keys:
            i = self._hash(int(k)) % self.slots
            while True:
                if self.keys[i] is None:
                    self.keys[i], self.vals[i] = int(k), self.counter
                    self.counter += 1
                    break
                if self.keys[i] == int(k):
                    break
                i = (i + 1) % self.slots
            out.append(self.vals[i])
        return np.array(out, dtype=np.uint64)

    def get_mark(self, keys):
        out = []
        for k in keys:
            i = self._hash(int(k)) % self.slots
            v = 0xFFFFFFFFFFFFFFFF  # std::numeric_limits<size_t>::max(), nv_hashtable.hpp:33
            while self.keys[i] is not None:
                if self.keys[i] == int(k):
                    v = self.vals[i]
                    break
                i = (i + 1) % self.slots
            out.append(v)
        return np.array(out, dtype=np.uint64)


# ---- forward / backward / update (forward_per_gpu_functor.cu:28-241, backward_functor.cu:26-104,
#      sparse_optimizer.cu:189-237,379-408,497-518) ---------------------------------------------------
def forward(ro, vi, table, combiner):
    B = len(ro) - 1
    D = table.shape[1]
    out = np.zeros((B, D), dtype=f32)
    for b in range(B):
        acc = np.zeros(D, dtype=f32)
        for j in range(ro[b], ro[b + 1]):
            if vi[j] != 0xFFFFFFFFFFFFFFFF:
                acc = (acc + table[int(vi[j])]).astype(f32)
        n = ro[b + 1] - ro[b]
        if combiner == 1 and n > 1:
            acc = (acc * (f32(1.0) / f32(n))).astype(f32)
        out[b] = acc
    return out


def wgrad_of(ro, top_grad, combiner):
    w = top_grad.astype(f32).copy()
    for b in range(len(ro) - 1):
        n = ro[b + 1] - ro[b]
        if combiner == 1 and n > 1:
            w[b] = (w[b] * (f32(1.0) / f32(n))).astype(f32)
    return w


def row_sums(ro, vi, wgrad, scaler):
    """stable sort by row, then per row the float32 sum in ascending bucket order, / scaler"""
    pairs = []
    for b in range(len(ro) - 1):
        for j in range(ro[b], ro[b + 1]):
            pairs.append((int(vi[j]), b))
    pairs.sort(key=lambda p: p[0])  # Python's sort is stable
    sums = {}
    order = []
    for row, b in pairs:
        if row not in sums:
            sums[row] = np.zeros(wgrad.shape[1], dtype=f32)
            order.append(row)
        sums[row] = (sums[row] + wgrad[b]).astype(f32)
    return [(r, (sums[r] / f32(scaler)).astype(f32)) for r in order]


def sgd(table, sums, lr):
    t = table.copy()
    for r, g in sums:
        t[r] = (t[r] + (-f32(lr)) * g).astype(f32)
    return t


def adam_local(table, m, v, sums, lr, b1, b2, eps, times):
    t, m, v = table.copy(), m.copy(), v.copy()
    bias = f32(np.sqrt(1.0 - float(b2) ** times) / (1.0 - float(b1) ** times))
    alpha = f32(lr) * bias
    b1, b2, eps = f32(b1), f32(b2), f32(eps)
    for r, g in sums:
        mi = (b1 * m[r] + (f32(1) - b1) * g).astype(f32)
        vi = (b2 * v[r] + ((f32(1) - b2) * g).astype(f32) * g).astype(f32)
        m[r], v[r] = mi, vi
        t[r] = (t[r] + (-alpha) * mi / (np.sqrt(vi).astype(f32) + eps)).astype(f32)
    return t, m, v


def adagrad(table, acc, sums, lr, eps):
    t, acc = table.copy(), acc.copy()
    for r, g in sums:
        a = (acc[r] + g * g).astype(f32)
        acc[r] = a
        t[r] = (t[r] + (-f32(lr)) * g / (np.sqrt(a).astype(f32) + f32(eps))).astype(f32)
    return t, acc


# ---- InteractionLayer (interaction_layer.cu:1046-1237; CPU check in
#      R/test/utest/layers/interaction_layer_test.cpp) ------------------------------------------------
def interaction_fwd(mlp, emb):
    B, W = mlp.shape
    n = emb.shape[1] + 1
    x = np.concatenate([mlp[:, None, :], emb], axis=1).astype(np.float64)
    out = np.zeros((B, W + n * (n - 1) // 2 + 1), dtype=np.float64)
    out[:, :W] = mlp
    for b in range(B):
        m = x[b] @ x[b].T
        k = W
        for i in range(n):
            for j in range(i):  # strict lower triangle, row-major
                out[b, k] = m[i, j]
                k += 1
    return out  # trailing pad column stays 0


def interaction_bwd(mlp, emb, g):
    B, W = mlp.shape
    n = emb.shape[1] + 1
    x = np.concatenate([mlp[:, None, :], emb], axis=1).astype(np.float64)
    dx = np.zeros_like(x)
    for b in range(B):
        dm = np.zeros((n, n))
        k = W
        for i in range(n):
            for j in range(i):
                dm[i, j] = g[b, k]
                k += 1
        dx[b] = (dm + dm.T) @ x[b]
        dx[b, 0] += g[b, :W]
    return dx[:, 0, :], dx[:, 1:, :]


# ---- MultiCross v1 (multi_cross_layer.cu:582-700; R/test/utest/layers/multi_cross_layer_test.cpp)
def cross_v1_fwd(x0, kernels, biases):
    x = x0.astype(np.float64)
    x0d = x.copy()
    for k, b in zip(kernels, biases):
        x = x0d * (x @ k.astype(np.float64))[:, None] + b.astype(np.float64) + x
    return x


def make_csr(rng, B, S, hot, vps, one_hot):
    ro = [0]
    keys = []
    for b in range(B):
        for s in range(S):
            n = 1 if one_hot else int(rng.integers(0, hot + 1))
            for _ in range(n):
                keys.append(s * vps + int(rng.integers(0, vps)))
            ro.append(len(keys))
    return np.array(ro, dtype=np.int64), np.array(keys, dtype=np.int64)


def main():
    rng = np.random.default_rng(20250919)

    # 1. hash / index stage: two train batches (insert), one eval batch (misses)
    for kb in (8, 4):
        cap = 300
        ht = PyHashTable(cap, kb)
        hi = 2**40 if kb == 8 else 2**31
        pool = rng.integers(0, hi, size=260, dtype=np.int64)
        b1 = pool[rng.integers(0, 180, size=400)]
        b2 = pool[rng.integers(0, 260, size=400)]
        ev = np.concatenate([pool[rng.integers(0, 260, size=100)],
                             rng.integers(0, hi, size=100, dtype=np.int64)])
        pack = (lambda k: struct.pack("<q", int(k))) if kb == 8 else \
            (lambda k: struct.pack("<I", int(k) & 0xFFFFFFFF))
        np.savez(os.path.join(HERE, f"hash_index_k{kb}.npz"), capacity=cap, key_bytes=kb,
                 slots=ht.slots, batch1=b1, batch2=b2, eval=ev,
                 hash1=np.array([murmur3_32(pack(k)) for k in b1], dtype=np.uint32),
                 vi1=ht.get_insert(b1), vi2=ht.get_insert(b2), vi_eval=ht.get_mark(ev),
                 size=ht.counter)

    # 2. embedding forward + update, ragged multi-hot mean and one-hot sum, three optimizers
    for name, D, comb, one_hot in (("mean_multihot", 16, 1, False), ("sum_onehot", 128, 0, True)):
        B, S, hot, vps = 24, 5, 4, 12  # small vocabulary -> many duplicate rows per batch
        ro, keys = make_csr(rng, B, S, hot, vps, one_hot)
        ht = PyHashTable(S * vps, 8)
        vi = ht.get_insert(keys)
        V = S * vps
        table = rng.uniform(-0.05, 0.05, size=(V, D)).astype(f32)
        g = rng.standard_normal((B * S, D)).astype(f32)
        out = forward(ro, vi, table, comb)
        wg = wgrad_of(ro, g, comb)
        scaler = 4.0
        sums = row_sums(ro, vi, wg, scaler)
        t_sgd = sgd(table, sums, 0.05)
        m0 = rng.uniform(-0.01, 0.01, size=(V, D)).astype(f32)
        v0 = rng.uniform(0.0, 0.01, size=(V, D)).astype(f32)
        t_adam, m1, v1 = adam_local(table, m0, v0, sums, 0.01, 0.9, 0.999, 1e-7, times=1)
        a0 = rng.uniform(0.0, 0.1, size=(V, D)).astype(f32)
        t_ada, a1 = adagrad(table, a0, sums, 0.05, 1e-6)
        np.savez(os.path.join(HERE, f"embedding_{name}.npz"), B=B, S=S, D=D, combiner=comb, V=V,
                 row_offset=ro, keys=keys, value_index=vi, table=table, top_grad=g, out=out,
                 wgrad=wg, scaler=scaler, sgd_lr=0.05, table_sgd=t_sgd,
                 adam=np.array([0.01, 0.9, 0.999, 1e-7, 1]), m0=m0, v0=v0, table_adam=t_adam,
                 m1=m1, v1=v1, adagrad=np.array([0.05, 1e-6]), a0=a0, table_adagrad=t_ada, a1=a1)

    # 3. interaction (DLRM shape n_emb=26, W=128, and a small odd shape)
    for name, B, n_emb, W in (("dlrm", 6, 26, 128), ("small", 5, 3, 8)):
        mlp = rng.standard_normal((B, W)).astype(f32)
        emb = rng.standard_normal((B, n_emb, W)).astype(f32)
        out = interaction_fwd(mlp, emb)
        g = rng.standard_normal(out.shape).astype(f32)
        g[:, -1] = 0
        dmlp, demb = interaction_bwd(mlp, emb, g.astype(np.float64))
        np.savez(os.path.join(HERE, f"interaction_{name}.npz"), mlp=mlp, emb=emb, out=out,
                 top_grad=g, dmlp=dmlp, demb=demb)

    # 4. cross v1 forward
    B, w, L = 7, 24, 3
    x0 = rng.standard_normal((B, w)).astype(f32)
    ks = rng.standard_normal((L, w)).astype(f32) * f32(0.3)
    bs = rng.standard_normal((L, w)).astype(f32) * f32(0.1)
    np.savez(os.path.join(HERE, "cross_v1.npz"), x0=x0, kernels=ks, biases=bs,
             out=cross_v1_fwd(x0, ks, bs))
    print("wrote", sorted(f for f in os.listdir(HERE) if f.endswith(".npz")))


if __name__ == "__main__":
    main()
