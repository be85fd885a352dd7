"""CPU restatement of the dynamic embedding table (TEST INFRASTRUCTURE ONLY, like everything under
oracle/): det::DynamicEmbeddingTable semantics
(R/third_party/dynamic_embedding_table/dynamic_embedding_table.cu:129-260,
cuCollections/include/cuco/detail/dynamic_map_kernels.cuh:100-260) and the optimizer formulas of
R/HugeCTR/embedding_storage/optimizers.cuh:29-233, as plain Python dicts + numpy float32.
The optimizer steps and the lookup / update flow are PINNED against the reference's own CPU mirror
of the table (R/HugeCTR/embedding_storage/dynamic_embedding_cpu.hpp + optimizers.hpp compiled into
oracle/_ref/libref_det.so; tests/test_ref_det_cpu.py).  The random initializer stays unpinned (the
reference seeds it from std::random_device); constant initializers are exact."""
import numpy as np

f32 = np.float32
FLT_EPSILON = f32(1.1920929e-07)

FTRL, ADAM, RMSPROP, ADAGRAD, NESTEROV, MOMENTUM, SGD = range(7)  # Optimizer_t values


class DetOracle:
    def __init__(self, dims, init_value):
        self.dims = list(dims)
        self.init = f32(init_value)
        self.maps = [dict() for _ in dims]

    def _each(self, keys, id_spaces, offsets):
        for i, c in enumerate(id_spaces):
            for j in range(offsets[i], offsets[i + 1]):
                yield c, int(keys[j])

    def lookup(self, keys, id_spaces, offsets):
        out = []
        for c, k in self._each(keys, id_spaces, offsets):
            if k not in self.maps[c]:  # insert-if-missing with the initializer (lookup kernel)
                self.maps[c][k] = np.full(self.dims[c], self.init, dtype=f32)
            out.append(self.maps[c][k].copy())
        return np.concatenate(out) if out else np.zeros(0, dtype=f32)

    def scatter(self, keys, elements, id_spaces, offsets, add):
        off = 0
        for c, k in self._each(keys, id_spaces, offsets):
            d = self.dims[c]
            if k in self.maps[c]:  # missing keys are skipped
                if add:
                    self.maps[c][k] = (self.maps[c][k] + elements[off:off + d]).astype(f32)
                else:
                    self.maps[c][k] = elements[off:off + d].astype(f32).copy()
            off += d

    def remove(self, keys, id_spaces, offsets):
        for c, k in self._each(keys, id_spaces, offsets):
            self.maps[c].pop(k, None)

    def size_per_class(self):
        return [len(m) for m in self.maps]


def update(weights: DetOracle, states: DetOracle, opt, keys, id_spaces, offsets, ev_start, wgrad,
           lr, scaler=1.0, beta1=0.9, beta2=0.999, eps=1e-7, momentum=0.9, rms_beta=0.9,
           lambda1=0.0, lambda2=0.0, ftrl_beta=0.0, times=1):
    """dynamic_embedding.cu:176-330: state lookup (zeros), per-element formula, scatter_add"""
    lr, scaler = f32(lr), f32(scaler)
    b1, b2, eps, mom, rb = f32(beta1), f32(beta2), f32(eps), f32(momentum), f32(rms_beta)
    # AdamOptHyperParams::bias() (optimizer.hpp:58-60): std::pow(float beta, uint64 times) promotes
    # the FLOAT beta to double -- 0.999f, not 0.999 (found by pinning against the reference build)
    bias = f32(np.sqrt(1.0 - float(b2) ** times) / (1.0 - float(b1) ** times))
    lr_scaled_bias = lr * bias
    l2b = f32(lambda2) + f32(ftrl_beta) / lr
    pos = 0
    for i, c in enumerate(id_spaces):
        d = weights.dims[c]
        for j in range(offsets[i], offsets[i + 1]):
            k = int(keys[j])
            g = (wgrad[ev_start[pos]:ev_start[pos] + d].astype(f32) / scaler).astype(f32)
            pos += 1
            if opt != SGD:
                sd = states.dims[c]
                if k not in states.maps[c]:
                    states.maps[c][k] = np.zeros(sd, dtype=f32)
                st = states.maps[c][k]
            if opt == FTRL and k not in weights.maps[c]:
                weights.maps[c][k] = np.full(d, weights.init, dtype=f32)
            if opt == SGD:
                delta = (-lr * g).astype(f32)
            elif opt == MOMENTUM:
                st[:] = (mom * st - lr * g).astype(f32)
                delta = st.copy()
            elif opt == NESTEROV:
                prev = st.copy()
                st[:] = (mom * prev - lr * g).astype(f32)
                delta = (st + mom * st - mom * prev).astype(f32)
            elif opt == ADAGRAD:
                st[:] = (st + g * g).astype(f32)
                delta = (-lr * g / (np.sqrt(st).astype(f32) + eps)).astype(f32)
            elif opt == RMSPROP:
                st[:] = (rb * st + (f32(1) - rb) * g * g).astype(f32)
                delta = (-lr * g / (np.sqrt(st).astype(f32) + eps)).astype(f32)
            elif opt == ADAM:
                m, v = st[:d], st[d:]
                m[:] = (b1 * m + (f32(1) - b1) * g).astype(f32)
                v[:] = (b2 * v + (f32(1) - b2) * g * g).astype(f32)
                delta = (-lr_scaled_bias * m / (np.sqrt(v).astype(f32) + eps)).astype(f32)
            else:  # FTRL
                n, z = st[:d], st[d:]
                w = weights.maps[c][k]
                n_prev_sqrt = np.sqrt(n + FLT_EPSILON).astype(f32)
                n[:] = (n + g * g).astype(f32)
                n_sqrt = np.sqrt(n + FLT_EPSILON).astype(f32)
                sigma = ((n_sqrt - n_prev_sqrt) / lr).astype(f32)
                z[:] = (z + g - sigma * w).astype(f32)
                p = ((f32(1) - f32(2) * np.signbit(z).astype(f32)) * f32(lambda1) - z).astype(f32)
                q = (n_sqrt / lr + l2b).astype(f32)
                delta = ((p / q) * np.signbit(f32(lambda1) - np.abs(z)).astype(f32) - w).astype(f32)
            if k in weights.maps[c]:  # scatter_add skips missing keys
                weights.maps[c][k] = (weights.maps[c][k] + delta).astype(f32)
