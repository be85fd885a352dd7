"""CPU: pin the oracle.  The reference holds no golden vectors for this path, so the
oracle is pinned by (1) public MurmurHash3_x86_32 known-answer vectors, (2) facts read off the
reference sources (table sizing, sentinels, layouts), (3) closed-form optimizer steps and
hand-worked small cases."""
import ctypes
import math
import struct

import numpy as np
import pytest


# MurmurHash3_x86_32 known answers (public-domain reference implementation / SMHasher vectors)
MURMUR_KATS = [
    (b"", 0, 0x00000000),
    (b"", 1, 0x514E28B7),
    (b"", 0xFFFFFFFF, 0x81F16F39),
    (b"\xff\xff\xff\xff", 0, 0x76293B50),
    (b"\x21\x43\x65\x87", 0, 0xF55B516B),
    (b"\x21\x43\x65\x87", 0x5082EDEE, 0x2362F9DE),
    (b"\x21\x43\x65", 0, 0x7E4A8634),
    (b"\x21\x43", 0, 0xA0F7B07A),
    (b"\x21", 0, 0x72661CF4),
    (b"\x00\x00\x00\x00", 0, 0x2362F9DE),
    (b"\x00\x00\x00", 0, 0x85F0B427),
    (b"\x00\x00", 0, 0x30F4C306),
    (b"\x00", 0, 0x514E28B7),
    (b"aaaa", 0x9747B28C, 0x5A97808A),
    (b"aaa", 0x9747B28C, 0x283E0130),
    (b"aa", 0x9747B28C, 0x5D211726),
    (b"a", 0x9747B28C, 0x7FA09EA6),
    (b"abcd", 0x9747B28C, 0xF0478627),
    (b"abc", 0x9747B28C, 0xC84A62DD),
    (b"ab", 0x9747B28C, 0x74875592),
    (b"Hello, world!", 0x9747B28C, 0x24884CBA),
    (b"The quick brown fox jumps over the lazy dog", 0x9747B28C, 0x2FA826CD),
]


@pytest.mark.parametrize("data,seed,expect", MURMUR_KATS)
def test_murmur3_kat(oracle, data, seed, expect):
    assert oracle.murmur3_32(data, seed) == expect


def test_hash_key_is_murmur_of_raw_key_bytes(oracle):
    # hash_functions.cuh:66-107: len = sizeof(Key), seed 0, little-endian key bytes
    for k in [0, 1, 7, 12345, 2**31 - 1, 2**32 - 2]:
        assert int(oracle.hash_keys([k], 4)[0]) == oracle.murmur3_32(struct.pack("<I", k), 0)
    for k in [0, 1, 7, 12345, 2**40 + 3, 2**62]:
        assert int(oracle.hash_keys([k], 8)[0]) == oracle.murmur3_32(struct.pack("<q", k), 0)
    # 4 zero bytes seed 0 is a published vector
    assert int(oracle.hash_keys([0], 4)[0]) == 0x2362F9DE


def test_hashtable_sizing_float_division(oracle):
    # nv_hashtable.cu:178: static_cast<size_t>(capacity / 0.75f) -- FLOAT arithmetic
    for cap in [3, 100, 2600, 100000, 187767399, 39884406]:
        ht = oracle.HashTable(cap, 8) if cap < 10**6 else None
        expect = int(np.float32(cap) / np.float32(0.75))
        if ht is not None:
            assert ht.table_size() == expect
    assert int(np.float32(187767399) / np.float32(0.75)) != 187767399 * 4 // 3  # float rounding matters


def test_hashtable_first_occurrence_order(oracle):
    ht = oracle.HashTable(16, 8)
    keys = np.array([42, 7, 42, 1000, 7, 5, 42], dtype=np.int64)
    idx = ht.get_insert(keys)
    assert idx.tolist() == [0, 1, 0, 2, 1, 3, 0]
    assert ht.value_head() == 4 and ht.size() == 4
    # second batch: known keys keep their rows, new keys continue the count
    idx2 = ht.get_insert(np.array([5, 99, 42, 99], dtype=np.int64))
    assert idx2.tolist() == [3, 4, 0, 4]
    # get_mark: miss -> SIZE_MAX (nv_hashtable.cu:74-83)
    m = ht.get_mark(np.array([7, 123456], dtype=np.int64))
    assert int(m[0]) == 1 and m[1] == oracle.INVALID
    k, v = ht.dump()
    assert dict(zip(k.tolist(), v.tolist())) == {42: 0, 7: 1, 1000: 2, 5: 3, 99: 4}


def test_hashtable_linear_probing_collisions(oracle):
    # tiny table forces collisions; every key must still resolve to a unique row
    ht = oracle.HashTable(6, 4)  # 8 physical slots
    assert ht.table_size() == 8
    keys = np.arange(6, dtype=np.int64) * 8 + 3
    idx = ht.get_insert(keys)
    assert idx.tolist() == list(range(6))
    assert ht.get_insert(keys[::-1].copy()).tolist() == list(range(5, -1, -1))


def test_forward_sum_mean_handworked(oracle):
    table = np.array([[1, 2], [10, 20], [100, 200], [1000, 2000]], dtype=np.float32)
    ro = np.array([0, 2, 2, 3, 6], dtype=np.int64)  # buckets: {0,1}, {}, {2}, {3,0,INVALID}
    vi = np.array([0, 1, 2, 3, 0, oracle.INVALID], dtype=np.uint64)
    s = oracle.forward(ro, vi, table, 2, 0)
    assert s.tolist() == [[11, 22], [0, 0], [100, 200], [1001, 2002]]
    m = oracle.forward(ro, vi, table, 2, 1)
    # mean: n=2 -> /2 ; n=0 -> 0 ; n=1 -> unscaled ; n=3 (miss still counts) -> /3
    np.testing.assert_allclose(m, [[5.5, 11], [0, 0], [100, 200], [1001 / 3, 2002 / 3]], rtol=1e-6)
    g = np.arange(8, dtype=np.float32).reshape(4, 2)
    w = oracle.backward(ro, g, 2, 1)
    np.testing.assert_allclose(w, g * np.array([[0.5], [1], [1], [1 / 3]], dtype=np.float32), rtol=1e-6)
    assert (oracle.backward(ro, g, 2, 0) == g).all()


def test_localized_filter_and_reorder(oracle):
    # 2 samples x 5 slots on 2 GPUs: gpu0 owns slots 0,2,4 ; gpu1 owns 1,3
    ro = np.array([0, 1, 3, 3, 4, 6, 7, 7, 9, 10, 10], dtype=np.int64)
    keys = np.arange(10, dtype=np.int64) + 100
    r0, k0 = oracle.localized_filter(ro, keys, 2, 5, 0, 2)
    r1, k1 = oracle.localized_filter(ro, keys, 2, 5, 1, 2)
    assert r0.tolist() == [0, 1, 1, 3, 4, 6, 6] and k0.tolist() == [100, 104, 105, 106, 107, 108]
    assert r1.tolist() == [0, 2, 3, 3, 4] and k1.tolist() == [101, 102, 103, 109]
    assert oracle.slots_on_gpu(5, 0, 2) == 3 and oracle.slots_on_gpu(5, 1, 2) == 2
    # reorder: recv buffer [gpu][b][slot_in_gpu][D] -> [b][slot][D]; bpg = 1 sample per GPU
    D = 2
    recv = np.arange((3 + 2) * D, dtype=np.float32)  # gpu0 block: slots 0,2,4 ; gpu1 block: 1,3
    out = oracle.forward_reorder(recv, 1, 5, D, 2)
    assert out[0, :, 0].tolist() == [0, 6, 2, 8, 4]
    back = oracle.backward_reorder(out, 1, 5, D, 2)
    assert (back == recv).all()


def _opt(oracle, **kw):
    o = oracle.OptParamsC()
    o.optimizer = kw.get("optimizer", oracle.OPT_SGD)
    o.update_type = kw.get("update_type", oracle.UPDATE_LOCAL)
    o.lr = kw.get("lr", 0.1)
    o.beta1, o.beta2, o.epsilon = kw.get("beta1", 0.9), kw.get("beta2", 0.999), kw.get("epsilon", 1e-7)
    o.momentum_factor = kw.get("momentum_factor", 0.9)
    o.scaler = kw.get("scaler", 1.0)
    o.times = kw.get("times", 1)
    return o


def test_update_sgd_closed_form_and_sort_equivalence(oracle):
    rng = np.random.default_rng(0)
    D, vocab, buckets = 4, 7, 40
    ro = np.arange(buckets + 1, dtype=np.int64)
    vi = rng.integers(0, vocab, size=buckets).astype(np.uint64)
    g = rng.standard_normal((buckets, D)).astype(np.float32)
    t0 = rng.standard_normal((vocab, D)).astype(np.float32)
    ta, tb = t0.copy(), t0.copy()
    na = oracle.update_params(ro, vi, g, _opt(oracle, lr=0.5, scaler=2.0), ta, fast_sort=True)
    nb = oracle.update_params(ro, vi, g, _opt(oracle, lr=0.5, scaler=2.0), tb, fast_sort=False)
    assert na == nb == len(np.unique(vi))
    assert (ta == tb).all()  # stable merge sort == the reference's odd-even transposition sort
    ref = t0.astype(np.float64)
    for r in range(vocab):
        ref[r] -= 0.5 * g[vi == r].astype(np.float64).sum(0) / 2.0
    np.testing.assert_allclose(ta, ref, rtol=1e-5, atol=1e-6)


def test_update_adam_first_step_closed_form(oracle):
    # t = 1: m = (1-b1) g, v = (1-b2) g^2, alpha_1 = lr*sqrt(1-b2)/(1-b1)
    # => dw = -lr * g / (|g| + eps/sqrt(1-b2))  ~= -lr * sign(g)
    D, vocab = 3, 2
    ro = np.array([0, 1, 2], dtype=np.int64)
    vi = np.array([1, 0], dtype=np.uint64)
    g = np.array([[0.5, -2.0, 1e-3], [3.0, -1.0, 0.25]], dtype=np.float32)
    t = np.zeros((vocab, D), dtype=np.float32)
    m = np.zeros_like(t)
    v = np.zeros_like(t)
    oracle.update_params(ro, vi, g, _opt(oracle, optimizer=oracle.OPT_ADAM, lr=0.01, times=1), t, m, v)
    expect = lambda gg: -0.01 * gg / (np.abs(gg) + 1e-7 / math.sqrt(1 - 0.999))
    np.testing.assert_allclose(t[1], expect(g[0].astype(np.float64)), rtol=1e-4)
    np.testing.assert_allclose(t[0], expect(g[1].astype(np.float64)), rtol=1e-4)
    np.testing.assert_allclose(m[1], 0.1 * g[0], rtol=1e-6)
    np.testing.assert_allclose(v[0], 0.001 * g[1] ** 2, rtol=1e-4)


def test_update_adagrad_momentum_nesterov_closed_form(oracle):
    ro = np.array([0, 1], dtype=np.int64)
    vi = np.array([0], dtype=np.uint64)
    g = np.array([[2.0, -4.0]], dtype=np.float32)
    # AdaGrad: accum = g^2 ; w -= lr * g / (sqrt(accum) + eps) = lr * sign(g)
    t = np.zeros((1, 2), np.float32); a = np.zeros_like(t)
    oracle.update_params(ro, vi, g, _opt(oracle, optimizer=oracle.OPT_ADAGRAD, lr=0.1, epsilon=0.0), t, a)
    np.testing.assert_allclose(t, [[-0.1, 0.1]], rtol=1e-6); np.testing.assert_allclose(a, g * g)
    # Momentum local: mo = f*mo - lr*g ; w += mo
    t = np.zeros((1, 2), np.float32); mo = np.full_like(t, 1.0)
    oracle.update_params(ro, vi, g, _opt(oracle, optimizer=oracle.OPT_MOMENTUM, lr=0.1, momentum_factor=0.5), t, mo)
    np.testing.assert_allclose(mo, [[0.5 - 0.2, 0.5 + 0.4]], rtol=1e-6); np.testing.assert_allclose(t, mo)
    # Nesterov local: new = mu*old - lr*g ; w += -mu*old + (1+mu)*new
    t = np.zeros((1, 2), np.float32); ac = np.full_like(t, 1.0)
    oracle.update_params(ro, vi, g, _opt(oracle, optimizer=oracle.OPT_NESTEROV, lr=0.1, momentum_factor=0.5), t, ac)
    new = 0.5 - 0.1 * g
    np.testing.assert_allclose(ac, new, rtol=1e-6)
    np.testing.assert_allclose(t, -0.5 + 1.5 * new, rtol=1e-6)


def test_update_adam_global_touches_every_row(oracle):
    # the global sweep decays and applies ALL max_vocab rows, also never-touched ones
    D, vocab = 2, 3
    ro = np.array([0, 1], dtype=np.int64)
    vi = np.array([1], dtype=np.uint64)
    g = np.array([[1.0, -1.0]], dtype=np.float32)
    t = np.zeros((vocab, D), np.float32)
    m = np.full((vocab, D), 0.5, np.float32)
    v = np.full((vocab, D), 0.25, np.float32)
    oracle.update_params(ro, vi, g, _opt(oracle, optimizer=oracle.OPT_ADAM, update_type=oracle.UPDATE_GLOBAL, lr=0.1, times=3), t, m, v)
    bias = math.sqrt(1 - 0.999 ** 3) / (1 - 0.9 ** 3)
    # untouched rows 0, 2: m *= b1, v *= b2, w -= alpha * m / (sqrt(v)+eps)
    np.testing.assert_allclose(m[0], 0.45, rtol=1e-6)
    np.testing.assert_allclose(t[0], -0.1 * bias * 0.45 / (math.sqrt(0.25 * 0.999) + 1e-7), rtol=1e-5)
    # touched row: m += (1-b1) g / b1 first, then decays => m = b1*0.5 + (1-b1) g
    np.testing.assert_allclose(m[1], 0.9 * 0.5 + 0.1 * g[0], rtol=1e-5)


def test_interaction_layout_handworked(oracle):
    # n_emb = 2, W = 2: x0 = mlp, x1, x2 ; out = [mlp | (1,0) (2,0) (2,1) | 0]
    mlp = np.array([[1.0, 2.0]], np.float32)
    emb = np.array([[[3.0, 4.0], [5.0, 6.0]]], np.float32)
    out = oracle.interaction_fwd(mlp, emb)
    assert out.tolist() == [[1, 2, 3 + 8, 5 + 12, 15 + 24, 0]]
    # bprop: grads on pairs p10, p20, p21 ; dX0 = p10*x1 + p20*x2 (+ passthrough), dX1 = p10*x0 + p21*x2
    g = np.array([[0.1, 0.2, 1.0, 10.0, 100.0, 7.0]], np.float32)
    mg, eg = oracle.interaction_bwd(mlp, emb, g)
    np.testing.assert_allclose(mg, [[0.1 + 1 * 3 + 10 * 5, 0.2 + 1 * 4 + 10 * 6]], rtol=1e-6)
    np.testing.assert_allclose(eg[0, 0], [1 * 1 + 100 * 5, 1 * 2 + 100 * 6], rtol=1e-6)
    np.testing.assert_allclose(eg[0, 1], [10 * 1 + 100 * 3, 10 * 2 + 100 * 4], rtol=1e-6)


def test_cross_v1_matches_formula_and_numeric_gradient(oracle):
    rng = np.random.default_rng(1)
    B, w, L = 3, 5, 2
    x0 = rng.standard_normal((B, w)).astype(np.float32) * 0.5
    k = rng.standard_normal((L, w)).astype(np.float32) * 0.5
    b = rng.standard_normal((L, w)).astype(np.float32) * 0.1
    outs, hid = oracle.cross_v1_fwd(x0, k, b)
    xl = x0.astype(np.float64)
    for l in range(L):
        xl = x0 * (xl @ k[l])[:, None] + b[l] + xl
    np.testing.assert_allclose(outs[-1], xl, rtol=1e-5, atol=1e-6)
    og = rng.standard_normal((B, w)).astype(np.float32)
    ig, kg, bg = oracle.cross_v1_bwd(x0, k, outs, hid, og)
    # numeric gradient of sum(out * og) w.r.t. x0[0,0] and k[0,1]
    def loss(x0_, k_):
        o, _ = oracle.cross_v1_fwd(x0_, k_, b)
        return float((o[-1].astype(np.float64) * og).sum())
    eps = 1e-2
    xp, xm = x0.copy(), x0.copy(); xp[0, 0] += eps; xm[0, 0] -= eps
    np.testing.assert_allclose(ig[0, 0], (loss(xp, k) - loss(xm, k)) / (2 * eps), rtol=2e-2, atol=1e-3)
    kp, km = k.copy(), k.copy(); kp[0, 1] += eps; km[0, 1] -= eps
    np.testing.assert_allclose(kg[0, 1], (loss(x0, kp) - loss(x0, km)) / (2 * eps), rtol=2e-2, atol=1e-3)
    np.testing.assert_allclose(bg[L - 1], og.sum(0), rtol=1e-5)


def test_cross_v2_matches_formula(oracle):
    rng = np.random.default_rng(2)
    B, w, p, L = 4, 6, 3, 2
    x0 = rng.standard_normal((B, w)).astype(np.float32) * 0.3
    U = rng.standard_normal((L, w, p)).astype(np.float32) * 0.3
    V = rng.standard_normal((L, p, w)).astype(np.float32) * 0.3
    b = rng.standard_normal((L, w)).astype(np.float32) * 0.1
    outs, hid, xus = oracle.cross_v2_fwd(x0, U, V, b)
    xl = x0.astype(np.float64)
    for l in range(L):
        xl = x0 * (xl @ U[l] @ V[l] + b[l]) + xl
    np.testing.assert_allclose(outs[-1], xl, rtol=1e-5, atol=1e-6)
    og = rng.standard_normal((B, w)).astype(np.float32)
    ig, dU, dV, db = oracle.cross_v2_bwd(x0, U, V, outs, hid, xus, og)
    def loss(U_):
        o, _, _ = oracle.cross_v2_fwd(x0, U_, V, b)
        return float((o[-1].astype(np.float64) * og).sum())
    eps = 1e-2
    Up, Um = U.copy(), U.copy(); Up[0, 1, 2] += eps; Um[0, 1, 2] -= eps
    np.testing.assert_allclose(dU[0, 1, 2], (loss(Up) - loss(Um)) / (2 * eps), rtol=2e-2, atol=1e-3)


def test_powerlaw_keys_range_and_skew(oracle):
    k = oracle.powerlaw_keys(1234, 20000, 1000, 1.3)
    assert k.min() >= 0 and k.max() < 1000
    # heavy head (round() gives key 0 only half a bin, data_generator.hpp:118-123)
    counts = np.bincount(k, minlength=1000)
    assert counts[:2].min() > counts[2:].max() and counts[:10].sum() > 0.5 * k.size
    assert (oracle.powerlaw_keys(1234, 100, 1000, 1.3) == k[:100]).all()  # seeded, reproducible


def test_round_half_matches_ieee_binary16(oracle):
    """the oracle's software float -> binary16 -> float rounding (fp16 optimizer state)
    against numpy's float16: normals, ties to even, subnormals, overflow to inf, signed zero"""
    import ctypes
    L = oracle.lib()
    L.hco_round_half.restype = ctypes.c_float
    L.hco_round_half.argtypes = [ctypes.c_float]
    rng = np.random.default_rng(0)
    xs = np.concatenate([
        rng.standard_normal(5000).astype(np.float32) * np.float32(10) ** rng.integers(-9, 6, 5000).astype(np.float32),
        np.array([0, -0.0, 65504, 65519.9, 65520, 70000, -65520, 6.1e-5, 6.0e-5, 5.96e-8, 2.98e-8,
                  2.9e-8, 1e-10, 0.1, 1 + 2 ** -11, 1 + 2 ** -11 + 2 ** -20, 1 + 3 * 2 ** -11], np.float32)])
    with np.errstate(over="ignore"):
        want = xs.astype(np.float16).astype(np.float32)
    got = np.array([L.hco_round_half(float(x)) for x in xs], np.float32)
    assert (got.view(np.uint32) == want.view(np.uint32)).all()
