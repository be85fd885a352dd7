"""TEST INFRASTRUCTURE ONLY: randomized differential test of the dense ops on the path
(Interaction forward / backward in fp32, fp16, bf16 incl. the indexed and the gather-fused forms;
Cross v1 forward / backward) -- the kernels' source under the host interpreter (tests/emu, MFMA
modelled lane for lane) against the oracle's fp32 references, over shapes no parametrized test
enumerates (any width, 1-40 embeddings, 1-300 samples, 1-8 cross layers up to width 1000).

    python tests/emu/fuzz_dense.py --seed 0 --cases 100"""
import argparse
import os
import sys
import traceback

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
for p in (ROOT, os.path.join(ROOT, "tests"), HERE):
    if p not in sys.path:
        sys.path.insert(0, p)

import emu  # noqa: E402
from oracle import pyoracle as orc  # noqa: E402
from util import assert_close  # noqa: E402

lib = emu.load(os.environ.get("HCTR_EMU_VARIANT"),
               os.path.basename(os.path.normpath(os.environ["HCTR_EMU_VARIANT"]))
               if os.environ.get("HCTR_EMU_VARIANT") else None)
emu.bind(lib)


def to16(a, dt):
    if dt == 1:
        return a.astype(np.float16)
    return (orc.round_to(a, "bf16").view(np.uint32) >> 16).astype(np.uint16)


def from16(a, dt):
    if dt == 1:
        return a.astype(np.float32)
    return (a.astype(np.uint32) << 16).view(np.float32)


def interaction_case(rng, desc):
    dt = int(rng.choice([0, 1, 2]))
    B = int(rng.choice([1, 2, 31, 64, 65, 130, 300]))
    n_emb = int(rng.choice([1, 2, 3, 7, 13, 26, 31, 40]))
    W = int(rng.choice([16, 32, 64, 128])) if dt else int(rng.choice([1, 3, 8, 16, 24, 32, 64, 128, 200]))
    if dt and n_emb > 31:
        n_emb = 31
    desc.update(op="interaction", dt=dt, B=B, n_emb=n_emb, W=W)
    mlp = rng.standard_normal((B, W)).astype(np.float32)
    emb = rng.standard_normal((B, n_emb, W)).astype(np.float32)
    if dt:
        mlp, emb = from16(to16(mlp, dt), dt), from16(to16(emb, dt), dt)
    n_ins = n_emb + 1
    olen = W + n_ins * (n_ins - 1) // 2 + 1
    m_d, e_d = (mlp, emb) if dt == 0 else (to16(mlp, dt), to16(emb, dt))
    out = np.empty((B, olen), dtype=m_d.dtype)
    emu.check(lib, lib.hctr_interaction_fwd(B, n_emb, W, emu.ptr(m_d), emu.ptr(e_d), emu.ptr(out),
                                            dt, None))
    want = orc.interaction_fwd(mlp, emb)
    tol = 2e-3 if dt == 0 else (2e-2 if dt == 1 else 1.2e-1)
    scale = float(np.abs(want).max()) + 1.0
    got = out if dt == 0 else from16(out, dt)
    assert_close(got, want, tol, tol * scale * 0.05, f"{desc} forward")
    g = rng.standard_normal((B, olen)).astype(np.float32)
    if dt:
        g = from16(to16(g, dt), dt)
    g_d = g if dt == 0 else to16(g, dt)
    mg, eg = np.empty_like(m_d), np.empty_like(e_d)
    emu.check(lib, lib.hctr_interaction_bwd(B, n_emb, W, emu.ptr(m_d), emu.ptr(e_d), emu.ptr(g_d),
                                            emu.ptr(mg), emu.ptr(eg), dt, None))
    wmg, weg = orc.interaction_bwd(mlp, emb, g)
    sc = float(max(np.abs(wmg).max(), np.abs(weg).max())) + 1.0
    assert_close(mg if dt == 0 else from16(mg, dt), wmg, tol, tol * sc * 0.05, f"{desc} mlp grad")
    assert_close(eg if dt == 0 else from16(eg, dt), weg, tol, tol * sc * 0.05, f"{desc} emb grad")
    if dt and W in (16, 32, 64, 128):
        # the gather fused into the interaction == pooled vectors + plain interaction, bit for bit
        V = int(rng.choice([1, 5, 300]))
        table = rng.standard_normal((V, W)).astype(np.float32)
        vi = rng.integers(0, V, size=B * n_emb).astype(np.uint64)
        if rng.random() < 0.3:
            vi[rng.integers(0, vi.size, size=max(1, vi.size // 10))] = np.uint64(0xFFFFFFFFFFFFFFFF)
        pooled = np.empty((B, n_emb, W), dtype=m_d.dtype)
        out1 = np.empty_like(out)
        emu.check(lib, lib.hctr_interaction_fwd_gather(B, n_emb, W, emu.ptr(m_d), emu.ptr(table),
                                                       emu.ptr(vi), emu.ptr(pooled), emu.ptr(out1),
                                                       dt, None))
        rows = np.where(vi == np.uint64(0xFFFFFFFFFFFFFFFF), 0, vi).astype(np.int64)
        src = table[rows]
        src[vi == np.uint64(0xFFFFFFFFFFFFFFFF)] = 0.0
        want_p = to16(src, dt).reshape(B, n_emb, W)
        assert (pooled.view(np.uint16) == want_p.view(np.uint16)).all(), (desc, "pooled")
        out2 = np.empty_like(out)
        emu.check(lib, lib.hctr_interaction_fwd(B, n_emb, W, emu.ptr(m_d), emu.ptr(want_p),
                                                emu.ptr(out2), dt, None))
        assert (out1.view(np.uint16) == out2.view(np.uint16)).all(), (desc, "gather-fused output")
        # indexed form (rows of distinct vectors + an index per (sample, embedding))
        if W >= 32:
            R = int(rng.choice([1, 9, 200]))
            rws = to16(rng.standard_normal((R, W)).astype(np.float32), dt)
            row_of = rng.integers(0, R, size=(B, n_emb)).astype(np.uint32)
            out3 = np.empty_like(out)
            emu.check(lib, lib.hctr_interaction_fwd_indexed(B, n_emb, W, emu.ptr(m_d), emu.ptr(rws),
                                                            emu.ptr(row_of), emu.ptr(out3), dt,
                                                            None))
            dense = np.ascontiguousarray(rws[row_of.astype(np.int64)])
            out4 = np.empty_like(out)
            emu.check(lib, lib.hctr_interaction_fwd(B, n_emb, W, emu.ptr(m_d), emu.ptr(dense),
                                                    emu.ptr(out4), dt, None))
            assert (out3.view(np.uint16) == out4.view(np.uint16)).all(), (desc, "indexed output")


def cross_case(rng, desc):
    B = int(rng.choice([1, 3, 64, 257, 1024]))
    w = int(rng.choice([1, 5, 63, 64, 65, 429, 1000]))
    L = int(rng.integers(1, 9))
    desc.update(op="cross", B=B, w=w, L=L)
    x0 = rng.standard_normal((B, w)).astype(np.float32)
    k = (rng.standard_normal((L, w)) / np.sqrt(w)).astype(np.float32)
    b = (rng.standard_normal((L, w)) * 0.05).astype(np.float32)
    outs = np.empty((L, B, w), np.float32)
    hid = np.empty((L, B), np.float32)
    emu.check(lib, lib.hctr_cross_v1_fwd(B, w, L, emu.ptr(x0), emu.ptr(k), emu.ptr(b),
                                         emu.ptr(outs), emu.ptr(hid), None))
    o2, h2 = orc.cross_v1_fwd(x0, k, b)
    sc = float(np.abs(o2).max()) + 1.0
    assert_close(outs, o2, 1e-4, 1e-5 * sc, f"{desc} forward")
    og = rng.standard_normal((B, w)).astype(np.float32)
    ig, kg, bg = np.empty_like(x0), np.empty_like(k), np.empty_like(k)
    ws = np.empty(lib.hctr_cross_v1_bwd_workspace_bytes(B, w, L) // 4, np.float32)
    emu.check(lib, lib.hctr_cross_v1_bwd(B, w, L, emu.ptr(x0), emu.ptr(k), emu.ptr(o2), emu.ptr(h2),
                                         emu.ptr(og), emu.ptr(ig), emu.ptr(kg), emu.ptr(bg),
                                         emu.ptr(ws), None))
    wi, wk, wb = orc.cross_v1_bwd(x0, k, o2, h2, og)
    for name, got, want in (("dx", ig, wi), ("dw", kg, wk), ("db", bg, wb)):
        assert_close(got, want, 2e-4, 2e-5 * (float(np.abs(want).max()) + 1.0), f"{desc} {name}")


def one_case(seed):
    rng = np.random.default_rng(seed)
    desc = dict(seed=seed)
    if rng.random() < 0.7:
        interaction_case(rng, desc)
    else:
        cross_case(rng, desc)
    return desc


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--seed", type=int, default=0)
    ap.add_argument("--cases", type=int, default=50)
    a = ap.parse_args()
    bad = 0
    for i in range(a.cases):
        seed = a.seed * 1_000_003 + i
        try:
            one_case(seed)
        except Exception as e:  # noqa: BLE001
            bad += 1
            print(f"FAIL seed {seed}: {type(e).__name__} {str(e)[:500]}", flush=True)
            if os.environ.get("FUZZ_TRACE"):
                traceback.print_exc()
    print(f"{a.cases - bad} / {a.cases} cases agree with the oracle", flush=True)
    sys.exit(1 if bad else 0)


if __name__ == "__main__":
    main()
