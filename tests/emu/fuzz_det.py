"""TEST INFRASTRUCTURE ONLY: randomized differential test of the dynamic embedding table
(hctr_det_*: the product's Python + the kernels' source under the host interpreter, tests/emu)
against the dict / numpy restatement in oracle/det_oracle.py -- random sequences of lookup (with
insertion and growth), scatter_add / scatter_update, remove, optimizer steps and export over
random class dimensions, key widths, id-space layouts (empty spaces, several spaces of one class)
and tiny initial capacities.

    python tests/emu/fuzz_det.py --seed 0 --cases 100"""
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
os.environ["HCTR_EMU"] = "1"

import fakecuda  # noqa: E402

fakecuda.install(os.environ.get("HCTR_EMU_VARIANT"))

import torch  # noqa: E402

from hugectr_amd import _lib  # noqa: E402
from hugectr_amd.dynamic_table import DynamicEmbeddingTable, DynamicTableOptimizer  # noqa: E402
from oracle import det_oracle as D  # noqa: E402
from util import assert_close  # noqa: E402

OPTS = {"sgd": (_lib.OPT_SGD, D.SGD, 0), "momentum": (_lib.OPT_MOMENTUM_SGD, D.MOMENTUM, 1),
        "nesterov": (_lib.OPT_NESTEROV, D.NESTEROV, 1), "adagrad": (_lib.OPT_ADAGRAD, D.ADAGRAD, 1),
        "rmsprop": (_lib.OPT_RMSPROP, D.RMSPROP, 1), "adam": (_lib.OPT_ADAM, D.ADAM, 2),
        "ftrl": (_lib.OPT_FTRL, D.FTRL, 2)}


def one_case(seed):
    rng = np.random.default_rng(seed)
    kb = int(rng.choice([8, 4]))
    ncls = int(rng.integers(1, 4))
    dims = [int(rng.choice([1, 4, 8, 20, 64, 128])) for _ in range(ncls)]
    cap0 = int(rng.choice([1, 8, 64, 1024]))
    name = str(rng.choice(list(OPTS)))
    code, ocode, ns = OPTS[name]
    desc = dict(seed=seed, kb=kb, dims=dims, cap0=cap0, opt=name)
    kw = dict(lr=0.05, scaler=2.0, beta1=0.9, beta2=0.999, epsilon=1e-6, momentum_factor=0.8,
              rmsprop_beta=0.95, ftrl_lambda1=0.01, ftrl_lambda2=0.02, ftrl_beta=0.5)
    t = DynamicEmbeddingTable(dims, "0.25", initial_capacity=cap0,
                              key_dtype=torch.int64 if kb == 8 else torch.uint32)
    opt = DynamicTableOptimizer(t, code, initial_capacity=cap0, **kw)
    ow = D.DetOracle(dims, 0.25)
    os_ = D.DetOracle([d * max(ns, 1) for d in dims], 0.0)
    hi = 2**40 if kb == 8 else 2**31
    pool = rng.integers(0, hi, size=int(rng.choice([5, 60, 800])), dtype=np.int64)

    def dev_keys(k):
        a = k.astype(np.int64) if kb == 8 else k.astype(np.uint32).view(np.int32)
        return torch.from_numpy(np.ascontiguousarray(a)).cuda()

    def spaces(unique):
        """a random id-space layout: 1-4 spaces (classes drawn with repetition), some empty"""
        nsp = int(rng.integers(1, 5))
        sp = [int(rng.integers(0, ncls)) for _ in range(nsp)]
        parts = []
        for _ in sp:
            n = int(rng.choice([0, 1, 7, 60, 300]))
            k = pool[rng.integers(0, pool.size, size=n)] if n else pool[:0]
            parts.append(np.unique(k) if unique else k)
        if unique:  # one class must not see a key twice in a call (scatter / update semantics)
            seen = {}
            for j, c in enumerate(sp):
                s = seen.setdefault(c, set())
                keep = [x for x in parts[j].tolist() if x not in s]
                s.update(keep)
                parts[j] = np.array(keep, dtype=np.int64)
        so = np.concatenate([[0], np.cumsum([p.size for p in parts])]).astype(np.int64).tolist()
        keys = np.concatenate(parts + [pool[:0]]).astype(np.int64)
        return keys, sp, so

    times = 0
    for step in range(int(rng.integers(3, 9))):
        op = str(rng.choice(["lookup", "lookup", "scatter_add", "scatter_update", "remove",
                             "update", "update"]))
        if op == "lookup":
            keys, sp, so = spaces(False)
            if keys.size == 0:
                continue
            got = t.lookup(dev_keys(keys), sp, so).cpu().numpy()
            assert np.array_equal(got, ow.lookup(keys, sp, so)), (desc, step, op)
        elif op in ("scatter_add", "scatter_update"):
            keys, sp, so = spaces(True)
            if keys.size == 0:
                continue
            lens = np.concatenate([[dims[c]] * (so[j + 1] - so[j]) for j, c in enumerate(sp)] + [[]])
            upd = rng.standard_normal(int(lens.sum())).astype(np.float32)
            fn = t.scatter_add if op == "scatter_add" else t.scatter_update
            fn(dev_keys(keys), torch.from_numpy(upd).cuda(), sp, so)
            ow.scatter(keys, upd, sp, so, add=(op == "scatter_add"))
        elif op == "remove":
            keys, sp, so = spaces(False)
            if keys.size == 0:
                continue
            t.remove(dev_keys(keys), sp, so)
            ow.remove(keys, sp, so)
            if ns:  # (the optimizer's state rows go with the weights in the reference's table)
                opt.states.remove(dev_keys(keys), sp, so)
                os_.remove(keys, sp, so)
        else:
            keys, sp, so = spaces(True)
            if keys.size == 0:
                continue
            if name != "ftrl":  # training order: the forward lookup has inserted the keys
                t.lookup(dev_keys(keys), sp, so)
                ow.lookup(keys, sp, so)
            lens = np.concatenate([[dims[c]] * (so[j + 1] - so[j]) for j, c in enumerate(sp)] + [[]])
            ev = np.concatenate([[0], np.cumsum(lens)]).astype(np.int32)
            wg = rng.standard_normal(int(ev[-1])).astype(np.float32)
            times += 1
            opt.update(dev_keys(keys), torch.from_numpy(ev).cuda(), torch.from_numpy(wg).cuda(), sp, so)
            D.update(ow, os_, ocode, keys, sp, so, ev, wg, lr=kw["lr"], scaler=kw["scaler"],
                     beta1=kw["beta1"], beta2=kw["beta2"], eps=kw["epsilon"],
                     momentum=kw["momentum_factor"], rms_beta=kw["rmsprop_beta"],
                     lambda1=kw["ftrl_lambda1"], lambda2=kw["ftrl_lambda2"],
                     ftrl_beta=kw["ftrl_beta"], times=times)
            got = t.lookup(dev_keys(keys), sp, so).cpu().numpy()
            assert_close(got, ow.lookup(keys, sp, so), 1e-4, 1e-5, f"{desc} weights step {step}")
            if ns:
                gs = opt.states.lookup(dev_keys(keys), sp, so).cpu().numpy()
                assert_close(gs, os_.lookup(keys, sp, so), 1e-4, 1e-6, f"{desc} state step {step}")
        assert t.size_per_class() == ow.size_per_class(), (desc, step, op, t.size_per_class(),
                                                           ow.size_per_class())
    for c in range(ncls):
        k, v = t.export(c)
        k = k.cpu().numpy()
        k = k.astype(np.int64) if kb == 8 else k.view(np.uint32).astype(np.int64)
        v = v.cpu().numpy()
        assert len(k) == len(ow.maps[c]) and len(set(k.tolist())) == len(k), (desc, "export", c)
        for kk, vv in zip(k.tolist(), v):
            assert np.allclose(vv, ow.maps[c][kk], rtol=1e-4, atol=1e-5), (desc, "export value", c)
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
