"""Generates tests/golden/sharding_plans.json by running the REFERENCE's own planner
(/root/reference/samples/dlrm/sharding, plain Python + numpy -- importable in the build container,
absent on the GPU box) on a list of configurations.  The fixture stores inputs and outputs; the
parity test (tests/test_sharding_cpu.py) feeds the inputs to hugectr_amd.sharding and compares.

    python tests/golden/make_sharding_golden.py
"""
import ast
import json
import os
import sys
from argparse import Namespace

import numpy as np

REF = "/root/reference/samples/dlrm"
HERE = os.path.dirname(os.path.abspath(__file__))


def mlperf_tables():
    """TABLE_SIZE_ARRAY / MULTI_HOT_SIZES of the sample (train.py:30-85), read, not imported (the
    script itself needs mpi4py / hugectr)"""
    tree = ast.parse(open(os.path.join(REF, "train.py")).read())
    out = {}
    for node in tree.body:
        if isinstance(node, ast.Assign) and getattr(node.targets[0], "id", "") in (
                "TABLE_SIZE_ARRAY", "MULTI_HOT_SIZES"):
            out[node.targets[0].id] = ast.literal_eval(node.value)
    return out["TABLE_SIZE_ARRAY"], out["MULTI_HOT_SIZES"]


def main():
    sys.path.insert(0, REF)
    import sharding  # the reference package
    sizes, hot = mlperf_tables()
    base = dict(optimizer="adagrad", ev_size=128, dp_sharding_threshold=0.0, num_gpus_per_node=8,
                mem_comm_bw_ratio=3.35e12 / 450e9, mem_comm_work_ratio=8 / 2,
                memory_cap_for_embedding=60.0)  # the sample's defaults (train.py:203-247)
    rng = np.random.default_rng(7)
    cases = []

    def add(name, sizes_, hot_, nodes, gpus, **kw):
        a = dict(base)
        a.update(kw)
        cases.append(dict(name=name, slot_size_array=[int(x) for x in sizes_],
                          multi_hot_sizes=[int(x) for x in hot_], num_nodes=nodes, num_gpus=gpus,
                          args=a))

    for plan in ("round_robin", "uniform", "auto"):
        for gpus in (1, 2, 4, 8):
            add(f"mlperf_{plan}_{gpus}", sizes, hot, 1, gpus, sharding_plan=plan,
                num_gpus_per_node=gpus)
    add("mlperf_auto_16_two_nodes", sizes, hot, 2, 16, sharding_plan="auto")
    add("mlperf_hier_auto_2x8", sizes, hot, 2, 16, sharding_plan="hier_auto")
    add("mlperf_hier_auto_4x4", sizes, hot, 4, 16, sharding_plan="hier_auto", num_gpus_per_node=4)
    add("mlperf_auto_sgd_tight_memory", sizes, hot, 1, 8, sharding_plan="auto", optimizer="sgd",
        memory_cap_for_embedding=16.0)
    add("mlperf_auto_dp_threshold", sizes, hot, 1, 8, sharding_plan="auto",
        dp_sharding_threshold=0.01)
    add("mlperf_auto_mi355x_ratios", sizes, hot, 1, 8, sharding_plan="auto", optimizer="sgd",
        mem_comm_bw_ratio=8.0e12 / (7 * 153e9), memory_cap_for_embedding=240.0)
    one_hot = [39884406, 39043, 17289, 7420, 20263, 3, 7120, 1543, 63, 38532951, 2953546, 403346,
               10, 2208, 11938, 155, 4, 976, 14, 39979771, 25641295, 39664984, 585935, 12972, 108,
               36]  # Criteo-1TB one-hot table sizes (BASELINE config 3)
    add("criteo1tb_one_hot_auto_8", one_hot, [1] * 26, 1, 8, sharding_plan="auto", optimizer="sgd")
    add("criteo1tb_one_hot_auto_4", one_hot, [1] * 26, 1, 4, sharding_plan="auto", optimizer="sgd",
        num_gpus_per_node=4)
    for i in range(12):  # random shapes: ties in hotness, tiny and huge tables, odd GPU counts
        n = int(rng.integers(3, 40))
        s = (10 ** rng.uniform(0.5, 7.4, size=n)).astype(np.int64) + 1
        h = rng.choice([1, 1, 1, 2, 3, 5, 8, 20, 100], size=n)
        g = int(rng.choice([2, 3, 4, 6, 8]))
        add(f"random_{i}", s, h, 1, g, sharding_plan="auto", num_gpus_per_node=g,
            optimizer=str(rng.choice(["sgd", "adagrad"])),
            memory_cap_for_embedding=float(rng.choice([24.0, 40.0, 60.0])),
            dp_sharding_threshold=float(rng.choice([0.0, 0.0, 0.001])))
    for c in cases:
        try:
            m, s = sharding.generate_plan(c["slot_size_array"], c["multi_hot_sizes"], c["num_nodes"],
                                          c["num_gpus"], Namespace(**c["args"]), False)
            c["shard_matrix"], c["shard_strategy"] = m, [[k, v] for k, v in s]
        except Exception as e:  # an error is a result too (e.g. OOM with every plan)
            c["error"] = str(e)
    with open(os.path.join(HERE, "sharding_plans.json"), "w") as f:
        json.dump(cases, f, indent=0)
    print(len(cases), "cases;", sum("error" in c for c in cases), "raise")


if __name__ == "__main__":
    main()
