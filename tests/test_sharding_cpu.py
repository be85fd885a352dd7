"""hugectr_amd.sharding against the REFERENCE planner's own outputs (tests/golden/sharding_plans.json,
produced by importing R/samples/dlrm/sharding in the build container): identical shard_matrix and
shard_strategy -- or the identical error -- for every configuration."""
import json
import os
from argparse import Namespace

import pytest

HERE = os.path.dirname(os.path.abspath(__file__))
CASES = json.load(open(os.path.join(HERE, "golden", "sharding_plans.json")))


@pytest.mark.parametrize("case", CASES, ids=[c["name"] for c in CASES])
def test_plan_equals_reference(case):
    from hugectr_amd.sharding import generate_plan
    args = Namespace(**case["args"])
    call = lambda: generate_plan(case["slot_size_array"], case["multi_hot_sizes"], case["num_nodes"],
                                 case["num_gpus"], args, False)
    if "error" in case:
        with pytest.raises(Exception) as e:
            call()
        assert str(e.value) == case["error"]
        return
    matrix, strategy = call()
    assert matrix == case["shard_matrix"]
    assert [[k, v] for k, v in strategy] == case["shard_strategy"]


def test_mi355x_defaults_plan_the_criteo_tables():
    """MI355X ratios, 240 GB per GPU, one-hot Criteo-1TB tables: every table placed, lookups per
    GPU balanced (the cost model counts row reads + tables, all tables are equally hot here)"""
    from hugectr_amd.sharding import generate_plan, mi355x_args
    sizes = [39884406, 39043, 17289, 7420, 20263, 3, 7120, 1543, 63, 38532951, 2953546, 403346, 10,
             2208, 11938, 155, 4, 976, 14, 39979771, 25641295, 39664984, 585935, 12972, 108, 36]
    matrix, strategy = generate_plan(sizes, [1] * 26, 1, 8, mi355x_args(), False)
    assert strategy == [("mp", [str(i) for i in range(26)])]
    assert set(t for row in matrix for t in row) == {str(i) for i in range(26)}
    assert all(len(set(row)) == len(row) for row in matrix)
    assert max(len(r) for r in matrix) - min(len(r) for r in matrix) <= 1
    owners = {t: sum(t in row for row in matrix) for t in map(str, range(26))}
    gb = [sum(sizes[int(t)] / owners[t] for t in row) * 128 * 4e-9 for row in matrix]
    assert max(gb) < 240.0
