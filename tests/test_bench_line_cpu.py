"""The bench contract's stdout line must survive the driver's channel: one line, <= 4 KB, valid
JSON, the contract's keys present (a 21 KB line was not parsed).  The full record
goes to a side file.  Canned input: round 4's committed full record (profiles/)."""
import copy
import json
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)

REQUIRED = ("metric", "value", "unit", "n_gpus", "steps", "warmup", "ms_per_step",
            "higher_is_better", "scaling", "vs_baseline", "dtype", "data", "config", "roofline",
            "cpu_baseline", "extra_file")


def _canned():
    return json.load(open(os.path.join(ROOT, "profiles", "r4_bench_n1_final.json")))


def _check(line):
    assert "\n" not in line
    assert len(line.encode()) < 4096, len(line)
    j = json.loads(line)
    for k in REQUIRED:
        assert k in j, k
    assert isinstance(j["config"], dict) and "workload" in j["config"]
    assert "model" not in j["config"]
    r = j["roofline"]
    for k in ("bound", "achieved", "peak", "unit", "frac", "traffic"):
        assert k in r, k
    assert abs(r["frac"] - r["achieved"] / r["peak"]) < 1e-3
    c = j["cpu_baseline"]
    for k in ("value", "unit", "cores", "kind", "sample"):
        assert k in c, k
    return j


def test_line_of_the_round4_record_fits_and_parses():
    import bench
    full = _canned()
    assert len(json.dumps(full)) > 20000  # the record that broke the channel
    j = _check(bench.compact_line(full, "bench_extra.json"))
    assert j["value"] == bench._r(full["value"])
    assert j["dtype"] == "fp16"
    assert j["roofline_update"]["frac"] == bench._r(full["roofline_update"]["frac"])
    assert "roofline_uniform" in j and "roofline_index" in j


def test_line_of_an_eight_rank_record_fits():
    import bench
    full = _canned()
    full["n_gpus"] = 8
    full.pop("cpu_baseline")
    full["cpu_baseline"] = {"value": 1.0, "unit": "samples/s", "cores": 1, "kind": "reference",
                            "sample": "x" * 2000}
    rank = {"rank": 0, "slots": 4, "table_rows": 10 ** 8, "stage_us": full["stage_us_per_step"],
            "exchange": {"payload": "rows", "note": "y" * 500}, "new_keys_per_step": 1.5}
    full["per_rank"] = [copy.deepcopy(rank) for _ in range(8)]
    strong = copy.deepcopy({k: v for k, v in full.items() if k not in ("extra", "strong")})
    full["strong"] = strong
    j = _check(bench.compact_line(full, "/some/long/path/" + "d" * 100 + "/bench_extra.json"))
    assert "per_rank" not in j


def test_pathological_record_still_fits():
    import bench
    full = _canned()
    full["config"]["workload"] = "w" * 5000
    full["extra"] = {f"leg{i}": {"ms_per_step": 1.0} for i in range(400)}
    _check(bench.compact_line(full, "bench_extra.json"))


def test_emit_writes_the_file_and_prints_one_line(tmp_path):
    code = ("import json, sys; sys.path.insert(0, %r); import bench; "
            "bench.emit(json.load(open(%r)), %r)" % (
                ROOT, os.path.join(ROOT, "profiles", "r4_bench_n1_final.json"),
                str(tmp_path / "x.json")))
    p = subprocess.run([sys.executable, "-c", code], capture_output=True, text=True, check=True)
    lines = p.stdout.strip().split("\n")
    assert len(lines) == 1
    j = _check(lines[0])
    assert json.load(open(j["extra_file"]))["extra"].keys() == _canned()["extra"].keys()
