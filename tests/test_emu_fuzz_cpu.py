"""A few seeds of the differential fuzzers of tests/emu on every CPU run (each fuzzer installs the
host interpreter in place of the GPU for its whole process, so it runs as a child process; the long
campaigns are run by hand: `python tests/emu/fuzz_*.py --seed S --cases N`)."""
import os
import subprocess
import sys

import pytest

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(HERE)
sys.path.insert(0, os.path.join(HERE, "emu"))
import emu  # noqa: E402

pytestmark = pytest.mark.skipif(not emu.available(), reason="no host clang++ / make")


@pytest.mark.parametrize("script,seed,cases", [("fuzz_ebc_dynamic.py", 0, 10), ("fuzz_ebc_dynamic.py", 30, 10),
                                               ("fuzz_det.py", 0, 10)])
def test_fuzzer_seeds_agree(script, seed, cases):
    emu.build()  # (the child would build it as well; here a failure reads better)
    env = dict(os.environ)
    for k in ("HCTR_EBC_DIRECT", "HCTR_DYNAMIC_FLAT"):
        env.pop(k, None)
    r = subprocess.run([sys.executable, os.path.join(HERE, "emu", script), "--seed", str(seed),
                        "--cases", str(cases)], cwd=ROOT, env=env, capture_output=True, text=True,
                       timeout=900)
    assert r.returncode == 0, r.stdout[-3000:] + r.stderr[-3000:]
    assert f"{cases} / {cases} cases agree" in r.stdout, r.stdout[-2000:]
