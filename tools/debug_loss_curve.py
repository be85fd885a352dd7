import os, sys, tempfile, pathlib
sys.path.insert(0, "/root/repo"); sys.path.insert(0, "/root/repo/tests")
os.environ["HCTR_DEBUG_LOSS_CURVE"] = "1"
import hugectr
from oracle import pyoracle as oracle
import test_loss_curve_gpu as T
oracle.build()
T.STEPS = int(sys.argv[2]) if len(sys.argv) > 2 else 8
orig = hugectr.Model._run_batch
def rb(self, batch, train):
    r = orig(self, batch, train)
    return r
d = pathlib.Path(tempfile.mkdtemp())
got, want, h, table, ht = T._run(hugectr, oracle, d, sys.argv[1] if len(sys.argv) > 1 else "dlrm", False)
