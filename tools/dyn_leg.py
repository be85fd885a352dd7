"""one dynamic-table leg of bench.py on its own (for a kernel trace):
   HCTR_DYNAMIC_FLAT=0|1 python tools/dyn_leg.py adagrad|adam|sgd [steps]"""
import json
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch  # noqa: E402

import bench  # noqa: E402

opt = sys.argv[1] if len(sys.argv) > 1 else "adagrad"
steps = int(sys.argv[2]) if len(sys.argv) > 2 else 6
torch.cuda.set_device(0)
r = bench.ebc_leg("multi_hot", steps, 3, torch.device("cuda", 0), 1.1, dynamic=True, optimizer=opt)
print(json.dumps({k: r[k] for k in ("forward_us", "backward_update_us", "forward_backward_update_us")}))
