"""one extra leg of bench.py alone (argv[1]: dynamic | tiered | c1 | multi_hot), for a rocprofv3
kernel trace of it: python tools/leg_profile.py LEG"""
import json
import os
import sys

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
import torch  # noqa: E402
import bench  # noqa: E402

dev = torch.device("cuda:0")
torch.cuda.set_device(0)
leg = sys.argv[1]
if leg == "dynamic":
    r = bench.ebc_leg("multi_hot", 6, 3, dev, 1.1, dynamic=True)
elif leg == "multi_hot":
    r = bench.ebc_leg("multi_hot", 6, 3, dev, 1.1)
elif leg == "tiered":
    r = bench.tiered_leg(6, 3, dev, 1.1)
else:
    r = bench.small_config_leg("c1", 30, 10, dev)
print(json.dumps({k: v for k, v in r.items() if not isinstance(v, dict)}))
