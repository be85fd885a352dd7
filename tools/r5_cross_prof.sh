#!/bin/bash
# (on the GPU box) kernel stats of MultiCross v2 forward + backward at X1's shape
cd "$(dirname "$0")/.."
mkdir -p gpurun_out
cat > /tmp/cross_drv.py <<'P'
import sys, torch
sys.path.insert(0, "/root/repo")
from hugectr_amd.layers import MultiCrossLayer
B, w, p, L = int(sys.argv[1]), 3456, 512, 3
layer = MultiCrossLayer(w, L, p).cuda()
x = torch.randn(B, w, device="cuda").half().requires_grad_(True)
g = torch.randn(B, w, device="cuda").half()
for _ in range(6):
    o = layer(x); o.backward(g); x.grad = None; layer.zero_grad(set_to_none=True)
torch.cuda.synchronize()
P
for B in ${BATCHES:-8192}; do
  rm -rf /tmp/kc
  ( cd /tmp && export TMPDIR=/tmp && env $1 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/kc -o k -- python /tmp/cross_drv.py $B > /tmp/kc.out 2>&1 )
  echo "==== B=$B $1"
  python - <<P
import csv,glob
f=glob.glob('/tmp/kc/**/*kernel_stats.csv',recursive=True)[0]
rows=list(csv.DictReader(open(f)))
rows.sort(key=lambda r:-float(r['TotalDurationNs']))
for r in rows[:14]:
    print(f"{float(r['TotalDurationNs'])/1e3:10.1f} us  n={r['Calls']:>4}  avg={float(r['AverageNs'])/1e3:8.1f}  {r['Name'][:110]}")
P
done
