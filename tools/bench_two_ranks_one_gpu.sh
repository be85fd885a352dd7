#!/bin/bash
# functional check of bench.py's N = 2 orchestration on ONE GPU (gloo, collectives staged through
# the host): both exchange payloads and the warm-up auto-selection.  Not a measurement.
set -e
cd "$(dirname "$0")/.."
export HCTR_BENCH_BACKEND=gloo
for ex in rows unique unique16 auto; do
  python -m torch.distributed.run --nnodes=1 --nproc-per-node 2 --master-addr 127.0.0.1 \
    --master-port $((29610 + RANDOM % 200)) bench.py --gpus 2 --steps 4 --warmup 2 --batch 8192 \
    --table-scale 0.02 --exchange $ex --tunable off 2>&1 | grep '^{' | python -c "
import sys, json
j = json.loads(sys.stdin.read())
print('$ex', j['n_gpus'], round(j['ms_per_step'], 2), 'ms', j['config']['exchange'], j['config']['exchange_warmup_ms_per_step'], 'loss', j['config']['final_loss'])"
done
