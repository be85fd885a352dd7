#!/bin/bash
# functional check of bench.py's N-rank orchestration (default N = 2; usage: bench_ranks_one_gpu.sh [N]) on ONE GPU (gloo, collectives staged through
# the host): both exchange payloads and the warm-up auto-selection.  Not a measurement.
set -e
cd "$(dirname "$0")/.."
export HCTR_BENCH_BACKEND=gloo
N=${1:-2}
for ex in rows unique unique16 auto; do
  python -m torch.distributed.run --nnodes=1 --nproc-per-node $N --master-addr 127.0.0.1 \
    --master-port $((29610 + RANDOM % 200)) bench.py --gpus $N --steps 4 --warmup 2 --batch 8192 \
    --table-scale 0.02 --exchange $ex --tunable off 2>&1 | grep '^{' | python -c "
import sys, json
j = json.loads(sys.stdin.read())
print('$ex', j['n_gpus'], round(j['ms_per_step'], 2), 'ms', j['config']['exchange'], j['config']['exchange_warmup_ms_per_step'], 'loss', j['config']['final_loss'])"
done
