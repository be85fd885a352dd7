#!/bin/bash
# functional check of the N-rank training step bench.py drives through hugectr.Model (default
# N = 2; usage: bench_ranks_one_gpu.sh [N]) on ONE GPU (gloo, collectives staged through the host):
# every exchange payload, the auto-selection, overlap off, weak and strong scaling.  Not a
# measurement.
set -e
cd "$(dirname "$0")/.."
export HCTR_BENCH_BACKEND=gloo
N=${1:-2}
run() {
  python -m torch.distributed.run --nnodes=1 --nproc-per-node $N --master-addr 127.0.0.1 \
    --master-port $((29610 + RANDOM % 200)) bench.py --gpus $N --steps 4 --warmup 2 --batch 8192 \
    --table-scale 0.02 --tunable off --extra-file /tmp/bench_ranks_extra.json "$@" 2>&1 | grep '^{' | python -c "
import sys, json
line = sys.stdin.read().strip()
assert len(line.encode()) < 4096, len(line)
json.loads(line)
j = json.load(open('/tmp/bench_ranks_extra.json'))
for name, l in (('weak', j), ('strong', j.get('strong'))):
    if isinstance(l, dict) and 'ms_per_step' in l:
        c = l['config']
        print('$*', name, l['n_gpus'], 'gpus', round(l['ms_per_step'], 2), 'ms', c['exchange'],
              c['exchange_selection_ms_per_step'], 'B', c['global_batch'], 'loss', c['final_loss'])
    elif isinstance(l, dict):
        print('$*', name, l)"
}
for ex in rows unique unique16 auto; do
  run --exchange $ex
done
run --exchange rows --no-overlap --scaling weak
