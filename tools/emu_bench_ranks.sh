#!/bin/bash
# (no GPU) the driver's N-rank bench command line, end to end, on the kernels' source under the host
# interpreter (tests/emu) with gloo: one process per rank, tiny batch and tables -- a LOGIC check of
# the N > 1 schedule (uneven slot counts, exchange selection, weak + strong lines, the <= 4 KB
# stdout line), not a measurement.
#   tools/emu_bench_ranks.sh 8 [rows|unique|unique16|auto]
N=${1:-8}; EX=${2:-auto}
cd "$(dirname "$0")/.."
X=/tmp/emu_bench_extra_$N.json
HCTR_EMU=1 PYTHONPATH=$PWD/tests/emu/site:$PYTHONPATH HCTR_BENCH_BACKEND=gloo \
  python -m torch.distributed.run --nnodes=1 --nproc-per-node $N --master-addr 127.0.0.1 --master-port 29641 \
  bench.py --gpus $N --steps 3 --warmup $([ "$EX" = auto ] && echo 20 || echo 3) --batch 256 --table-scale 0.001 \
  --no-cpu-baseline --exchange $EX --extra-file $X 2>&1 | grep "^{" | python -c "
import json, sys
line = sys.stdin.read().strip()
assert len(line.encode()) < 4096 and '\n' not in line, len(line)
l = json.loads(line)
j = json.load(open('$X')); c = j['config']
assert l['n_gpus'] == j['n_gpus'] and abs(l['value'] / j['value'] - 1) < 1e-3 and 'strong' in l
print('line bytes', len(line), 'n_gpus', j['n_gpus'], 'exchange', c['exchange'], 'loss', c['final_loss'],
      'strong loss', j['strong']['config']['final_loss'], 'slots per rank', [r['slots'] for r in j['per_rank']])"
