#!/bin/bash
# A/B of index-stage variants on ONE box: parity tests per variant, then kernel medians of the main
# leg and the all-unseen (uniform keys) leg for the product and every variant.
#   bash tools/idx_ab.sh "idx7 idx8"   -> gpurun_out/r6_idx_ab.txt
VARS=${1:-"idx7 idx8"}
cd "$(dirname "$0")/.."
mkdir -p gpurun_out
OUT=gpurun_out/r6_idx_ab.txt; : > $OUT
FILES="tests/test_hash_gpu.py tests/test_embedding_gpu.py tests/test_golden_gpu.py tests/test_det_gpu.py tests/test_ebc_dynamic_gpu.py tests/test_fullsize_gpu.py"
for V in $VARS; do
  echo "== tests: $V" >> $OUT
  HCTR_LIB_VARIANT=$V timeout 900 python -m pytest $FILES -x -q -m gpu -p no:cacheprovider 2>&1 | tail -3 >> $OUT
done
for V in "" $VARS; do
  echo "== main leg, library: ${V:-product}" >> $OUT
  HCTR_LIB_VARIANT=$V timeout 300 bash tools/profile_index_stage.sh 2>&1 | tail -8 >> $OUT
  grep "^{" /tmp/ks.out | tail -1 | python -c "import json,sys; j=json.loads(sys.stdin.read()); print('ms_per_step', j['ms_per_step'], 'index', j.get('roofline_index',{}).get('us'), 'update', j.get('roofline_update',{}).get('us'))" >> $OUT 2>&1 || true
  echo "== uniform leg (every key unseen), library: ${V:-product}" >> $OUT
  HCTR_LIB_VARIANT=$V timeout 400 python bench.py --extra uniform --no-cpu-baseline --steps 5 --warmup 3 --extra-file /tmp/ex_$V.json > /tmp/uni_$V.out 2>&1
  python - "$V" >> $OUT 2>&1 <<'PY'
import json, sys
v = sys.argv[1]
try:
    d = json.load(open(f"/tmp/ex_{v}.json"))
    u = d.get("extra", d).get("uniform_big_tables", {})
    print("uniform ms_per_step", u.get("ms_per_step"), "index", json.dumps(u.get("roofline_index")), "stage", json.dumps(u.get("stage_us_per_step"))[:400])
except Exception as e:
    print("no extra file:", e)
PY
done
cat $OUT
