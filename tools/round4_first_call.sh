#!/bin/bash
# (on the GPU box) ONE call that decides the prepared variants: the full `-m gpu` suite on the
# product, then per variant the index / embedding / sort tests on the hardware and the kernel medians
# of the bench's main leg next to the product's; the sort variant also with its tile / scan knobs.
# Everything lands in gpurun_out/round4_*.  Each step has its own timeout (a hung kernel must not
# eat the budget).
cd "$(dirname "$0")/.."
mkdir -p gpurun_out
timeout 900 python -m pytest tests -x -q -m gpu > gpurun_out/round4_tests_product.log 2>&1; echo "product tests rc=$?" | tee gpurun_out/round4_summary.txt
for v in index_segments index_segments_v2 sort_first_pass; do
  [ -f hugectr_amd/libhugectr_amd_$v.so ] || { echo "$v: not built" | tee -a gpurun_out/round4_summary.txt; continue; }
  timeout 900 bash tools/variant.sh gpu $v > gpurun_out/round4_variant_$v.log 2>&1
  echo "== $v" >> gpurun_out/round4_summary.txt; cat gpurun_out/variant_$v.txt >> gpurun_out/round4_summary.txt 2>/dev/null
done
if [ -f hugectr_amd/libhugectr_amd_sort_first_pass.so ]; then
  for r in 16 8 4; do for b in 32 8; do
    echo "== sort_first_pass rounds $r scanbins $b" >> gpurun_out/round4_summary.txt
    HCTR_LIB_VARIANT=sort_first_pass HCTR_RS_ROUNDS=$r HCTR_RS_SCANBINS=$b timeout 240 python bench.py --extra none --no-cpu-baseline --steps 30 --warmup 8 2>/dev/null | grep "^{" | tail -1 | \
      python -c "import json,sys; j=json.loads(sys.stdin.read()); print('ms_per_step', round(j['ms_per_step'],4), 'update us', j['roofline_update']['us'], 'stages', j['stage_us_per_step'])" >> gpurun_out/round4_summary.txt 2>&1
  done; done
fi
cat gpurun_out/round4_summary.txt
